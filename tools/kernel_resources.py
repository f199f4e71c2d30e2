"""profiles/rN_kernel_resources.txt: registers, scratch, occupancy and LDS of every kernel, from
hipcc -Rpass-analysis=kernel-resource-usage (no GPU needed).   python tools/kernel_resources.py 3 > profiles/r3_kernel_resources.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gpu_pattern_matching_amd", "csrc")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.splitlines()


def main(rnd):
    print("kernel resource usage (hipcc -Rpass-analysis=kernel-resource-usage, gfx950, -O3), round %s" % rnd)
    print("%-58s %6s %6s %8s %9s %10s" % ("kernel", "VGPRs", "SGPRs", "scratch", "occupancy", "LDS bytes"))
    for src in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-x", "hip",
               "--cuda-device-only", "-c", os.path.join(CSRC, src), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
        rows, cur = [], None
        for line in err.splitlines():
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                cur = {"name": m.group(1)}
                rows.append(cur)
                continue
            for key, pat in (("v", r" VGPRs: (\d+)"), ("s", r"TotalSGPRs: (\d+)"), ("x", r"ScratchSize \[bytes/lane\]: (\d+)"),
                             ("o", r"Occupancy \[waves/SIMD\]: (\d+)"), ("l", r"LDS Size \[bytes/block\]: (\d+)")):
                m = re.search(pat, line)
                if m and cur is not None:
                    cur[key] = int(m.group(1))
        names = demangle([r["name"] for r in rows])
        for r, nm in zip(rows, names):
            nm = re.sub(r"\(anonymous namespace\)::", "", nm)
            nm = re.sub(r"^void ", "", nm)
            nm = re.sub(r"\(.*\)$", "", nm)
            print("%-58s %6d %6d %8d %9d %10d" % (nm[:58], r.get("v", -1), r.get("s", -1), r.get("x", -1), r.get("o", -1), r.get("l", -1)))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "?")
