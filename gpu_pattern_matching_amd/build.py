"""Build libacmatch.so in-tree with hipcc for gfx950.

    python -m gpu_pattern_matching_amd.build [--force] [--keep-temps]

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with
the source tree to the GPU box.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "_build")
LIB = os.path.join(PKG, "libacmatch.so")
CLI = os.path.join(PKG, "acm_grep")          # native CLI (csrc/acm_grep.cpp), host code only

SOURCES = ["automaton.cpp", "compact_tables.cpp", "multi.cpp", "device_dfa.hip", "scan.hip", "lds_walk.hip", "sparse.hip", "post.hip", "runtime.hip", "compat.hip"]
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _deps():
    out = [os.path.join(ROOT, "include", "acmatch.h")]
    for f in os.listdir(CSRC):
        out.append(os.path.join(CSRC, f))
    return out


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, keep_temps=False, verbose=False):
    """Compile every translation unit and link libacmatch.so. Returns its path."""
    if not force and not keep_temps and not _stale(LIB, _deps()) and not _stale(CLI, _deps() + [LIB]):
        return LIB          # up to date (object files are scratch and need not exist)
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-I" + os.path.join(ROOT, "include"),
             "-I" + CSRC, "-Wall", "-Wno-unused-function", "-Wno-unused-value"]
    if keep_temps:
        flags += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
    headers = [d for d in _deps() if d.endswith(".h")]
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + flags + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd, cwd=OBJ)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    if force or _stale(CLI, [os.path.join(CSRC, "acm_grep.cpp"), os.path.join(ROOT, "include", "acmatch.h"), LIB]):
        cmd = [shutil.which("g++") or "g++", "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"),
               os.path.join(CSRC, "acm_grep.cpp"), "-o", CLI, "-L" + PKG, "-lacmatch", "-lpthread",
               "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, keep_temps="--keep-temps" in sys.argv, verbose=True)
    print(path)
