"""Summarise rocprofv3 --pmc counter CSVs into profiles/<tag>_traffic.json.

    python tests/pmc_summarize.py gpurun_out/pmc1 profiles/r1_traffic.json

Per kernel: average FETCH_SIZE / WRITE_SIZE per launch (the counters are in KiB) and the HBM
bytes per launch with the gfx950 correction MI355X_MICROARCH.md prescribes for 16-B-per-lane
loads: FETCH_SIZE counts 128-B requests as 64 B, so reads are doubled; writes are taken as is.
"""
import collections
import csv
import glob
import json
import sys


def main(prefix, out):
    res = collections.defaultdict(dict)
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for f in glob.glob("%s_%s/*/*counter_collection.csv" % (prefix, ctr)):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == ctr:
                    agg[r["Kernel_Name"]][0] += float(r["Counter_Value"])
                    agg[r["Kernel_Name"]][1] += 1
        for k, (v, n) in agg.items():
            res[k][ctr + "_KiB_per_launch"] = v / n
            res[k]["launches"] = n
    for k, d in res.items():
        f = d.get("FETCH_SIZE_KiB_per_launch", 0.0)
        w = d.get("WRITE_SIZE_KiB_per_launch", 0.0)
        d["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, d in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
        print("%-70s %10.1f MB/launch" % (k[:70], d["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
