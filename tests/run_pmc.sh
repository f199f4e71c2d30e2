#!/bin/bash
# gpurun helper: HBM traffic counters for the bench command, one counter per pass
# (FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2 -- MI355X_MICROARCH.md, rocprofv3 PMC slots)
TAG=${1:-pmc}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/${TAG}_$ctr -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/${TAG}_$ctr.json 2> gpurun_out/${TAG}_$ctr.err || { tail -5 gpurun_out/${TAG}_$ctr.err; exit 1; }
done
python3 - <<PY
import csv, glob, collections
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob("gpurun_out/${TAG}_%s/*/*counter_collection.csv" % ctr)
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != ctr:
                continue
            k = r["Kernel_Name"]
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    for k, (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        if "k_" in k:
            print("%-11s %-60s launches %4d  avg %12.1f" % (ctr, k[:60], n, v / n))
PY
