#!/bin/bash
# kernel timeline of the headline bench (who overlaps whom)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
WL=${1:-clamav2000}
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3t2 -- python3 bench.py --workload $WL --steps 192 --texts 64 --repeats 2 --warmup 16 --sub= --no-extra --no-cpu-baseline --no-e2e --no-verify > gpurun_out/r3t2.json 2> gpurun_out/r3t2.err || { tail -5 gpurun_out/r3t2.err; exit 1; }
f=$(find gpurun_out/r3t2 -name "*kernel_trace.csv" | head -1)
python3 tools/experiments/r3_trace.py $f k_sieve > gpurun_out/r3t2_$WL.txt
rm -rf gpurun_out/r3t2
python3 -c "
import json; d=json.load(open('gpurun_out/r3t2.json')); print(d['value'], d['ms_per_step'])"
