#!/bin/bash
# kernel timeline of the sentiment bench with 2 and 3 workers (who overlaps whom)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for W in 2; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3t_w$W -- python3 bench.py --workload sentiment --steps 192 --texts 64 --workers $W --repeats 2 --warmup 16 --sub= --no-extra --no-cpu-baseline --no-e2e --no-verify > gpurun_out/r3t_w$W.json 2> gpurun_out/r3t_w$W.err || { tail -5 gpurun_out/r3t_w$W.err; exit 1; }
  f=$(find gpurun_out/r3t_w$W -name "*kernel_trace.csv" | head -1)
  python3 tools/experiments/r3_trace.py $f > gpurun_out/r3t_w$W.txt
  rm -rf gpurun_out/r3t_w$W
  tail -40 gpurun_out/r3t_w$W.txt
done
