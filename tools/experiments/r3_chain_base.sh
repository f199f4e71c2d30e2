#!/bin/bash
# round 3 baseline: exclusive per-kernel times of the chain pipeline (one worker, no groups) on clamav2000 / clamav15000 / sentiment
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for WL in clamav2000 clamav15000 sentiment; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3base_$WL -- python3 bench.py --workload $WL --mode chain --workers 1 --group 1 --steps 40 --warmup 4 --repeats 2 --texts 8 --sub= --no-extra --no-cpu-baseline --no-e2e > gpurun_out/r3base_$WL.json 2> gpurun_out/r3base_$WL.err || { tail -5 gpurun_out/r3base_$WL.err; exit 1; }
  echo "== $WL"; cut -c1-300 gpurun_out/r3base_$WL.json
  cat $(find gpurun_out/r3base_$WL -name "*kernel_stats.csv" | head -1) | cut -c1-160 | head -8
done
