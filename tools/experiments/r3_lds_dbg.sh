#!/bin/bash
# where the LDS walk's time goes: the kernel with parts switched off (timing only, results wrong)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for D in 0 1 2 3 4; do
  ACM_LDS_DEBUG=$D timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3dbg_$D -- python3 bench.py --workload sentiment --mode chain --workers 1 --group 1 --steps 30 --warmup 4 --repeats 2 --texts 8 --sub= --no-extra --no-cpu-baseline --no-e2e --no-verify > gpurun_out/r3dbg_$D.json 2> gpurun_out/r3dbg_$D.err || { tail -5 gpurun_out/r3dbg_$D.err; exit 1; }
  echo "== debug $D"; cat $(find gpurun_out/r3dbg_$D -name "*kernel_stats.csv" | head -1) | cut -c1-130 | grep "k_lds"
done
