#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for G in 16 12 8; do for W in 3 4; do for ST in 200 20; do
  timeout -k 10 300 python3 bench.py --group $G --workers $W --steps $ST --warmup 10 --texts 64 --sub= --no-extra --no-cpu-baseline --no-e2e --no-verify > gpurun_out/r3g_$G$W$ST.json 2> gpurun_out/r3g_$G$W$ST.err || { tail -5 gpurun_out/r3g_$G$W$ST.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3g_$G$W$ST.json')); print('group $G workers $W steps $ST:', d['value'], 'GB/s', round(d['ms_per_step']*1000,2), 'us/step', d['blocks_ms'])"
done; done; done
