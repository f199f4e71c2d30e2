#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for G in 7 5 4 3 2; do for W in 3 4; do
  timeout -k 10 300 python3 bench.py --group $G --workers $W --steps 20 --warmup 5 --texts 64 --repeats 9 --sub= --no-extra --no-cpu-baseline --no-e2e --no-verify > gpurun_out/r3sh_$G$W.json 2> gpurun_out/r3sh_$G$W.err || { tail -5 gpurun_out/r3sh_$G$W.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3sh_$G$W.json')); print('group $G workers $W:', d['value'], 'GB/s', round(d['ms_per_step']*1000,2), 'us/step', d['blocks_ms'])"
done; done
