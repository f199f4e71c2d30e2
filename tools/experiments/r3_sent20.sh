#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for W in 2 3; do for G in 16 5 4; do
  timeout -k 10 300 python3 bench.py --workload sentiment --workers $W --group $G --steps 20 --warmup 5 --texts 64 --repeats 7 --sub= --no-extra --no-cpu-baseline --no-e2e --no-verify > gpurun_out/r3s20_$W$G.json 2> gpurun_out/r3s20_$W$G.err || { tail -5 gpurun_out/r3s20_$W$G.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3s20_$W$G.json')); print('workers $W group $G:', d['value'], 'GB/s', round(d['ms_per_step']*1000,2), 'us/step', d['blocks_ms'], d['config']['batches_per_launch_group'])"
done; done
