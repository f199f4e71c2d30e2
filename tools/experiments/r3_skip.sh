#!/bin/bash
# what the latency-bound kernels cost the job: the 200-step bench without the emit kernel, without check and emit (results void)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for SK in x e ce; do for W in 3 4; do
  ACM_SIEVE_SKIP=$SK timeout -k 10 300 python3 bench.py --steps 200 --texts 64 --workers $W --repeats 3 --sub= --no-extra --no-cpu-baseline --no-e2e --no-verify > gpurun_out/r3k_$SK$W.json 2> gpurun_out/r3k_$SK$W.err || { tail -5 gpurun_out/r3k_$SK$W.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3k_$SK$W.json')); print('skip $SK workers $W:', d['value'], 'GB/s', round(d['ms_per_step']*1000,2), 'us/step', d['blocks_ms'])"
done; done
