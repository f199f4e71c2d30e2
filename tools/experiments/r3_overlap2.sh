#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for P in 0 1; do for ST in 192 20; do
  ACM_LDS_SCATTER_PRIO=$P timeout -k 10 300 python3 bench.py --workload sentiment --steps $ST --texts 64 --sub= --no-extra --no-cpu-baseline --no-e2e > gpurun_out/r3o2_p${P}s$ST.json 2> gpurun_out/r3o2_p${P}s$ST.err || { tail -5 gpurun_out/r3o2_p${P}s$ST.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3o2_p${P}s$ST.json')); print('prio $P steps $ST:', d['value'], 'GB/s', round(d['ms_per_step']*1000,2), 'us/step', d['parity'][:9], d['config']['workers'], d['blocks_ms'])"
done; done
