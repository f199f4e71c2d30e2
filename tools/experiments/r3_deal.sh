#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for WL in clamav2000 sentiment; do for D in groups even; do
  timeout -k 10 300 python3 bench.py --workload $WL --deal $D --steps 20 --warmup 5 --sub= --no-extra --no-cpu-baseline --no-e2e > gpurun_out/r3d_${WL}_$D.json 2> gpurun_out/r3d_${WL}_$D.err || { tail -5 gpurun_out/r3d_${WL}_$D.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3d_${WL}_$D.json')); print('$WL $D:', d['value'], 'GB/s', round(d['ms_per_step']*1000,2), 'us/step', d['parity'][:9], d['config']['workers'], d['blocks_ms'], d['roofline']['batches_per_launch'], d['roofline']['launches_timed'])"
done; done
