#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_scan.py -x -q -p no:cacheprovider 2>&1 | tail -2
for WL in clamav2000 clamav15000; do
  timeout -k 10 300 python3 bench.py --workload $WL --mode chain --steps 60 --warmup 6 --texts 32 --repeats 3 --sub= --no-cpu-baseline --no-e2e > gpurun_out/r3cs4_$WL.json 2> gpurun_out/r3cs4_$WL.err || { tail -5 gpurun_out/r3cs4_$WL.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3cs4_$WL.json')); print('$WL chain, 3 workers:', d['value'], 'GB/s', round(d['ms_per_step']*1000,2), 'us/step', d['parity'][:9], d['roofline_one_batch_in_flight'])"
done
