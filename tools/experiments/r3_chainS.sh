#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for S in 0 64 128 256; do
  timeout -k 10 300 python3 bench.py --mode chain --chain-bytes $S --workers 1 --steps 40 --warmup 5 --texts 16 --repeats 3 --sub= --no-cpu-baseline --no-e2e > gpurun_out/r3cs_$S.json 2> gpurun_out/r3cs_$S.err || { tail -5 gpurun_out/r3cs_$S.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3cs_$S.json')); print('chain bytes $S:', d['value'], 'GB/s', round(d['ms_per_step']*1000,2), 'us/step', d['parity'][:9], d['roofline_one_batch_in_flight'])"
done
for S in 0 64 128; do
  timeout -k 10 300 python3 bench.py --mode chain --chain-bytes $S --steps 60 --warmup 6 --texts 32 --repeats 3 --sub= --no-extra --no-cpu-baseline --no-e2e --no-verify > gpurun_out/r3cs3_$S.json 2> gpurun_out/r3cs3_$S.err || { tail -5 gpurun_out/r3cs3_$S.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3cs3_$S.json')); print('3 workers, chain bytes $S:', d['value'], 'GB/s', round(d['ms_per_step']*1000,2), 'us/step')"
done
