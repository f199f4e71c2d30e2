#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_gpu_ldswalk.py tests/test_gpu_carry.py -x -q -p no:cacheprovider > gpurun_out/r3w2_pytest.log 2>&1; tail -5 gpurun_out/r3w2_pytest.log
grep -q "passed" gpurun_out/r3w2_pytest.log || exit 1
grep -q "failed\|error" gpurun_out/r3w2_pytest.log && exit 1
for ST in 200 20; do
  timeout -k 10 300 python3 bench.py --workload sentiment --steps $ST --warmup 10 --texts 64 --sub= --no-cpu-baseline --no-e2e > gpurun_out/r3w2_$ST.json 2> gpurun_out/r3w2_$ST.err || { tail -5 gpurun_out/r3w2_$ST.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3w2_$ST.json')); print('steps $ST:', d['value'], 'GB/s', round(d['ms_per_step']*1000,2), 'us/step', d['parity'][:9], d['config']['workers'], d['blocks_ms'], d.get('roofline_one_batch_in_flight'), d.get('roofline_one_group_in_flight'))"
done
