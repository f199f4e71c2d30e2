#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python3 -m pytest tests/test_gpu_sparse.py -x -q -p no:cacheprovider > gpurun_out/r3w4_pytest.log 2>&1; tail -3 gpurun_out/r3w4_pytest.log
grep -q "passed" gpurun_out/r3w4_pytest.log || exit 1
grep -q "failed\|error" gpurun_out/r3w4_pytest.log && exit 1
for ST in 200 20; do
  timeout -k 10 300 python3 bench.py --steps $ST --warmup 10 --texts 64 --sub= --no-cpu-baseline --no-e2e > gpurun_out/r3w4_$ST.json 2> gpurun_out/r3w4_$ST.err || { tail -5 gpurun_out/r3w4_$ST.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3w4_$ST.json')); print('steps $ST:', d['value'], 'GB/s', round(d['ms_per_step']*1000,2), 'us/step', d['parity'][:9], d['blocks_ms'], d['roofline_one_batch_in_flight']['kernel_us'], d['roofline_one_batch_in_flight']['pipeline_us'], d['roofline_one_group_in_flight']['kernel_us'], d['roofline_one_group_in_flight']['pipeline_us'])"
done
ACM_SIEVE_SKIP=ce timeout -k 10 300 python3 bench.py --steps 200 --warmup 10 --texts 64 --repeats 3 --sub= --no-extra --no-cpu-baseline --no-e2e --no-verify > gpurun_out/r3w4_skip.json 2> gpurun_out/r3w4_skip.err || { tail -5 gpurun_out/r3w4_skip.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r3w4_skip.json')); print('bulk only:', d['value'], 'GB/s')"
