#!/bin/bash
# scatter of one stream's batches in the shadow of the next stream's walk: parity, then the 200-step bench per scatter workgroup size
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_gpu_ldswalk.py tests/test_gpu_carry.py -x -q -p no:cacheprovider > gpurun_out/r3o_pytest.log 2>&1; tail -3 gpurun_out/r3o_pytest.log
grep -q "passed" gpurun_out/r3o_pytest.log || exit 1
grep -q "failed\|error" gpurun_out/r3o_pytest.log && exit 1
for SB in 0 512; do for W in 1 2 3; do
  ACM_LDS_SCATTER_BLOCK=$SB timeout -k 10 300 python3 bench.py --workload sentiment --steps 192 --texts 64 --workers $W --sub= --no-extra --no-cpu-baseline --no-e2e > gpurun_out/r3o_s${SB}w$W.json 2> gpurun_out/r3o_s${SB}w$W.err || { tail -5 gpurun_out/r3o_s${SB}w$W.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3o_s${SB}w$W.json')); print('scatter block $SB workers $W:', d['value'], 'GB/s', d['ms_per_step']*1000, 'us/step', d['parity'][:9], d['roofline']['kernel'], d['roofline']['kernel_us'], d['roofline']['batches_per_launch'])"
done; done
