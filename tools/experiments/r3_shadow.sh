#!/bin/bash
# sparse pipeline: what runs in the shadow of the bulk kernel (one or two bulk workgroups per CU)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python3 -m pytest tests/test_gpu_sparse.py -x -q -p no:cacheprovider > gpurun_out/r3s_pytest.log 2>&1; tail -3 gpurun_out/r3s_pytest.log
grep -q "passed" gpurun_out/r3s_pytest.log || exit 1
grep -q "failed\|error" gpurun_out/r3s_pytest.log && exit 1
for KB in ${KBS:-56 84}; do for ST in 200 20; do
  ACM_SIEVE_LDS_KB=$KB timeout -k 10 300 python3 bench.py --steps $ST --texts 64 --sub= --no-extra --no-cpu-baseline --no-e2e > gpurun_out/r3s_k${KB}s$ST.json 2> gpurun_out/r3s_k${KB}s$ST.err || { tail -5 gpurun_out/r3s_k${KB}s$ST.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3s_k${KB}s$ST.json')); print('lds $KB KiB steps $ST:', d['value'], 'GB/s', round(d['ms_per_step']*1000,2), 'us/step', d['parity'][:9], d['blocks_ms'])"
done; done
