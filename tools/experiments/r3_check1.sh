#!/bin/bash
# sub-rows in k_sieve_check: the sparse pipeline's tests (real binary content included), the probe on real data, a short bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_sparse.py tests/test_gpu_scan.py -x -q -p no:cacheprovider > gpurun_out/r3c1_pytest.log 2>&1; tail -5 gpurun_out/r3c1_pytest.log
grep -q " passed" gpurun_out/r3c1_pytest.log || exit 1
grep -q "failed\|error" gpurun_out/r3c1_pytest.log && exit 1
timeout -k 10 300 python3 tools/real_data_probe.py 2000 15000 > gpurun_out/r3c1_probe.txt 2>&1 || { tail -5 gpurun_out/r3c1_probe.txt; exit 1; }
cat gpurun_out/r3c1_probe.txt
timeout -k 10 300 python3 bench.py --steps 200 --sub=clamav15000 --no-cpu-baseline --no-e2e > gpurun_out/r3c1_bench.json 2> gpurun_out/r3c1_bench.err || { tail -5 gpurun_out/r3c1_bench.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r3c1_bench.json')); print('headline', d['value'], d['parity'][:9], 'one batch', d['roofline_one_batch_in_flight']['pipeline_us'], 'group', d['roofline_one_group_in_flight']['pipeline_us'], '15000:', d['sub_records']['clamav15000']['value'])"
