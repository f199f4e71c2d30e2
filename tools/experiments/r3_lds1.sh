#!/bin/bash
# round 3: first run of the LDS walk: parity tests, then exclusive kernel times on the sentiment workload
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_gpu_ldswalk.py -x -q -p no:cacheprovider > gpurun_out/r3l1_pytest.log 2>&1; tail -15 gpurun_out/r3l1_pytest.log
grep -q "passed" gpurun_out/r3l1_pytest.log || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3l1_prof -- python3 bench.py --workload sentiment --mode chain --workers 1 --group 1 --steps 40 --warmup 4 --repeats 2 --texts 8 --sub= --no-extra --no-cpu-baseline --no-e2e > gpurun_out/r3l1_bench.json 2> gpurun_out/r3l1_bench.err || { tail -5 gpurun_out/r3l1_bench.err; exit 1; }
cut -c1-300 gpurun_out/r3l1_bench.json
cat $(find gpurun_out/r3l1_prof -name "*kernel_stats.csv" | head -1) | cut -c1-160 | head -6
timeout -k 10 300 python3 bench.py --workload sentiment --steps 200 --texts 64 --sub= --no-extra --no-cpu-baseline --no-e2e > gpurun_out/r3l1_bench200.json 2> gpurun_out/r3l1_bench200.err || { tail -5 gpurun_out/r3l1_bench200.err; exit 1; }
cut -c1-300 gpurun_out/r3l1_bench200.json
