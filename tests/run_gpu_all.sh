#!/bin/bash
# gpurun helper: GPU tests, then bench under rocprofv3 --kernel-trace --stats.
# usage: bash tests/run_gpu_all.sh <tag> [pytest-timeout]
TAG=${1:-run}
mkdir -p gpurun_out
timeout -k 10 ${2:-900} python3 -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/pytest_$TAG.log 2>&1
rc=$?
tail -5 gpurun_out/pytest_$TAG.log
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 50 --warmup 5 ${BENCH_ARGS} > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
rc=$?
cat gpurun_out/bench_$TAG.json
cat $(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
exit $rc
