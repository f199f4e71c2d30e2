#!/bin/bash
# gpurun helper: everything the round-end record needs, in one call.
#   GPU tests -> default bench.py (the line of record) -> the same command under
#   rocprofv3 --kernel-trace --stats -> PMC passes -> acm_grep end to end
TAG=${1:-r2}
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/pytest_$TAG.log 2>&1 || { tail -20 gpurun_out/pytest_$TAG.log; exit 1; }
tail -2 gpurun_out/pytest_$TAG.log
timeout -k 10 500 python3 bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || { tail -5 gpurun_out/bench_$TAG.err; exit 1; }
cut -c1-600 gpurun_out/bench_$TAG.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --no-cpu-baseline --no-e2e --no-extra --sub= > gpurun_out/bench_prof_$TAG.json 2> gpurun_out/bench_prof_$TAG.err || exit 1
cat $(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1) | cut -c1-150 | head -14
# ... and of the sentiment workload (chain pipeline) the same way
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profs_$TAG -- python3 bench.py --workload sentiment --texts 4 --no-cpu-baseline --no-e2e --no-extra --sub= > gpurun_out/bench_profs_$TAG.json 2> gpurun_out/bench_profs_$TAG.err || exit 1
cat $(find gpurun_out/profs_$TAG -name "*kernel_stats.csv" | head -1) | cut -c1-150 | head -8
bash tests/run_pmc.sh pmc_$TAG clamav2000 || exit 1
bash tests/run_pmc.sh pmcs_$TAG sentiment || exit 1
bash tests/e2e_cli.sh > gpurun_out/e2e_$TAG.txt 2>&1 || exit 1
cat gpurun_out/e2e_$TAG.txt
