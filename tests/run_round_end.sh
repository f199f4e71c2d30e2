#!/bin/bash
# gpurun helper: everything the round-end record needs, in one call (about ten minutes).
#   GPU tests -> the driver's bench command (the line of record) + the default one -> the same under
#   rocprofv3 --kernel-trace --stats (all workers; ONE launch group in flight: --workers 1; sentiment;
#   ClamAV on the chain pipeline) -> PMC passes -> real-binary table -> acm_grep end to end
TAG=${1:-r3}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/pytest_$TAG.log 2>&1 || { tail -20 gpurun_out/pytest_$TAG.log; exit 1; }
tail -2 gpurun_out/pytest_$TAG.log
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_$TAG.json 2> gpurun_out/bench_driver_$TAG.err || { tail -5 gpurun_out/bench_driver_$TAG.err; exit 1; }
cut -c1-400 gpurun_out/bench_driver_$TAG.json
timeout -k 10 500 python3 bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || { tail -5 gpurun_out/bench_$TAG.err; exit 1; }
cut -c1-400 gpurun_out/bench_$TAG.json
prof() {   # name, bench arguments...
  local name=$1; shift
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${name}_$TAG -- python3 bench.py --no-cpu-baseline --no-e2e --no-extra --sub= "$@" > gpurun_out/bench_prof_${name}_$TAG.json 2> gpurun_out/bench_prof_${name}_$TAG.err || { tail -5 gpurun_out/bench_prof_${name}_$TAG.err; exit 1; }
  cp $(find gpurun_out/prof_${name}_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/kernel_stats_${name}_$TAG.csv
  rm -rf gpurun_out/prof_${name}_$TAG
  cut -c1-150 gpurun_out/kernel_stats_${name}_$TAG.csv | head -8
}
prof headline || exit 1
prof driver --steps 20 --warmup 5 || exit 1
prof one_group --workers 1 || exit 1                       # one launch group in flight: the kernels' exclusive times
prof one_batch --workers 1 --group 1 --steps 40 || exit 1   # one batch in flight
prof sentiment --workload sentiment || exit 1
prof sentiment_one_group --workload sentiment --workers 1 || exit 1
prof clamav_chain --mode chain --workers 1 --steps 40 || exit 1
bash tests/run_pmc.sh pmc_$TAG clamav2000 || exit 1
bash tests/run_pmc.sh pmcs_$TAG sentiment || exit 1
bash tests/run_pmc.sh pmcc_$TAG clamav2000 chain || exit 1
timeout -k 10 300 python3 tools/real_data_probe.py 2000 15000 > gpurun_out/real_data_probe_$TAG.txt 2>&1 || { tail -5 gpurun_out/real_data_probe_$TAG.txt; exit 1; }
grep sigs gpurun_out/real_data_probe_$TAG.txt
bash tests/e2e_cli.sh > gpurun_out/e2e_$TAG.txt 2>&1 || exit 1
cat gpurun_out/e2e_$TAG.txt
