#!/bin/bash
# the chain-pipeline part of tests/run_round_end.sh (after a change to the cold-plane kernels only)
TAG=r3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_scan.py tests/test_gpu_carry.py tests/test_gpu_ldswalk.py tests/test_gpu_acm_grep.py tests/test_gpu_dropin_cli.py tests/test_gpu_compat_api.py -x -q -p no:cacheprovider 2>&1 | tail -2
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_clamav_chain_$TAG -- python3 bench.py --no-cpu-baseline --no-e2e --no-extra --sub= --mode chain --workers 1 --steps 40 > gpurun_out/bench_prof_clamav_chain_$TAG.json 2> gpurun_out/bench_prof_clamav_chain_$TAG.err || exit 1
cp $(find gpurun_out/prof_clamav_chain_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/kernel_stats_clamav_chain_$TAG.csv
rm -rf gpurun_out/prof_clamav_chain_$TAG
cut -c1-150 gpurun_out/kernel_stats_clamav_chain_$TAG.csv | head -6
bash tests/run_pmc.sh pmcc_$TAG clamav2000 chain || exit 1
timeout -k 10 300 python3 tools/real_data_probe.py 2000 15000 > gpurun_out/real_data_probe_$TAG.txt 2>&1 || exit 1
grep sigs gpurun_out/real_data_probe_$TAG.txt
