cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_c
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c -- python3 bench.py --mode chain --sub= --no-cpu-baseline --no-e2e --no-verify --workers 1 > gpurun_out/bench_prof_c.json 2> gpurun_out/bench_prof_c.err || exit 1
python3 -c "import json; d=json.loads(open('gpurun_out/bench_prof_c.json').read()); print(d['value'], d['ms_per_step'])"
cat $(find gpurun_out/prof_c -name "*kernel_stats.csv" | head -1) | cut -c1-150 | grep -E "k_|Name"
