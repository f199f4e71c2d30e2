cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_s
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_s -- python3 bench.py --workload sentiment --texts 4 --sub= --no-cpu-baseline --no-e2e --no-verify --workers 1 > gpurun_out/bench_prof_s.json 2> gpurun_out/bench_prof_s.err || exit 1
python3 -c "import json; d=json.loads(open('gpurun_out/bench_prof_s.json').read()); print(d['value'], d['ms_per_step'])"
cat $(find gpurun_out/prof_s -name "*kernel_stats.csv" | head -1) | cut -c1-170 | grep -E "k_|Name"
