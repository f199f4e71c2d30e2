cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "1 4" "1 1" "4 4"; do set -- $cfg
rm -rf gpurun_out/prof_g
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_g -- python3 bench.py --sub= --no-cpu-baseline --no-e2e --no-verify --workers $1 --group $2 > gpurun_out/bench_prof_g.json 2> gpurun_out/bench_prof_g.err || exit 1
echo "== workers $1 group $2"; python3 -c "import json; d=json.loads(open('gpurun_out/bench_prof_g.json').read()); print(d['value'], d['ms_per_step'])"
cat $(find gpurun_out/prof_g -name "*kernel_stats.csv" | head -1) | cut -c1-170 | grep -E "k_sieve"
done
