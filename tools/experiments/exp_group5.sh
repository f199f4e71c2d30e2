timeout -k 10 600 python3 -m pytest tests/test_gpu_sparse.py -x -q -p no:cacheprovider > gpurun_out/t_g.log 2>&1 || { tail -30 gpurun_out/t_g.log; exit 1; }
tail -1 gpurun_out/t_g.log
B="python3 bench.py --sub= --no-cpu-baseline --no-e2e"
for cfg in "4 4 200" "8 4 200" "8 3 200" "8 2 200" "4 4 20" "8 4 20" "8 2 20" "5 4 20" "4 4 50" "8 4 50"; do set -- $cfg
  timeout -k 10 200 $B --group $1 --workers $2 --steps $3 2>gpurun_out/bg.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('group $1 workers $2 steps $3', d['value'], d['ms_per_step'], d['blocks_ms'], d['parity'][:9])" || { tail -5 gpurun_out/bg.err; exit 1; }
done
