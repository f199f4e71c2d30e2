timeout -k 10 600 python3 -m pytest tests/test_gpu_sparse.py tests/test_gpu_scan.py -x -q -p no:cacheprovider > gpurun_out/t_g.log 2>&1 || { tail -30 gpurun_out/t_g.log; exit 1; }
tail -1 gpurun_out/t_g.log
B="python3 bench.py --sub= --no-cpu-baseline --no-e2e"
for g in 1 2 4; do for w in 2 3 4; do
  timeout -k 10 200 $B --group $g --workers $w 2>gpurun_out/bg.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('group $g workers $w', d['value'], d['ms_per_step'], d['host_enqueue_us_per_step'], d['parity'][:20])" || { tail -5 gpurun_out/bg.err; exit 1; }
done; done
timeout -k 10 200 $B --group 4 --workers 4 --steps 20 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('group 4 steps 20', d['value'], d['ms_per_step'], d['blocks_ms'])"
timeout -k 10 200 $B --group 4 --workers 2 --steps 20 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('group 4 w2 steps 20', d['value'], d['ms_per_step'], d['blocks_ms'])"
