B="python3 bench.py --sub= --no-cpu-baseline --no-e2e"
for st in 20 20 50 200; do
  timeout -k 10 200 $B --steps $st 2>gpurun_out/bg.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps $st', d['value'], d['ms_per_step'], d['blocks_ms'], d['roofline']['batches_per_launch'], d['roofline']['frac'], d['parity'][:30])" || { tail -5 gpurun_out/bg.err; exit 1; }
done
timeout -k 10 200 $B --steps 20 --issue threads 2>gpurun_out/bg.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('threads steps 20', d['value'], d['ms_per_step'], d['parity'][:30])" || { tail -5 gpurun_out/bg.err; exit 1; }
timeout -k 10 200 $B --steps 30 --issue main 2>gpurun_out/bg.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('main steps 30', d['value'], d['ms_per_step'], d['parity'][:30])" || { tail -5 gpurun_out/bg.err; exit 1; }
timeout -k 10 600 python3 -m pytest tests/test_gpu_multi.py -x -q -p no:cacheprovider 2>&1 | tail -2
