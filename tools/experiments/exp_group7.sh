for st in 20 200; do
timeout -k 10 300 python3 bench.py --sub= --no-cpu-baseline --no-e2e --steps $st 2>gpurun_out/bg.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('default steps $st', d['value'], d['ms_per_step'], d['blocks_ms'], d['config']['workers'], d['config']['batches_per_launch_group'], d['roofline'], d.get('roofline_one_group_in_flight'))" || { tail -5 gpurun_out/bg.err; exit 1; }
done
timeout -k 10 300 python3 bench.py --workload sentiment --sub= --no-cpu-baseline --no-e2e --texts 4 2>gpurun_out/bg.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('sentiment', d['value'], d['ms_per_step'], d['config']['workers'])" || { tail -5 gpurun_out/bg.err; exit 1; }
timeout -k 10 300 python3 bench.py --workload clamav15000 --sub= --no-cpu-baseline --no-e2e --texts 4 2>gpurun_out/bg.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('15k', d['value'], d['ms_per_step'], d['config']['workers'])" || { tail -5 gpurun_out/bg.err; exit 1; }
timeout -k 10 600 python3 -m pytest tests/test_gpu_multi.py -x -q -p no:cacheprovider 2>&1 | tail -2
