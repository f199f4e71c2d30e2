timeout -k 10 900 python3 -m pytest tests/test_gpu_scan.py tests/test_gpu_sparse.py tests/test_gpu_compat_api.py tests/test_gpu_acm_grep.py -x -q -p no:cacheprovider > gpurun_out/t_c.log 2>&1 || { tail -30 gpurun_out/t_c.log; exit 1; }
tail -1 gpurun_out/t_c.log
python3 bench.py --workload sentiment --texts 4 --sub= --no-cpu-baseline --no-e2e 2>gpurun_out/bs.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('sentiment', d['value'], d['ms_per_step'], d['roofline_one_batch_in_flight'], d['parity'][:30])" || { tail -5 gpurun_out/bs.err; exit 1; }
grep -i "hot\|states" gpurun_out/bs.err | head -5
python3 bench.py --mode chain --sub= --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('clamav2000 chain', d['value'], d['ms_per_step'], d['roofline_one_batch_in_flight'], d['parity'][:30])"
