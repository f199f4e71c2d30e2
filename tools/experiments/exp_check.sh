timeout -k 10 600 python3 -m pytest tests/test_gpu_sparse.py tests/test_gpu_scan.py -x -q -p no:cacheprovider > gpurun_out/t_b.log 2>&1 || { tail -30 gpurun_out/t_b.log; exit 1; }
tail -1 gpurun_out/t_b.log
python3 tools/sieve_probe.py 2000 32 2> gpurun_out/probe_b.log
tail -32 gpurun_out/probe_b.log | grep -E "check. (follow|start|end)|emit. (start|end)|sieve. end"
python3 bench.py --sub= --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline_one_batch_in_flight'], d['parity'][:30])"
python3 bench.py --sub= --no-cpu-baseline --no-e2e --steps 20 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
python3 bench.py --workload clamav15000 --texts 4 --sub= --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('15k', d['value'], d['ms_per_step'], d['roofline_one_batch_in_flight'], d['parity'][:30])"
