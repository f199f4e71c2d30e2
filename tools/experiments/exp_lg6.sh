ACM_SIEVE_STRIDE=4 timeout -k 10 600 python3 -m pytest tests/test_gpu_sparse.py tests/test_gpu_scan.py -x -q -p no:cacheprovider -k "not sample_heavy" > gpurun_out/t_l.log 2>&1 || { tail -30 gpurun_out/t_l.log; exit 1; }
tail -1 gpurun_out/t_l.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_sparse.py -x -q -p no:cacheprovider 2>&1 | tail -1
echo "== stride 4, 6-byte keys"
ACM_SIEVE_STRIDE=4 PROBE_STAMPS=1 timeout -k 10 500 python3 tools/real_data_probe.py 2000 15000 2>&1 | grep -E "sigs|stage-1"
echo "== bench stride 4 vs 8"
B="python3 bench.py --sub= --no-cpu-baseline --no-e2e"
ACM_SIEVE_STRIDE=4 timeout -k 10 200 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('W4', d['value'], d['ms_per_step'], d['roofline_one_batch_in_flight'], d['parity'][:20])"
timeout -k 10 200 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('W8', d['value'], d['ms_per_step'], d['roofline_one_batch_in_flight'], d['parity'][:20])"
