timeout -k 10 600 python3 -m pytest tests/test_gpu_sparse.py tests/test_gpu_scan.py -x -q -p no:cacheprovider > gpurun_out/t_r.log 2>&1 || { tail -30 gpurun_out/t_r.log; exit 1; }
tail -1 gpurun_out/t_r.log
PROBE_STAMPS=1 timeout -k 10 500 python3 tools/real_data_probe.py 2000 15000 2>&1 | grep -E "sigs|stage-1"
python3 bench.py --sub= --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline_one_batch_in_flight']['kernel_us'], d['roofline_one_group_in_flight'], d['parity'][:30])"
