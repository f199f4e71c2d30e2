timeout -k 10 600 python3 -m pytest tests/test_gpu_sparse.py tests/test_gpu_scan.py -x -q -p no:cacheprovider 2>&1 | tail -2
PROBE_STAMPS=1 timeout -k 10 500 python3 tools/real_data_probe.py 2000 15000 2>&1 | grep -E "sigs|stage-1"
bash tools/experiments/exp_ab.sh
