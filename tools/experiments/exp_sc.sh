timeout -k 10 900 python3 -m pytest tests/test_gpu_scan.py tests/test_gpu_compat_api.py tests/test_gpu_acm_grep.py -x -q -p no:cacheprovider 2>&1 | tail -2
for i in 1 2; do
python3 bench.py --workload sentiment --texts 4 --sub= --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('sentiment', d['value'], d['ms_per_step'], d['roofline_one_batch_in_flight']['pipeline_us'], d['parity'][:20])"
done
