# the documented debugging knobs still give the oracle's answers
for e in "ACM_SCAN_NO_PRELOAD=1" "ACM_SCAN_HALO=0" "ACM_SCAN_MODE=chain" "ACM_SCAN_GRAPHS=1"; do
  echo "== $e"
  env $e timeout -k 10 600 python3 -m pytest tests/test_gpu_scan.py tests/test_gpu_compat_api.py -x -q -p no:cacheprovider 2>&1 | tail -1
done
