B="python3 bench.py --workload clamav15000 --sub= --no-cpu-baseline --no-e2e --texts 4"
for cfg in "X=1" "ACM_SIEVE_STRIDE=4" "ACM_BLOOM_LOG_WORDS=14" "ACM_SIEVE_STRIDE=4 ACM_BLOOM_LOG_WORDS=14"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 $B --workers 4 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('w4', d['value'], d['ms_per_step'], d['roofline']['kernel_us'], d['roofline']['rest_of_pipeline_us'], d['roofline_one_batch_in_flight'])" || exit 1
done
echo "== 2000 sigs workers sweep"
for w in 1 2 3 4; do
 timeout -k 10 200 python3 bench.py --sub= --no-cpu-baseline --no-e2e --workers $w 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('w$w', d['value'], d['ms_per_step'])" || exit 1
 timeout -k 10 200 $B --workers $w 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('15k w$w', d['value'], d['ms_per_step'])" || exit 1
done
