B="python3 bench.py --sub= --no-cpu-baseline --no-e2e --no-verify"
cp gpu_pattern_matching_amd/libacmatch.so /tmp/cur.so
for round in 1 2; do for v in prev cur; do
  if [ $v = prev ]; then cp tools/_libs/libacmatch_prev.so gpu_pattern_matching_amd/libacmatch.so; else cp /tmp/cur.so gpu_pattern_matching_amd/libacmatch.so; fi
  for st in 200 20; do
  timeout -k 10 200 $B --steps $st 2>gpurun_out/bg.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v steps $st', d['value'], d['ms_per_step'], d['roofline_one_batch_in_flight']['pipeline_us'], d['roofline_one_group_in_flight']['pipeline_us'])" || { tail -5 gpurun_out/bg.err; exit 1; }
  done
done; done
cp /tmp/cur.so gpu_pattern_matching_amd/libacmatch.so
