B="python3 bench.py --workload sentiment --texts 4 --sub= --no-cpu-baseline --no-e2e"
for e in 0 1 0; do
  if [ $e = 1 ]; then export ACM_SCAN_NO_PRELOAD=1; else unset ACM_SCAN_NO_PRELOAD; fi
  timeout -k 10 300 $B 2>gpurun_out/bg.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('no_preload=$e', d['value'], d['ms_per_step'], d['roofline_one_batch_in_flight']['kernel_us'], d['roofline_one_batch_in_flight']['pipeline_us'], d['parity'][:20])" || { tail -5 gpurun_out/bg.err; exit 1; }
done
unset ACM_SCAN_NO_PRELOAD
for w in 2 3; do
  timeout -k 10 300 $B --workers $w 2>gpurun_out/bg.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('preload workers $w', d['value'], d['ms_per_step'])" || { tail -5 gpurun_out/bg.err; exit 1; }
done
