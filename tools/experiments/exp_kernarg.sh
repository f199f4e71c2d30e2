B="python3 bench.py --sub= --no-cpu-baseline --no-e2e --no-verify"
for e in 0 1; do for cfg in "1 4" "4 4"; do set -- $cfg
  HIP_FORCE_DEV_KERNARG=$e timeout -k 10 200 $B --group $1 --workers $2 2>gpurun_out/bg.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('devkernarg $e group $1 workers $2', d['value'], d['ms_per_step'], d['roofline_one_batch_in_flight']['kernel_us'], d['roofline_one_batch_in_flight']['pipeline_us'])" || { tail -5 gpurun_out/bg.err; exit 1; }
done; done
