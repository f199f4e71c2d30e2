timeout -k 10 900 python3 -m pytest tests/test_gpu_scan.py tests/test_gpu_sparse.py tests/test_gpu_compat_api.py tests/test_gpu_acm_grep.py tests/test_gpu_dropin_cli.py -x -q -p no:cacheprovider > gpurun_out/t_p.log 2>&1 || { tail -30 gpurun_out/t_p.log; exit 1; }
tail -1 gpurun_out/t_p.log
B="python3 bench.py --workload sentiment --texts 4 --sub= --no-cpu-baseline --no-e2e"
for e in 1 0 1 0; do
  if [ $e = 1 ]; then export ACM_SCAN_NO_PRELOAD=1; else unset ACM_SCAN_NO_PRELOAD; fi
  timeout -k 10 300 $B 2>gpurun_out/bg.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('no_preload=$e', d['value'], d['ms_per_step'], d['roofline_one_batch_in_flight'], d['parity'][:20])" || { tail -5 gpurun_out/bg.err; exit 1; }
done
unset ACM_SCAN_NO_PRELOAD
bash tests/run_pmc.sh pmcp sentiment | grep -E "k_spec|k_halo|k_scatter" | cut -c1-120
