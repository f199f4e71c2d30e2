B="python3 bench.py --sub= --no-cpu-baseline --no-e2e --no-verify"
for e in 0 1; do for st in 20 200; do
  if [ $e = 1 ]; then export ACM_SIEVE_TWO=1; else unset ACM_SIEVE_TWO; fi
  timeout -k 10 200 $B --steps $st 2>gpurun_out/bg.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('two=$e steps $st', d['value'], d['ms_per_step'], d['blocks_ms'])" || { tail -5 gpurun_out/bg.err; exit 1; }
done; done
export ACM_SIEVE_TWO=1
for w in 2 4; do
  timeout -k 10 200 $B --steps 200 --workers $w 2>gpurun_out/bg.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('two=1 workers $w', d['value'], d['ms_per_step'])" || { tail -5 gpurun_out/bg.err; exit 1; }
done
