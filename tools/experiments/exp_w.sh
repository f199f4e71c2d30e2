B="python3 bench.py --sub= --no-cpu-baseline --no-e2e --no-verify"
for st in 200 20; do for w in 2 3 4; do
  timeout -k 10 200 $B --steps $st --workers $w 2>gpurun_out/bg.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps $st workers $w', d['value'], d['ms_per_step'])" || { tail -5 gpurun_out/bg.err; exit 1; }
done; done
for wl in clamav15000 sentiment; do for w in 2 3 4; do
  timeout -k 10 300 $B --workload $wl --texts 4 --workers $w 2>gpurun_out/bg.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl workers $w', d['value'], d['ms_per_step'])" || { tail -5 gpurun_out/bg.err; exit 1; }
done; done
