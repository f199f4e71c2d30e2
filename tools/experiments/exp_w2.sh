B="python3 bench.py --sub= --no-cpu-baseline --no-e2e --no-verify"
for round in 1 2; do for w in 3 4; do for st in 20 50 200; do
  timeout -k 10 200 $B --steps $st --workers $w 2>gpurun_out/bg.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('workers $w steps $st', d['value'], d['ms_per_step'])" || { tail -5 gpurun_out/bg.err; exit 1; }
done; done; done
