B="python3 bench.py --sub= --no-cpu-baseline --no-e2e --no-extra"
for round in 1 2; do for g in 16 32; do for st in 200 20; do
  timeout -k 10 200 $B --steps $st --group $g 2>gpurun_out/bg.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('group $g steps $st', d['value'], d['ms_per_step'], d['parity'][:9])" || { tail -5 gpurun_out/bg.err; exit 1; }
done; done; done
