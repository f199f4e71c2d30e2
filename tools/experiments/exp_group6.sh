B="python3 bench.py --sub= --no-cpu-baseline --no-e2e"
for cfg in "8 2 20 native" "8 4 20 native" "8 2 200 native" "8 4 200 native" "8 2 20 threads" "8 2 50 native" "8 1 20 native" "8 3 20 native"; do set -- $cfg
  timeout -k 10 200 $B --group $1 --workers $2 --steps $3 --issue $4 2>gpurun_out/bg.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('group $1 workers $2 steps $3 $4', d['value'], d['ms_per_step'], d['blocks_ms'], d['host_enqueue_us_per_step'], d['parity'][:9])" || { tail -5 gpurun_out/bg.err; exit 1; }
done
