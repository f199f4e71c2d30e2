B="python3 bench.py --sub= --no-cpu-baseline --no-e2e"
for st in 20 50 200; do for iss in threads native; do
  timeout -k 10 200 $B --steps $st --issue $iss 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps $st $iss', d['value'], d['ms_per_step'], d['blocks_ms'], d['host_enqueue_us_per_step'], d['parity'][:20])" || exit 1
done; done
