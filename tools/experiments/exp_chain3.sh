for cb in 0 64 128; do
python3 bench.py --workload sentiment --texts 4 --sub= --no-cpu-baseline --no-e2e --chain-bytes $cb 2>gpurun_out/bs.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('sentiment chain-bytes $cb', d['value'], d['ms_per_step'], d['roofline_one_batch_in_flight'], d['parity'][:30])" || { tail -5 gpurun_out/bs.err; exit 1; }
done
for w in 2 3; do
python3 bench.py --workload sentiment --texts 4 --sub= --no-cpu-baseline --no-e2e --workers $w 2>gpurun_out/bs.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('sentiment workers $w', d['value'], d['ms_per_step'])" || { tail -5 gpurun_out/bs.err; exit 1; }
done
