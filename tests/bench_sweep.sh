#!/bin/bash
# gpurun helper: bench.py over pipelines and batches in flight (no profiler, no CPU baseline)
# usage: bash tests/bench_sweep.sh <tag> "<mode:workers> ..."
TAG=${1:-sweep}
mkdir -p gpurun_out
: > gpurun_out/sweep_$TAG.jsonl
for mw in ${2:-chain:2 sparse:1 sparse:2 sparse:3 sparse:4}; do
	mode=${mw%%:*}; w=${mw##*:}; extra=""
	case $w in *g) w=${w%g}; extra="--graphs";; esac
	timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --mode $mode --workers $w --no-cpu-baseline $extra ${BENCH_ARGS} \
		>> gpurun_out/sweep_$TAG.jsonl 2> gpurun_out/sweep_$TAG.err || exit 1
done
python3 - <<PY
import json
for l in open("gpurun_out/sweep_$TAG.jsonl"):
    if not l.startswith("{"): continue
    j = json.loads(l)
    r = j["roofline"]; s = j.get("roofline_one_batch_in_flight", {})
    print(j["config"]["pipeline"], "graphs" if j["config"].get("hip_graphs") else "plain ", "W=%d" % j["config"]["batches_in_flight"], "%.1f GB/s" % j["value"], "us/step %.1f" % (j["ms_per_step"] * 1e3), "host %.1f" % j["host_enqueue_us_per_step"],
          "kernels", r["kernels_us"], "pipe %.1f" % r["pipeline_us"], "solo", s.get("kernels_us"), s.get("pipeline_us"), j["parity"][:9])
PY
