#!/bin/bash
# SQ counters of the chain pipeline's kernels on one workload (two passes)
TAG=${1:-r3sq}
WL=${2:-sentiment}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/${TAG}_$name -- python3 bench.py --workload $WL --sub "" --mode chain --workers 1 --steps 20 --warmup 2 --repeats 2 --texts 12 --group 1 --no-extra --no-cpu-baseline --no-e2e --no-verify > gpurun_out/${TAG}_$name.json 2> gpurun_out/${TAG}_$name.err || { tail -5 gpurun_out/${TAG}_$name.err; exit 1; }
}
run SQ1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY
run SQ2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU
python3 tests/pmc_summarize.py gpurun_out/${TAG} gpurun_out/${TAG}_traffic_$WL.json
python3 - <<PY
import json
d=json.load(open('gpurun_out/${TAG}_traffic_$WL.json'))
for k,v in d.items():
    if k.startswith('k_'):
        print(k, {a: round(b) for a,b in v.items() if a.startswith('SQ')})
PY
