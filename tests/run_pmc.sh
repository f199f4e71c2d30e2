#!/bin/bash
# gpurun helper: counters for the bench command, one group per pass (FETCH_SIZE needs 3 of the 4 TCC
# slots, WRITE_SIZE 2 -- MI355X_MICROARCH.md, rocprofv3 PMC slots; never with a trace domain other than
# --kernel-trace).  One batch per launch (--group 1), 40 steps over 16 distinct texts = 512 MiB, twice the
# Infinity Cache: what is counted is HBM traffic.  usage: run_pmc.sh TAG WORKLOAD [MODE]
TAG=${1:-pmc}
WL=${2:-clamav2000}
MODE=${3:-auto}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {   # name, counters...
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/${TAG}_$name -- python3 bench.py --workload $WL --sub "" --mode $MODE --steps 40 --warmup 4 --repeats 2 --texts 16 --group 1 --no-extra --no-cpu-baseline --no-e2e --no-verify > gpurun_out/${TAG}_$name.json 2> gpurun_out/${TAG}_$name.err || { tail -5 gpurun_out/${TAG}_$name.err; exit 1; }
}
run FETCH_SIZE FETCH_SIZE
run WRITE_SIZE WRITE_SIZE
run SQ1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY
run SQ2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
python3 tests/pmc_summarize.py gpurun_out/${TAG} gpurun_out/${TAG}_traffic_${WL}$([ $MODE = auto ] || echo _$MODE).json
rm -rf gpurun_out/${TAG}_FETCH_SIZE gpurun_out/${TAG}_WRITE_SIZE gpurun_out/${TAG}_SQ1 gpurun_out/${TAG}_SQ2   # (the raw per-dispatch tables: tens of MB)
