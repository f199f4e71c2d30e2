#!/bin/bash
# LDS walk: parity, exclusive kernel times, and the 200-step bench with launch groups on 1..3 workers
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_gpu_ldswalk.py -x -q -p no:cacheprovider > gpurun_out/r3l2_pytest.log 2>&1; tail -3 gpurun_out/r3l2_pytest.log
grep -q "passed" gpurun_out/r3l2_pytest.log || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3l2_prof -- python3 bench.py --workload sentiment --mode chain --workers 1 --group 1 --steps 40 --warmup 4 --repeats 2 --texts 8 --sub= --no-extra --no-cpu-baseline --no-e2e > gpurun_out/r3l2_bench.json 2> gpurun_out/r3l2_bench.err || { tail -5 gpurun_out/r3l2_bench.err; exit 1; }
cat $(find gpurun_out/r3l2_prof -name "*kernel_stats.csv" | head -1) | cut -c1-130 | grep k_lds
for W in 1 2 3; do for G in 4 16; do
  timeout -k 10 300 python3 bench.py --workload sentiment --steps 192 --texts 64 --workers $W --group $G --sub= --no-extra --no-cpu-baseline --no-e2e > gpurun_out/r3l2_w${W}g$G.json 2> gpurun_out/r3l2_w${W}g$G.err || { tail -5 gpurun_out/r3l2_w${W}g$G.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3l2_w${W}g$G.json')); print('workers $W group $G:', d['value'], 'GB/s', d['ms_per_step']*1000, 'us/step', d['parity'][:9], d['roofline']['kernel'], d['roofline']['kernel_us'], d['roofline']['batches_per_launch'])"
done; done
