#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for NT in 0 1; do for ST in 200 20; do
  ACM_SIEVE_NT=$NT timeout -k 10 300 python3 bench.py --steps $ST --warmup 10 --texts 64 --sub= --no-cpu-baseline --no-e2e > gpurun_out/r3nt_$NT$ST.json 2> gpurun_out/r3nt_$NT$ST.err || { tail -5 gpurun_out/r3nt_$NT$ST.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3nt_$NT$ST.json')); print('nt $NT steps $ST:', d['value'], 'GB/s', round(d['ms_per_step']*1000,2), 'us/step', d['parity'][:9], d['blocks_ms'], d['roofline_one_batch_in_flight']['kernel_us'], d['roofline_one_batch_in_flight']['pipeline_us'], d['roofline_one_group_in_flight']['kernel_us'], d['roofline_one_group_in_flight']['pipeline_us'])"
done; done
for NT in 0 1; do
  ACM_SIEVE_NT=$NT ACM_SIEVE_SKIP=ce timeout -k 10 300 python3 bench.py --steps 200 --warmup 10 --texts 64 --repeats 3 --sub= --no-extra --no-cpu-baseline --no-e2e --no-verify > gpurun_out/r3nt_skip$NT.json 2> gpurun_out/r3nt_skip$NT.err || { tail -5 gpurun_out/r3nt_skip$NT.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3nt_skip$NT.json')); print('bulk only, nt $NT:', d['value'], 'GB/s')"
done
