#!/bin/bash
# two streams per worker (acm_scan_batch.finish_stream): parity, then the bench with and without
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python3 -m pytest tests/test_gpu_split.py -x -q -p no:cacheprovider > gpurun_out/r3sp_pytest.log 2>&1; tail -5 gpurun_out/r3sp_pytest.log
grep -q "passed" gpurun_out/r3sp_pytest.log || exit 1
grep -q "failed\|error" gpurun_out/r3sp_pytest.log && exit 1
for WL in clamav2000 sentiment; do for SP in "off 0" "on 2" "on 1" ; do set -- $SP; for ST in 200 20; do
  timeout -k 10 300 python3 bench.py --workload $WL --split $1 --workers $2 --steps $ST --warmup 10 --texts 64 --sub= --no-extra --no-cpu-baseline --no-e2e > gpurun_out/r3sp_${WL}_$1$2_$ST.json 2> gpurun_out/r3sp_${WL}_$1$2_$ST.err || { tail -5 gpurun_out/r3sp_${WL}_$1$2_$ST.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3sp_${WL}_$1$2_$ST.json')); print('$WL split $1 workers $2 steps $ST:', d['value'], 'GB/s', round(d['ms_per_step']*1000,2), 'us/step', d['parity'][:9], d['config']['workers'], d['blocks_ms'])"
done; done; done
