#!/bin/bash
# the whole GPU suite + the default bench line + the real-data table (round 3)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r3f}
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/${TAG}_pytest.log 2>&1; tail -4 gpurun_out/${TAG}_pytest.log
grep -q " passed" gpurun_out/${TAG}_pytest.log || exit 1
grep -q "failed\|error" gpurun_out/${TAG}_pytest.log && exit 1
timeout -k 10 300 python3 tools/real_data_probe.py 2000 15000 > gpurun_out/${TAG}_probe.txt 2>&1 || { tail -5 gpurun_out/${TAG}_probe.txt; exit 1; }
grep sigs gpurun_out/${TAG}_probe.txt
timeout -k 10 500 python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench.json')); r=d['roofline']; print('headline', d['value'], d['parity'][:9], 'roofline', r['kernel'], r['frac'], r['kernel_us'], r['batches_per_launch'], r['launches_timed']); print('one batch', d['roofline_one_batch_in_flight']); print('group', d['roofline_one_group_in_flight']); print('ungrouped', d.get('one_launch_set_per_step'))
for k,v in d['sub_records'].items(): print(k, v['value'], v['ms_per_step'], v['parity'][:9], v['roofline']['kernel'], v['roofline']['frac'])
print('e2e', d.get('e2e_with_h2d'), 'cpu', d.get('cpu_baseline',{}).get('value'))"
