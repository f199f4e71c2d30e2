#!/bin/bash
# helper waves / sub-row size sweep on real data (libraries built with -DACM_K... not needed: env knobs)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for H in 512 1024 2048; do for S in 128 256; do
  echo "== helpers $H subrow $S"
  ACM_SIEVE_HELPER_WAVES=$H ACM_SIEVE_SUBROW=$S timeout -k 10 200 python3 tools/real_data_probe.py 2000 15000 2>&1 | grep "real 0\|real 1\|synthetic" | cut -c1-60
done; done
