#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for HW in 2048 4096 8192; do for SR in 256 128; do
  echo "== helper waves $HW, sub-row $SR"
  ACM_SIEVE_HELPER_WAVES=$HW ACM_SIEVE_SUBROW=$SR timeout -k 10 300 python3 tools/real_data_probe.py 2000 15000 2>&1 | grep sigs | awk '{print $1, $3, $4, $6}' | tr '\n' ';'; echo
done; done
