#!/bin/bash
# end-to-end (file -> pinned host -> PCIe -> scan -> results back) throughput of acm_grep
mkdir -p gpurun_out
python3 - <<'PY'
import numpy as np, os, sys
sys.path.insert(0, "tests")
import synth
pats = synth.load_hex_patterns("tests/data/clamav/15000.txt", 2000)
os.makedirs("/tmp/acm_e2e", exist_ok=True)
for i in range(4):
    synth.clamav_corpus(256 << 20, 100 + i, pats, 32768).tofile("/tmp/acm_e2e/f%d.bin" % i)
open("/tmp/acm_e2e_sigs.txt", "w").writelines(open("tests/data/clamav/15000.txt").readlines()[:2000])
PY
for w in 1 2 4; do
  echo "== acm_grep -w $w (4 x 256 MiB files, 32 MiB buffers)"
  timeout -k 10 300 gpu_pattern_matching_amd/acm_grep -f /tmp/acm_e2e -p /tmp/acm_e2e_sigs.txt -x -B 4096 -D 0 -G 8192 -L 1024 -w $w | tee gpurun_out/e2e_w$w.log | grep -E "Matches:|Time|Processed bytes|Kernel launches|Throughput"
done
if [ -x oracle/_ref/ocl_aho_grep_acm ]; then
  echo "== reference CLI binary on libacmatch.so, -w 4"
  timeout -k 10 300 oracle/_ref/ocl_aho_grep_acm -f /tmp/acm_e2e -p /tmp/acm_e2e_sigs.txt -x -B 4096 -D 0 -G 8192 -L 1024 -w 4 | grep -E "Matches:|Time|Throughput"
fi
rm -rf /tmp/acm_e2e /tmp/acm_e2e_sigs.txt
