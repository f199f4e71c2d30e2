"""Debugging aid: scan the bench text a few times with ACM_SIEVE_STAMPS set; libacmatch prints
where the waves of k_sieve spend their time.  python tools/sieve_probe.py [sigs] [n_mib]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
if not os.environ.get("NO_STAMPS"):
    os.environ["ACM_SIEVE_STAMPS"] = "1"
import numpy as np
import synth
from gpu_pattern_matching_amd import Automaton, Matcher

sigs = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n = (int(sys.argv[2]) if len(sys.argv) > 2 else 32) << 20
pats = synth.load_hex_patterns(os.path.join(ROOT, "tests", "data", "clamav", "15000.txt"), sigs)
aut = Automaton()
for i, p in enumerate(pats):
    aut.add(p, i)
aut.compile()
text = synth.clamav_corpus(n, 7, pats, 4096)
m = Matcher(aut, 0, max_text=text.size)
m.set_mode("sparse")
from gpu_pattern_matching_amd import DeviceArray
d = DeviceArray.from_numpy(text)
for rep in range(12):
    print("---- scan", rep, file=sys.stderr)
    m.scan_async(d, text.size)
    pos, pat, last = m.fetch()
print("records", pos.size, "last", last, "path", m.path_taken(text.size))
m.close()
