"""Debugging aid: the scan pipelines on real binary content (the middle of the largest ROCm library on
the box) against the synthetic bench text, one 32 MiB batch in flight, in microseconds.
python tools/real_data_probe.py [sigs ...]"""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import synth
from gpu_pattern_matching_amd import Automaton, DeviceArray, Matcher

N = 32 << 20
libs = sorted(glob.glob("/opt/rocm/lib/*.so*"), key=lambda f: os.path.getsize(f) if os.path.isfile(f) else 0)
size = os.path.getsize(libs[-1])
with open(libs[-1], "rb") as fh:
    fh.seek((size // 2) & ~4095)
    blob = np.frombuffer(fh.read(4 * N), dtype=np.uint8)
pieces = [blob[i * N:(i + 1) * N] for i in range(blob.size // N)]
print("data:", os.path.basename(libs[-1]), "%d pieces of 32 MiB; zero bytes: %s" % (
    len(pieces), ", ".join("%.0f %%" % (100.0 * float((p == 0).mean())) for p in pieces)))


def timed(m, d, n, reps=8):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    m.scan_async(d, n)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        ev[0].record()
        m.scan_async(d, n)
        ev[1].record()
        torch.cuda.synchronize()
        best = min(best, ev[0].elapsed_time(ev[1]) * 1e3)
    return best


for sigs in [int(x) for x in sys.argv[1:]] or [2000, 15000]:
    pats = synth.load_hex_patterns(os.path.join(ROOT, "tests", "data", "clamav", "15000.txt"), sigs)
    a = Automaton()
    for i, p in enumerate(pats):
        a.add(p, i)
    a.compile()
    m = Matcher(a, 0, max_text=N, plane_capacity=1 << 22)
    texts = [("synthetic", synth.clamav_corpus(N, 7, pats, 4096))] + [("real %d" % i, p) for i, p in enumerate(pieces)]
    for name, t in texts:
        d = DeviceArray.from_numpy(np.ascontiguousarray(t))
        row = []
        for mode in ("sparse", "chain"):
            m.set_mode(mode)
            us = timed(m, d, t.size)
            pos, _, _ = m.fetch()
            row.append("%s %7.1f us" % (mode, us))
            if mode == "sparse" and os.environ.get("PROBE_STAMPS"):
                os.environ["ACM_SIEVE_STAMPS"] = "1"      # read at enqueue: prints where the waves spent their time
                m.scan_async(d, t.size)
                m.fetch()
                del os.environ["ACM_SIEVE_STAMPS"]
        print("%5d sigs  %-10s %s   records %d" % (sigs, name, "   ".join(row), pos.size))
    m.close()
    a.close()
