"""First-contact GPU probe (run by hand through gpurun, not collected by pytest).

Scans a handful of texts with the HIP pipeline, compares with the oracle and
prints timings.  Writes progress lines so a hang is visible.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import orc  # noqa: E402
import synth  # noqa: E402
from gpu_pattern_matching_amd import Automaton, DeviceArray, Matcher, api  # noqa: E402


def log(*a):
    print(*a, flush=True)


def check(name, m, o, text, init=0, chain=None):
    if chain is not None:
        m.set_chain_bytes(chain)
    pos, pat, last = m.scan(text, init)
    epos, epat, elast = o.scan(text, init)
    ok = np.array_equal(pos, epos) and np.array_equal(pat, epat) and last == elast
    log("%-28s n=%-9d S=%-4s matches=%-7d/%-7d last=%d/%d %s" % (
        name, len(text), chain, len(pos), len(epos), last, elast, "OK" if ok else "MISMATCH"))
    if not ok:
        k = 0
        while k < min(len(pos), len(epos)) and pos[k] == epos[k] and pat[k] == epat[k]:
            k += 1
        log("   first difference at record", k, "gpu", list(zip(pos[k:k + 4], pat[k:k + 4])),
            "oracle", list(zip(epos[k:k + 4], epat[k:k + 4])))
    return ok


def main():
    log(api.device_info(0))
    allok = True
    # ---- config 1 plumbing
    p, hx = orc.pattern_set("tests")
    o = orc.Oracle(); o.load(p, hx); o.compile()
    a = Automaton(); a.load_file(p, hx); a.compile()
    m = Matcher(a, 0, max_text=1 << 16)
    text = np.fromfile(os.path.join(orc.DATA, "ref_tests", "input.txt"), dtype=np.uint8)
    for S in (None, 16, 32, 64, 128, 256):
        allok &= check("tests/input.txt", m, o, text, 0, S)
    allok &= check("empty", m, o, np.zeros(0, np.uint8), 0, None)
    allok &= check("one byte", m, o, text[:1], 0, None)
    allok &= check("17 bytes", m, o, text[80:97], 0, 16)
    allok &= check("init state carry", m, o, text[83:400], int(o.scan(text[:83])[2]), 16)
    m.close()

    # ---- sentiment-like dense matches
    p, hx = orc.pattern_set("sentiment")
    o = orc.Oracle(); o.load(p, hx); o.compile()
    a = Automaton(); a.load_file(p, hx); a.compile()
    words = open(os.path.join(orc.DATA, "sentiment", "top5000_words.txt")).read().split()
    text = synth.word_corpus(1 << 20, 11, words)
    m = Matcher(a, 0, max_text=1 << 20)
    log("sentiment hot rows", m.hot_rows, "states", a.num_states)
    for S in (None, 16, 64, 256):
        allok &= check("sentiment 1MiB", m, o, text, 0, S)
    m.close()

    # ---- clamav 2000 full
    tmp = os.environ.get("TMPDIR", "/tmp")
    path = orc.clamav_file(2000, tmp)
    o = orc.Oracle(); o.load(path, True); o.compile()
    a = Automaton(); a.load_file(path, True); a.compile()
    pats = [o.pattern(i)[0] for i in range(o.num_patterns)]
    n = 32 << 20
    text = synth.clamav_corpus(n, 7, pats, 4096)
    log("building matcher", a.num_states, "states")
    m = Matcher(a, 0, max_text=n)
    log("hot rows", m.hot_rows, "device MB", m.device_bytes / 1e6)
    timing_only = os.environ.get("PROBE_TIMING_ONLY") == "1"
    if not timing_only:
        for S in (None, 32, 64, 128, 256):
            allok &= check("clamav2000 32MiB", m, o, text, 0, S)
        # a text made of one signature repeated: deep states everywhere
        rep = np.frombuffer((pats[5] * (1 + (1 << 20) // len(pats[5])))[: 1 << 20], dtype=np.uint8)
        allok &= check("repeated signature", m, o, rep, 0, 64)
        allok &= check("zeros", m, o, np.zeros(1 << 20, np.uint8), 0, 64)

    # ---- timing
    d_text = DeviceArray.from_numpy(text)
    lib = m.lib
    for S in [int(x) for x in os.environ.get("PROBE_S", "32,64,128,256").split(",")]:
        m.set_chain_bytes(S)
        for _ in range(3):
            m.scan_async(d_text, n)
        lib.acm_rt_device_sync()
        m.profile(True)
        t0 = time.perf_counter()
        K = 20
        for _ in range(K):
            m.scan_async(d_text, n)
        lib.acm_rt_device_sync()
        t1 = time.perf_counter()
        w, _, pl, cnt = m.profile_read()
        m.profile(False)
        log("S=%-3d wall %.1f us/scan  walk %.1f us  pipeline %.1f us  -> %.1f GB/s (walk %.1f GB/s)" % (
            S, (t1 - t0) / K * 1e6, w / cnt * 1e3, pl / cnt * 1e3, n / ((t1 - t0) / K) / 1e9,
            n / (w / cnt * 1e-3) / 1e9))
    m.close()
    log("ALL OK" if allok else "SOME MISMATCH")
    return 0 if allok else 1


if __name__ == "__main__":
    sys.exit(main())
