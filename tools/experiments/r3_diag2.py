# which pipeline differs from the oracle on the dense two-letter text of test_dense_candidates[2-binary]
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import orc
from gpu_pattern_matching_amd import Automaton, Matcher
pats = [bytes([0, 0, 1]), bytes([1, 0, 0, 1]), bytes([255, 0, 255]), bytes([1, 1, 1, 1, 1, 1]), bytes(range(16)), bytes([0, 1, 0, 1, 0, 1, 0])]
a, o = Automaton(), orc.Oracle()
for i, p in enumerate(pats):
    a.add(p, i + 1); o.add(p, i + 1)
a.compile(); o.compile()
m = Matcher(a, 0, max_text=1 << 18)
rng = np.random.default_rng(2 * 7 + len("binary"))
letters = np.frombuffer(bytes([0, 1, 255, 2, 3]), dtype=np.uint8)
for n in (64, 65, 200, 4096, 100003, 1 << 18):
    text = letters[rng.integers(0, 2, size=n)]
    exp = o.scan(text)
    for mode in ("chain", "sparse"):
        m.set_mode(mode)
        got = m.scan(text)
        ok = got[0].size == exp[0].size and np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1]) and got[2] == exp[2]
        msg = ""
        if not ok:
            k = min(got[0].size, exp[0].size)
            bad = np.nonzero((got[0][:k] != exp[0][:k]) | (got[1][:k] != exp[1][:k]))[0]
            msg = " first bad record %s got %s exp %s" % (bad[:3], got[0][bad[:3]], exp[0][bad[:3]])
        print(n, mode, "records", got[0].size, exp[0].size, "last", got[2], exp[2], "OK" if ok else "MISMATCH" + msg, flush=True)
