# one small scan through the LDS walk's fast tiles, outside pytest (its capture swallows the runtime's messages)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import orc, synth
from gpu_pattern_matching_amd import Automaton, Matcher
SENT = os.path.join(orc.DATA, "sentiment", "patterns_categorical.txt")
a = Automaton(); a.load_file(SENT, False, -1); a.compile()
o = orc.Oracle(); o.load(SENT); o.compile()
words = open(os.path.join(orc.DATA, "sentiment", "top5000_words.txt")).read().split()
text = synth.word_corpus(max(int(x) for x in sys.argv[1:]), 21, words)
m = Matcher(a, 0, max_text=text.size)
m.set_mode("chain")
for n in (int(x) for x in sys.argv[1:]):
    print("n", n, flush=True)
    got = m.scan(text[:n]); exp = o.scan(text[:n])
    ok = np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1]) and got[2] == exp[2]
    print(" records", got[0].size, exp[0].size, "last", got[2], exp[2], "OK" if ok else "MISMATCH", flush=True)
    if not ok:
        bad = np.nonzero(got[0][:min(got[0].size, exp[0].size)] != exp[0][:min(got[0].size, exp[0].size)])[0]
        print(" first differing record", bad[:5], got[0][bad[:5]], exp[0][bad[:5]], flush=True)
