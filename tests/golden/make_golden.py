"""Generate tests/golden/golden.json from the REFERENCE's own acsmx.c.

Run in the build container (needs /root/reference; uses oracle/_ref, i.e. the
reference's acsmx.c compiled unmodified behind oracle/ref_harness.c):

    python tests/golden/make_golden.py

For every fixture pattern set the reference builds its DFA; we record state
count, max pattern length, a digest of the serialised table (defined cells
only) and, for seeded synthetic texts, the record stream the serial walk over
the reference's state_table produces (count, FNV-1a digest, final state,
first records).  The texts themselves are regenerated from seeds
(tests/synth.py); their sha256 is stored so generator drift is detected.

The reference ships no expected outputs: these vectors are what
pins the oracle and the HIP path.
"""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
sys.path.insert(0, TESTS)

import orc  # noqa: E402
import synth  # noqa: E402

FULL = 32 << 20


def ref_and_patterns(path, hex_, max_len):
    # parse with the oracle's loader (ocl_worker.c parser restatement), build with the reference
    o = orc.Oracle()
    o.load(path, hex_, max_len)
    pats = o.patterns()
    r = orc.RefAcsmx()
    for b, iid in pats:
        r.add(b, iid)
    r.compile()
    return r, pats


def text_for(spec, pats):
    kind = spec["kind"]
    if kind == "file":
        return np.fromfile(os.path.join(orc.DATA, spec["path"]), dtype=np.uint8)
    if kind == "clamav":
        return synth.clamav_corpus(spec["n"], spec["seed"], [p for p, _ in pats], spec["n_plant"])
    if kind == "words":
        words = open(os.path.join(orc.DATA, "sentiment", "top5000_words.txt")).read().split()
        return synth.word_corpus(spec["n"], spec["seed"], words)
    if kind == "repeat":
        p = pats[spec["pattern"]][0]
        return np.frombuffer((p * (spec["n"] // len(p) + 1))[: spec["n"]], dtype=np.uint8).copy()
    if kind == "zeros":
        return np.zeros(spec["n"], dtype=np.uint8)
    raise KeyError(kind)


def main():
    tmp = tempfile.mkdtemp()
    sets = [
        # name, path, hex, max_len, texts
        ("tests", orc.pattern_set("tests")[0], False, -1,
         [{"kind": "file", "path": "ref_tests/input.txt"}]),
        ("tests1", orc.pattern_set("tests1")[0], False, -1,
         [{"kind": "file", "path": "ref_tests/1/input.txt"}]),
        ("tests2", orc.pattern_set("tests2")[0], False, -1,
         [{"kind": "words", "n": 1 << 20, "seed": 3}]),
        ("tests3", orc.pattern_set("tests3")[0], False, -1,
         [{"kind": "file", "path": "ref_tests/3/lala2_uncat.txt"}]),
        ("sentiment", orc.pattern_set("sentiment")[0], False, -1,
         [{"kind": "words", "n": 1 << 20, "seed": 11},
          {"kind": "words", "n": FULL, "seed": 11}]),
        ("clamav2000", orc.clamav_file(2000, tmp), True, -1,
         [{"kind": "clamav", "n": 1 << 20, "seed": 7, "n_plant": 256},
          {"kind": "clamav", "n": FULL, "seed": 7, "n_plant": 4096},
          {"kind": "repeat", "n": 1 << 20, "pattern": 5},
          {"kind": "zeros", "n": 1 << 20}]),
        ("clamav2000_m12", orc.clamav_file(2000, tmp), True, 12,
         [{"kind": "clamav", "n": FULL, "seed": 7, "n_plant": 4096}]),
        ("clamav10000", orc.clamav_file(10000, tmp), True, -1,
         [{"kind": "clamav", "n": FULL, "seed": 9, "n_plant": 4096}]),
        ("clamav15000", orc.clamav_file(15000, tmp), True, -1,
         [{"kind": "clamav", "n": 1 << 20, "seed": 8, "n_plant": 256},
          {"kind": "clamav", "n": FULL, "seed": 8, "n_plant": 4096}]),
        ("clamav15000_m12", orc.clamav_file(15000, tmp), True, 12,
         [{"kind": "clamav", "n": FULL, "seed": 8, "n_plant": 4096}]),
    ]
    out = {"generator": "tests/golden/make_golden.py", "source": "reference acsmx.c via oracle/_ref",
           "sets": {}}
    for name, path, hex_, max_len, texts in sets:
        print("==", name, flush=True)
        r, pats = ref_and_patterns(path, hex_, max_len)
        entry = {
            "hex": hex_, "max_len": max_len,
            "patterns": len(pats),
            "states": r.num_states,
            "max_pattern_len": r.max_pattern_len,
            "texts": [],
        }
        if r.num_states <= 100000:
            entry["table_digest"] = "%016x" % orc.table_digest(r.table())
        for spec in texts:
            t = text_for(spec, pats)
            pos, pat, fs = r.scan(t)
            rec = dict(spec)
            rec.update({
                "sha256": hashlib.sha256(t.tobytes()).hexdigest(),
                "count": int(pos.size),
                "records_digest": "%016x" % orc.records_digest(pos, pat),
                "final_state": int(fs),
                "first_records": [[int(a), int(b)] for a, b in zip(pos[:8], pat[:8])],
            })
            entry["texts"].append(rec)
            print("   ", spec, "->", pos.size, "records, final", fs, flush=True)
        out["sets"][name] = entry
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote golden.json")


if __name__ == "__main__":
    main()
