"""The oracle (oracle/acref.c) against what pins it.

* the known-answer vectors of SURVEY.md App. D (derived from the reference's
  own table for the reference's own tests/ fixtures),
* tests/golden/golden.json: digests produced by the reference's acsmx.c
  compiled unmodified (oracle/_ref) on every fixture pattern set,
* the compiled reference itself, cell for cell, where oracle/_ref is present.
"""
import hashlib
import json
import os

import numpy as np
import pytest

import fixtures
import orc

KAT_TESTS = [(85, 0), (591, 1), (911, 2), (1165, 3), (1359, 4), (1745, 5), (2431, 6), (2532, 8),
             (2666, 7), (2946, 8), (3012, 8), (3026, 9), (3234, 10), (3872, 11), (4757, 12),
             (5196, 13), (5751, 14), (6131, 15), (6263, 16), (6788, 17), (6870, 18), (7310, 19),
             (8330, 20), (8956, 21)]
KAT_TESTS_STATES = [197, 189, 176, 169, 156, 151, 138, 129, 134, 129, 129, 125, 118, 103, 98, 90,
                    85, 81, 66, 57, 42, 29, 14, 9]
KAT_TESTS1_OFFSETS = [236, 391, 876, 1135, 2306, 2547, 2931, 3094, 3189, 3356, 4394, 4693, 4916,
                      5043, 5158, 5219, 5505, 5561, 5808, 6453, 7467, 7789, 7900, 8407, 8702]
KAT_TESTS1_STATES = [244, 231, 218, 213, 203, 192, 187, 181, 171, 162, 153, 144, 132, 123, 110,
                     99, 90, 86, 75, 61, 49, 40, 28, 15, 4]
ROOT_ROW_TESTS = {"a": 86, "c": 104, "d": 43, "e": 152, "f": 91, "g": 15, "h": 126, "k": 10,
                  "l": 170, "n": 30, "o": 99, "q": 119, "t": 157, "u": 1, "v": 67, "w": 190}

with open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")) as _f:
    GOLDEN = json.load(_f)["sets"]

SMALL_SETS = ["tests", "tests1", "tests2", "tests3", "sentiment", "clamav2000", "clamav2000_m12"]
BIG_SETS = ["clamav10000", "clamav15000", "clamav15000_m12"]


def test_kat_tests_input():
    o = fixtures.oracle_for("tests")
    text = np.fromfile(os.path.join(orc.DATA, "ref_tests", "input.txt"), dtype=np.uint8)
    pos, pat, fs = o.scan(text)
    assert list(zip(pos.tolist(), pat.tolist())) == KAT_TESTS
    assert fs == 0
    assert o.num_states == 198 and o.max_pattern_len == 15
    # matched reference state ids and the root row (state numbering check)
    t = o.table()
    state, hit_states = 0, []
    for b in text:
        nxt = int(t[state, 0, b])
        if nxt < 0:
            hit_states.append(-nxt)
            nxt = -nxt
        state = nxt
    assert hit_states == KAT_TESTS_STATES
    row = {chr(c): abs(int(t[0, 0, c])) for c in range(256) if t[0, 0, c] != 0}
    assert row == ROOT_ROW_TESTS


def test_kat_tests1_input():
    o = fixtures.oracle_for("tests1")
    text = np.fromfile(os.path.join(orc.DATA, "ref_tests", "1", "input.txt"), dtype=np.uint8)
    pos, pat, fs = o.scan(text)
    assert pos.tolist() == KAT_TESTS1_OFFSETS
    assert pat.tolist() == list(range(25))
    assert o.num_states == 245 and o.max_pattern_len == 14
    t = o.table()
    state, hit_states = 0, []
    for b in text:
        nxt = int(t[state, 0, b])
        if nxt < 0:
            hit_states.append(-nxt)
            nxt = -nxt
        state = nxt
    assert hit_states == KAT_TESTS1_STATES


def test_kat_tests3():
    o = fixtures.oracle_for("tests3")
    assert o.num_states == 8
    text = np.fromfile(os.path.join(orc.DATA, "ref_tests", "3", "lala2_uncat.txt"), dtype=np.uint8)
    assert o.scan(text)[0].size == 0


@pytest.mark.parametrize("name", SMALL_SETS + BIG_SETS)
def test_oracle_matches_golden(name):
    g = GOLDEN[name]
    o = fixtures.oracle_for(name)
    assert o.num_patterns == g["patterns"]
    assert o.num_states == g["states"]
    assert o.max_pattern_len == g["max_pattern_len"]
    if "table_digest" in g:
        assert "%016x" % o.table_digest() == g["table_digest"]
    pats = fixtures.patterns_of(name)
    for spec in g["texts"]:
        text = fixtures.text_for(spec, pats)
        assert hashlib.sha256(text.tobytes()).hexdigest() == spec["sha256"], "corpus generator drifted"
        pos, pat, fs = o.scan(text)
        assert pos.size == spec["count"]
        assert "%016x" % orc.records_digest(pos, pat) == spec["records_digest"]
        assert fs == spec["final_state"]
        assert [[int(a), int(b)] for a, b in zip(pos[:8], pat[:8])] == spec["first_records"]


@pytest.mark.parametrize("name", ["tests", "tests1", "tests3", "sentiment", "clamav2000_m12"])
def test_oracle_equals_compiled_reference(name):
    """Cell-for-cell against the reference's acsmx.c (only where oracle/_ref exists)."""
    if orc.rlib() is None:
        pytest.skip("oracle/_ref not available (needs /root/reference to build)")
    o = fixtures.oracle_for(name)
    r = orc.RefAcsmx()
    for b, iid in o.patterns():
        r.add(b, iid)
    r.compile()
    assert r.num_states == o.num_states
    assert r.max_pattern_len == o.max_pattern_len
    assert np.array_equal(r.table(), o.table())
    for s in range(0, o.num_states, max(1, o.num_states // 2000)):
        assert r.match_list(s) == o.match_list(s)
    rng = np.random.default_rng(5)
    pats = fixtures.patterns_of(name)
    text = fixtures.text_for({"kind": "clamav", "n": 1 << 16, "seed": 5, "n_plant": 64}, pats)
    init = int(rng.integers(0, o.num_states))
    a = r.scan(text, init)
    b = o.scan(text, init)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    # all-patterns walk: every pattern of the reference's own match list of
    # the state entered at each hit, walked over the reference's own table
    table = r.table().reshape(-1, 512)
    small = text[:1 << 14]
    exp_pos, exp_pat, state = [], [], init
    for k, c in enumerate(small):
        nxt = int(table[state, c])
        if nxt < 0:
            nxt = -nxt
            for p in r.match_list(nxt):
                exp_pos.append(k)
                exp_pat.append(p)
        state = nxt
    got = o.scan_all(small, init)
    assert got[0].tolist() == exp_pos and got[1].tolist() == exp_pat and got[2] == state


def test_reference_hex_decoder_agrees():
    if orc.rlib() is None:
        pytest.skip("oracle/_ref not available")
    r = orc.RefAcsmx()
    for h in ["00ff10", "deadBEEF", "0a0B0c0D", "7f"]:
        assert r.hex_to_bytes(h) == bytes.fromhex(h)


def test_threads_scan_equals_serial():
    o = fixtures.oracle_for("clamav2000_m12")
    pats = fixtures.patterns_of("clamav2000_m12")
    text = fixtures.text_for({"kind": "clamav", "n": 1 << 20, "seed": 21, "n_plant": 512}, pats)
    s = o.scan(text, 0)
    for nt in (1, 2, 3, 8):
        t = o.scan_threads(text, nt, 0)
        assert np.array_equal(s[0], t[0]) and np.array_equal(s[1], t[1]) and s[2] == t[2]


def test_reference_kernel_semantics_differ_from_serial():
    """ahomatch.cl's chunk restart + overflow continuation is not a serial scan.

    Emulated on tests/input.txt with 16-byte chunks: 25 reported matches vs 24 serial.
    The product targets the serial answer; this test documents the difference.
    """
    o = fixtures.oracle_for("tests")
    text = np.fromfile(os.path.join(orc.DATA, "ref_tests", "input.txt"), dtype=np.uint8)
    B = 16
    chunks = (text.size + B - 1) // B
    data = np.zeros(chunks * B, dtype=np.uint8)
    data[: text.size] = text
    indices = np.arange(chunks, dtype=np.int32) * B
    sizes = np.full(chunks, B, dtype=np.int32)
    sizes[-1] = text.size - (chunks - 1) * B
    res, res2 = orc.refkernel_scan(o.table().reshape(-1), data, indices, sizes, 0, o.max_pattern_len, 16)
    assert int(res[:chunks].sum()) == 25
    assert o.scan(text)[0].size == 24
