"""Host half of the product (libacmatch.so, no GPU needed): pattern loader and
automaton builder against the oracle and the golden vectors.

Mirrors what a reference test of acsmx.c would check: state count, max
pattern length, every defined cell of the serialised table (acsmx.c:640-658),
the pattern a final state reports, the patterns table chains.
"""
import os

import numpy as np
import pytest

import fixtures
import orc
from gpu_pattern_matching_amd import AcmError, Automaton


def product_for(name):
    path, hx, max_len = fixtures.set_source(name)
    a = Automaton()
    n = a.load_file(path, hx, max_len)
    a.compile()
    return a, n


@pytest.mark.parametrize("name", ["tests", "tests1", "tests2", "tests3", "sentiment", "clamav2000",
                                  "clamav2000_m12"])
def test_table_equals_oracle_and_golden(lib, golden, name):
    a, n = product_for(name)
    o = fixtures.oracle_for(name)
    g = golden["sets"][name]
    assert n == g["patterns"] == a.num_patterns
    assert a.num_states == g["states"]
    assert a.max_pattern_len == g["max_pattern_len"]
    t = a.reference_table()
    assert np.array_equal(t, o.table())
    if "table_digest" in g:
        assert "%016x" % orc.table_digest(t) == g["table_digest"]
    for i in range(0, n, max(1, n // 200)):
        b, iid, _ = a.pattern(i)
        assert (b, iid) == o.pattern(i)
    assert np.array_equal(np.array([a.pattern(i)[2] for i in range(n)]), o.patterns_chain())
    for s in range(0, a.num_states, max(1, a.num_states // 500)):
        assert a.state_output(s) == o.head_index(s)


def test_big_set_counts_only(lib, golden):
    a, n = product_for("clamav15000_m12")
    g = golden["sets"]["clamav15000_m12"]
    assert (n, a.num_states, a.max_pattern_len) == (g["patterns"], g["states"], g["max_pattern_len"])


def _write(tmp_path, name, text):
    p = tmp_path / name
    p.write_bytes(text)
    return str(p)


def test_loader_plain_and_limit(lib, tmp_path):
    p = _write(tmp_path, "p.txt", b"alpha\nbeta\n\"quoted\"\nlast")
    a = Automaton()
    assert a.load_file(p) == 4
    assert [a.pattern(i)[:2] for i in range(4)] == [(b"alpha", 0), (b"beta", 1), (b"quoted", 2),
                                                    (b"last", 3)]
    a = Automaton()
    a.load_file(p, max_len=3)
    assert [a.pattern(i)[0] for i in range(4)] == [b"alp", b"bet", b"quo", b"las"]


def test_loader_categorical(lib, tmp_path):
    p = _write(tmp_path, "c.txt", b"-1 \"died\"\n-2 \"death\"\n+7\tseven\n12 twelve words\n")
    a = Automaton()
    assert a.load_file(p) == 4
    assert [a.pattern(i)[:2] for i in range(4)] == [(b"died", -1), (b"death", -2), (b"seven", 7),
                                                    (b"twelve words", 12)]
    # same file through the oracle's restatement of ocl_worker.c:74-145
    o = orc.Oracle()
    o.load(p)
    assert [o.pattern(i) for i in range(4)] == [a.pattern(i)[:2] for i in range(4)]
    # a first line that is not "ID pattern" switches the whole file to plain mode
    p2 = _write(tmp_path, "n.txt", b"word 1\n2 two\n")
    a = Automaton()
    a.load_file(p2)
    assert [a.pattern(i)[:2] for i in range(2)] == [(b"word 1", 0), (b"2 two", 1)]


def test_loader_hex(lib, tmp_path):
    p = _write(tmp_path, "h.txt", b"00ff10\ndeadBEEF\n0a0b0c0d0e\n")
    a = Automaton()
    assert a.load_file(p, hex=True) == 3
    assert [a.pattern(i)[0] for i in range(3)] == [b"\x00\xff\x10", b"\xde\xad\xbe\xef",
                                                   b"\x0a\x0b\x0c\x0d\x0e"]
    a = Automaton()
    a.load_file(p, hex=True, max_len=2)
    assert [a.pattern(i)[0] for i in range(3)] == [b"\x00\xff", b"\xde\xad", b"\x0a\x0b"]
    assert a.max_pattern_len == 2


def test_loader_errors(lib, tmp_path):
    a = Automaton()
    with pytest.raises(AcmError) as e:
        a.load_file(str(tmp_path / "missing.txt"))
    assert e.value.code == -6
    with pytest.raises(AcmError) as e:
        Automaton().load_file(_write(tmp_path, "odd.txt", b"abc\n"), hex=True)
    assert e.value.code == -7   # the reference prints "ERROR: reading pattern!" and exits
    with pytest.raises(AcmError) as e:
        Automaton().load_file(_write(tmp_path, "bad.txt", b"zz\n"), hex=True)
    assert e.value.code == -7


def test_add_after_compile_is_refused(lib):
    a = Automaton()
    a.add(b"abc", 1)
    a.compile()
    with pytest.raises(AcmError):
        a.add(b"x", 2)


def test_byte_classes(lib):
    """Bytes no pattern uses share class 0, every used byte has a class of its own; a set that uses
    (nearly) every byte value is left alone."""
    a = Automaton()
    for i, w in enumerate([b"died", b"death", b"lost", b"zebra"]):
        a.add(w, i + 1)
    a.compile()
    n, m = a.byte_classes()
    used = sorted(set(b"dieddeathlostzebra"))
    assert n == len(used) + 1
    assert all(m[b] == 0 for b in range(256) if b not in used)
    assert sorted(int(m[b]) for b in used) == list(range(1, len(used) + 1))
    a.close()
    a = Automaton()
    a.add(bytes(range(256)), 1)
    a.compile()
    n, m = a.byte_classes()
    assert n == 256 and list(m) == list(range(256))
    a.close()
    a = Automaton()
    path, hx = orc.pattern_set("sentiment")
    a.load_file(path, hx, -1)
    a.compile()
    assert a.byte_classes()[0] == 27              # a-z and "anything else"
    a.close()


def test_lifo_numbering_and_outputs(lib):
    """the LAST pattern gets states 1..n; duplicates and suffixes decide
    which index a final state reports (head of the match list)."""
    a = Automaton()
    for i, p in enumerate([b"abcd", b"cd", b"d", b"cd", b"xabcd"]):
        a.add(p, 100 + i)
    a.compile()
    o = orc.Oracle()
    for i, p in enumerate([b"abcd", b"cd", b"d", b"cd", b"xabcd"]):
        o.add(p, 100 + i)
    o.compile()
    t = a.reference_table()
    assert np.array_equal(t, o.table())
    assert abs(int(t[0, 0, ord("x")])) == 1          # last pattern's first state is 1
    text = b"zzxabcdzzcdzd"
    pos, pat, _ = o.scan(text)
    # match lists (head first), by hand from acsmx.c:299-312 and :417-429:
    #   "d"     [2]
    #   "cd"    reverse([2]) ++ own [1, 3]              = [2, 1, 3]     -> reports 2
    #   "abcd"  reverse([2, 1, 3]) ++ [0]               = [3, 1, 2, 0]  -> reports 3
    #   "xabcd" reverse([3, 1, 2, 0]) ++ [4]            = [0, 2, 1, 3, 4] -> reports 0
    assert pos.tolist() == [6, 10, 12]
    assert pat.tolist() == [0, 2, 2]
    ref_state_of = {}
    for s in range(a.num_states):
        ref_state_of[s] = a.state_output(s)
    assert sorted(v for v in ref_state_of.values() if v >= 0) == sorted([2, 2, 3, 0])


def test_empty_pattern_never_reports(lib):
    a = Automaton()
    a.add(b"", 5)
    a.add(b"ab", 6)
    a.compile()
    t = a.reference_table()
    o = orc.Oracle()
    o.add(b"", 5)
    o.add(b"ab", 6)
    o.compile()
    assert np.array_equal(t, o.table())
    # transitions into state 0 are never negative (acsmx.c:645: -0 == 0) ...
    assert (t[:, 0, :] == 0).any() and int(t[0, 0, ord("x")]) == 0
    # ... but depth >= 2 states whose fail state is the root inherit the empty pattern and
    # become final (acsmx.c:417-429), depth-1 states do not (:376-382): "ab" reports index 0
    pos, pat, _ = o.scan(b"xxabxx")
    assert pos.tolist() == [3] and pat.tolist() == [0]


@pytest.mark.parametrize("name", ["tests", "tests1", "sentiment", "clamav2000_m12"])
def test_state_match_lists_equal_oracle(lib, name):
    """acm_automaton_state_matches (all-patterns reporting) = the oracle's match lists, which
    test_oracle_equals_compiled_reference pins against acsmx.c; the head is what the table reports."""
    o = fixtures.oracle_for(name)
    path, hx, ml = fixtures.set_source(name)
    a = Automaton()
    a.load_file(path, hx, ml)
    a.compile()
    assert a.num_states == o.num_states
    multi = 0
    for s in range(0, o.num_states, max(1, o.num_states // 5000)):
        exp = o.match_list(s) if s else []          # the root is never reported (transitions into 0 are not final)
        got = a.state_matches(s)
        assert got == exp, s
        assert a.state_output(s) == (exp[0] if exp else -1)
        multi += len(exp) > 1
    if name == "sentiment":
        assert multi > 0                            # nested patterns: lists longer than the head
    a.close()
