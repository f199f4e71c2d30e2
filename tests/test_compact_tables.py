"""The LDS-resident form of a small automaton (csrc/compact_tables.cpp) -- host side, no GPU:
every (state, byte) transition of the 8-byte records + full rows equals the dense DFA's, on the
reference's own small-alphabet fixtures; sets that cannot qualify are turned down."""
import ctypes as C
import os

import numpy as np
import pytest

import orc
import synth
from gpu_pattern_matching_amd import Automaton, _lib

STATS = ("states", "classes", "rows", "side", "image_bytes", "promoted", "plain", "with_overrides", "deferring")


def selftest(a, lds_bytes=0):
    lib = _lib.load()
    st = (C.c_uint32 * 9)()
    rc = lib.acm_compact_selftest(a.h, lds_bytes, st)
    assert rc >= 0, lib.acm_last_error()
    return rc, dict(zip(STATS, list(st)))


def automaton_of(path=None, patterns=None):
    a = Automaton()
    if path:
        a.load_file(path, False, -1)
    else:
        for i, p in enumerate(patterns):
            a.add(p, i)
    a.compile()
    return a


@pytest.mark.parametrize("rel", ["sentiment/patterns_categorical.txt", "ref_tests/patterns.txt",
                                 "ref_tests/1/patterns.txt", "ref_tests/3/patterns.txt"])
def test_every_transition_matches_the_dense_dfa(rel):
    a = automaton_of(os.path.join(orc.DATA, rel))
    rc, st = selftest(a)
    assert rc == 1, st
    assert st["states"] == a.num_states and st["image_bytes"] <= 160 * 1024 - 512
    assert st["rows"] >= 1 and st["plain"] + st["with_overrides"] + st["deferring"] + st["rows"] == st["states"]
    a.close()


def test_smaller_budgets_still_exact_or_refused():
    """Whatever the LDS budget, the tables are either exact or the set is turned down."""
    a = automaton_of(os.path.join(orc.DATA, "sentiment", "patterns_categorical.txt"))
    full = selftest(a)[1]
    seen_refusal = False
    for budget in (full["image_bytes"], 150 * 1024, 140 * 1024, 130 * 1024, 64 * 1024):
        rc, st = selftest(a, budget)
        assert rc in (0, 1)
        if rc == 1:
            assert st["image_bytes"] <= budget and st["rows"] <= full["rows"]
        seen_refusal |= rc == 0
    assert seen_refusal       # 15704 records of 8 bytes alone are 123 KiB
    a.close()


def test_sets_that_do_not_qualify():
    pats = synth.load_hex_patterns(os.path.join(orc.DATA, "clamav", "15000.txt"), 300)
    a = automaton_of(patterns=pats)                      # 256 byte classes
    assert selftest(a)[0] == 0
    a.close()
    rng = np.random.default_rng(5)                        # small alphabet, too many states
    many = [bytes(rng.integers(97, 101, size=12, dtype=np.uint8)) for _ in range(3000)]
    a = automaton_of(patterns=many)
    assert a.num_states > 16384 and selftest(a)[0] == 0
    a.close()


def test_nested_and_duplicate_patterns():
    pats = [b"a", b"ab", b"abc", b"abcd", b"bcd", b"cd", b"d", b"abcd", b"dab", b"dabc", b"aa", b"aaa", b"aaaa"]
    a = automaton_of(patterns=pats)
    rc, st = selftest(a)
    assert rc == 1 and st["classes"] == 5
    a.close()


def test_profile_counts_steps():
    lib = _lib.load()
    a = automaton_of(os.path.join(orc.DATA, "sentiment", "patterns_categorical.txt"))
    words = open(os.path.join(orc.DATA, "sentiment", "top5000_words.txt")).read().split()
    text = synth.word_corpus(1 << 18, 3, words)
    cnt = (C.c_uint64 * 5)()
    assert lib.acm_compact_profile(a.h, text.ctypes.data, text.size, cnt) == 1
    steps, direct, one, more, finals = list(cnt)
    assert steps == text.size and direct + one + more == steps
    o = orc.Oracle()
    o.load(os.path.join(orc.DATA, "sentiment", "patterns_categorical.txt"))
    o.compile()
    assert finals == o.scan(text)[0].size       # a record per final state entered
    assert (one + more) < 0.02 * steps          # deferring to the fail state's record is the exception
    o.close()
    a.close()
