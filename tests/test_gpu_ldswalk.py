"""The chain pipeline's LDS-resident walk (lds_walk.hip: the whole automaton in the CU's LDS as
8-byte records + full rows, halo chains, per-lane hit lists, ordered scatter) against the oracle:
same planes bit for bit -- sizes around every tile and chain border, carried-in states, shard
halos, launch groups, final-state reporting -- and against the cold-plane walk kernels it replaces."""
import os

import numpy as np
import pytest

import fixtures
import orc
import synth
from gpu_pattern_matching_amd import Automaton, DeviceArray, Matcher, _lib

pytestmark = pytest.mark.gpu

SENT = os.path.join(orc.DATA, "sentiment", "patterns_categorical.txt")


def assert_same(got, exp):
    assert got[0].size == exp[0].size, "record count %d != %d" % (got[0].size, exp[0].size)
    assert np.array_equal(got[0], exp[0]), "offsets differ"
    assert np.array_equal(got[1], exp[1]), "pattern indices differ"
    assert got[2] == exp[2], "final state %d != %d" % (got[2], exp[2])


def sentiment():
    a = Automaton()
    a.load_file(SENT, False, -1)
    a.compile()
    o = orc.Oracle()
    o.load(SENT)
    o.compile()
    return a, o


def words():
    return open(os.path.join(orc.DATA, "sentiment", "top5000_words.txt")).read().split()


def _planes(pat, off, cap, stream=None):
    p = pat.to_numpy(np.int32, cap, stream=stream)
    q = off.to_numpy(np.int32, cap, stream=stream)
    m = int(p[0])
    return q[1:1 + m].astype(np.uint32), p[1:1 + m].copy(), int(p[m + 1])


def test_sizes_around_every_border(gpu):
    a, o = sentiment()
    m = Matcher(a, 0, max_text=1 << 20)
    m.set_mode("chain")
    text = synth.word_corpus(1 << 20, 21, words())
    # chain = 64 bytes, wave tile = 8 KiB, workgroup round = 128 KiB
    for n in (1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 129, 8191, 8192, 8193, 8192 + 63, 16384, 131071, 131072, 131073,
              (1 << 20) - 1, 1 << 20):
        assert_same(m.scan(text[:n]), o.scan(text[:n]))
        assert m.path_taken(n) == "chain"
    m.close()
    a.close()


def test_carried_state_and_streaming(gpu):
    """A text cut into pieces of odd sizes, the state carried from piece to piece: same records as one scan."""
    a, o = sentiment()
    m = Matcher(a, 0, max_text=1 << 18)
    m.set_mode("chain")
    text = synth.word_corpus(300000, 22, words())
    state, at = 0, 0
    rng = np.random.default_rng(4)
    while at < text.size:
        k = int(rng.integers(1, 40000))
        piece = text[at:at + k]
        exp = o.scan(piece, state)
        assert_same(m.scan(piece, state), exp)
        state = exp[2]
        at += k
    assert state == o.scan(text)[2]
    m.close()
    a.close()


def test_shard_halo_and_offset_shift(gpu):
    """acm_scan_shard_async: records that end inside the halo are dropped, offsets are shifted."""
    a, o = sentiment()
    n = 200000
    text = synth.word_corpus(n, 23, words())
    m = Matcher(a, 0, max_text=n)
    m.set_mode("chain")
    d = DeviceArray.from_numpy(text)
    for halo, shift in ((16, 1000), (1, -1), (63, 0), (64, 5), (65, 7), (8192, 0), (8200, 123456)):
        m.scan_async(d, n, 0, halo=halo, offset_shift=shift)
        got = m.fetch()
        epos, epat, elast = o.scan(text)
        keep = epos >= halo
        assert_same(got, ((epos[keep].astype(np.int64) + shift).astype(np.uint32), epat[keep], elast))
    d.free()
    m.close()
    a.close()


@pytest.mark.parametrize("group", [1, 4, 16])
def test_launch_groups(gpu, group):
    """acm_scan_batches_async: consecutive batches of one size share the two launches; every batch's planes
    are those of scanning it alone (different texts, carried-in states, a shard halo, a state report)."""
    a, o = sentiment()
    n = (1 << 18) + 24
    m = Matcher(a, 0, max_text=n)
    m.set_mode("chain")
    assert m.set_max_group(group) == group
    nb = 19
    w = words()
    texts = [synth.word_corpus(n, 30 + k, w) for k in range(5)]
    inits = [0, o.scan(b"unhapp")[2], 0, o.scan(b"dis")[2], 0]
    ws_bytes = m.lib.acm_scan_workspace_bytes(m.dfa, n)
    cap = 1 << 16
    d_texts = [DeviceArray.from_numpy(t) for t in texts]
    wss = [DeviceArray(ws_bytes) for _ in range(nb)]
    planes = [(DeviceArray(cap * 4), DeviceArray(cap * 4)) for _ in range(nb)]
    batches = []
    for k in range(nb):
        batches.append(m.make_batch(d_texts[k % 5], n, m.stream, planes[k][0], planes[k][1], cap, (wss[k], ws_bytes),
                                    init_state=inits[k % 5], halo=100 if k == 3 else 0, offset_shift=7 if k == 3 else 0,
                                    report=_lib.REPORT_STATE if k == 6 else 0))
    m.enqueue_many(batches)
    for k in range(nb):
        got = _planes(planes[k][0], planes[k][1], cap, m.stream)
        epos, epat, elast = o.scan(texts[k % 5], init_state=inits[k % 5])
        if k == 3:
            keep = epos >= 100
            epos, epat = epos[keep] + 7, epat[keep]
        if k == 6:      # the final states themselves (reference numbering): their head pattern is what the others report
            assert np.array_equal(got[0], epos) and got[2] == elast
            heads = np.array([a.state_output(int(s)) for s in got[1]], dtype=np.int32)
            assert np.array_equal(heads, epat)
            continue
        assert_same(got, (epos, epat, elast))
    m.close()
    a.close()


def test_same_planes_as_the_cold_plane_walk(gpu, monkeypatch):
    """ACM_SCAN_NO_LDSWALK=1 brings back the row-in-LDS / cold-plane kernels of scan.hip: same planes."""
    a, o = sentiment()
    text = synth.word_corpus(1 << 19, 24, words())
    m = Matcher(a, 0, max_text=text.size)
    m.set_mode("chain")
    new = m.scan(text)
    m.close()
    monkeypatch.setenv("ACM_SCAN_NO_LDSWALK", "1")
    m2 = Matcher(a, 0, max_text=text.size)
    m2.set_mode("chain")
    old = m2.scan(text)
    m2.close()
    assert_same(new, old)
    assert_same(new, o.scan(text))
    a.close()


SMALL = {
    "nested": [b"a", b"ab", b"abc", b"abcd", b"bcd", b"cd", b"d", b"dab", b"aa", b"aaa", b"aaaa", b"abcabcabc"],
    "two_letters": [b"ab", b"ba", b"aab", b"bba", b"abab", b"bbbb"],
    "long_words": [b"internationalisation", b"nationalisation", b"nation", b"ion", b"isation", b"counterrevolutionaries",
                   b"revolution", b"evolution", b"volution", b"aries"],
}


@pytest.mark.parametrize("name", sorted(SMALL))
def test_small_sets_fuzz(gpu, name):
    pats = SMALL[name]
    a, o = Automaton(), orc.Oracle()
    for i, p in enumerate(pats):
        a.add(p, i + 1)
        o.add(p, i + 1)
    a.compile()
    o.compile()
    m = Matcher(a, 0, max_text=1 << 16)
    m.set_mode("chain")
    alphabet = np.frombuffer(b"".join(pats) + b" xz", dtype=np.uint8)
    rng = np.random.default_rng(len(name))
    for n in (1, 7, 64, 100, 4097, 50000):
        text = alphabet[rng.integers(0, alphabet.size, size=n)]
        assert_same(m.scan(text), o.scan(text))
    # all one byte: every position a hit, every chain's list full
    text = np.full(20000, ord("a"), dtype=np.uint8)
    assert_same(m.scan(text), o.scan(text))
    m.close()
    a.close()


def test_plane_capacity_overflow_is_reported_not_written(gpu):
    a, o = sentiment()
    text = synth.word_corpus(1 << 16, 25, words())
    exp = o.scan(text)
    assert exp[0].size > 600
    m = Matcher(a, 0, max_text=text.size, plane_capacity=512)
    m.set_mode("chain")
    d = DeviceArray.from_numpy(text)
    m.scan_async(d, text.size)
    head = m.pat_plane.to_numpy(np.int32, 512)
    offs = m.off_plane.to_numpy(np.int32, 512)
    assert int(head[0]) == exp[0].size                 # the count says what did not fit
    assert np.array_equal(offs[1:510].astype(np.uint32), exp[0][:509])
    d.free()
    m.close()
    a.close()
