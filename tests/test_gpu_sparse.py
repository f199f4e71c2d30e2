"""The sparse pipeline (strided 3-gram sieve -> exact prefix check -> trie-path followers ->
ordered emit, sparse.hip) against the oracle and against the chain pipeline: same planes, bit for
bit, whichever pipeline produced them -- including texts that are all matches or one endless
trie path, which the sparse pipeline handles itself (slowly) and AUTO mode learns to hand to the
chain pipeline.
"""
import os

import numpy as np
import pytest

import fixtures
import orc
import synth
from gpu_pattern_matching_amd import Automaton, DeviceArray, Matcher

pytestmark = pytest.mark.gpu


def build(patterns):
    a, o = Automaton(), orc.Oracle()
    for i, p in enumerate(patterns):
        a.add(p, i + 1)
        o.add(p, i + 1)
    a.compile()
    o.compile()
    return a, o


def assert_same(got, exp):
    assert got[0].size == exp[0].size, "record count %d != %d" % (got[0].size, exp[0].size)
    assert np.array_equal(got[0], exp[0]), "offsets differ"
    assert np.array_equal(got[1], exp[1]), "pattern indices differ"
    assert got[2] == exp[2], "final state %d != %d" % (got[2], exp[2])


def test_eligibility(gpu):
    """Only sets whose shortest pattern has 3 bytes qualify; the others stay on the chain path."""
    a, _ = build([b"abc", b"abcd", b"xyz12"])
    m = Matcher(a, 0, max_text=4096)
    assert m.sparse_eligible() and m.set_mode("sparse") == "sparse"
    m.close()
    a, o = build([b"abc", b"de"])
    m = Matcher(a, 0, max_text=4096)
    assert not m.sparse_eligible()
    m.set_mode("sparse")
    text = np.frombuffer(b"xxabcdexxdeabc" * 40, dtype=np.uint8)
    assert_same(m.scan(text), o.scan(text))
    assert m.path_taken(text.size) == "chain"
    m.close()


SMALL_SETS = {
    "overlap": [b"abc", b"bcd", b"cde", b"abcde", b"aaa", b"cab", b"bca", b"eeeee", b"dcba"],
    "nested": [b"abcabc", b"bcab", b"cabcab", b"abc", b"bbb", b"ccc", b"abcabcabcabc"],
    "binary": [bytes([0, 0, 1]), bytes([1, 0, 0, 1]), bytes([255, 0, 255]), bytes([1, 1, 1, 1, 1, 1]),
               bytes(range(16)), bytes([0, 1, 0, 1, 0, 1, 0])],
}


@pytest.mark.parametrize("setname", sorted(SMALL_SETS))
@pytest.mark.parametrize("alphabet", [2, 3, 5, 26])
def test_dense_candidates(gpu, setname, alphabet):
    """Small alphabets: nearly every position is a candidate, deep runs overlap and chain into
    each other, walkers start inside other walkers' runs -- the prefix-max rule decides who
    reports what."""
    pats = SMALL_SETS[setname]
    a, o = build(pats)
    m = Matcher(a, 0, max_text=1 << 18)
    rng = np.random.default_rng(alphabet * 7 + len(setname))
    letters = np.frombuffer(b"abcde" if setname != "binary" else bytes([0, 1, 255, 2, 3]),
                            dtype=np.uint8)
    for n in (64, 65, 200, 4096, 100003, 1 << 18):
        if alphabet <= 5:
            text = letters[rng.integers(0, min(alphabet, letters.size), size=n)]
        else:
            text = rng.integers(0, 256, size=n, dtype=np.uint8)
            for _ in range(n // 50):   # plant patterns into the noise
                p = np.frombuffer(pats[int(rng.integers(len(pats)))], dtype=np.uint8)
                at = int(rng.integers(0, max(1, n - p.size)))
                text[at:at + p.size] = p[:n - at]
        exp = o.scan(text)
        got = {}
        for mode in ("chain", "sparse"):
            m.set_mode(mode)
            got[mode] = m.scan(text)
            assert_same(got[mode], exp)
    m.close()


def test_sparse_path_is_the_one_that_ran(gpu):
    """The sparse kernels produce the planes themselves (there is no fallback behind them): on
    ordinary data, on a repeated signature, on zero pages, and on a text that is one endless trie
    path with a hit every byte or so (followers with more than four hits take the slow passes)."""
    name = "clamav2000"
    o = fixtures.oracle_for(name)
    pats = fixtures.patterns_of(name)
    path, hx, ml = fixtures.set_source(name)
    a = Automaton()
    a.load_file(path, hx, ml)
    a.compile()
    m = Matcher(a, 0, max_text=1 << 21)
    a.close()
    assert m.sparse_eligible()
    m.set_mode("sparse")
    text = fixtures.text_for({"kind": "clamav", "n": 1 << 21, "seed": 31, "n_plant": 2000}, pats)
    assert_same(m.scan(text), o.scan(text))
    assert m.path_taken(text.size) == "sparse"
    p = pats[5]
    rep = np.frombuffer((p * ((1 << 18) // len(p) + 1))[:1 << 18], dtype=np.uint8)
    assert_same(m.scan(rep), o.scan(rep))          # a follower per repetition
    assert m.path_taken(rep.size) == "sparse"
    zeros = np.zeros(1 << 18, dtype=np.uint8)
    assert_same(m.scan(zeros), o.scan(zeros))
    assert m.path_taken(zeros.size) == "sparse"
    m.close()
    a, o = build(SMALL_SETS["nested"])
    m = Matcher(a, 0, max_text=1 << 18)
    m.set_mode("sparse")
    endless = np.frombuffer(b"abc" * 80000, dtype=np.uint8)   # one endless path, a hit every byte or so
    assert_same(m.scan(endless), o.scan(endless))
    assert m.path_taken(endless.size) == "sparse"
    quiet = np.frombuffer(b"abd" * 80000, dtype=np.uint8)
    assert_same(m.scan(quiet), o.scan(quiet))
    assert m.path_taken(quiet.size) == "sparse"
    m.close()


def test_long_patterns_do_not_fall_back(gpu):
    """A planted 2000-byte signature is one long unary path: the walker fast-forwards through
    it, no cap is hit."""
    rng = np.random.default_rng(5)
    pats = [bytes(rng.integers(0, 256, size=k, dtype=np.uint8)) for k in (2000, 700, 64, 9, 3)]
    a, o = build(pats)
    m = Matcher(a, 0, max_text=1 << 20)
    text = rng.integers(0, 256, size=1 << 20, dtype=np.uint8)
    for i in range(200):
        p = np.frombuffer(pats[i % len(pats)], dtype=np.uint8)
        at = int(rng.integers(0, text.size - p.size))
        text[at:at + p.size] = p
    m.set_mode("sparse")
    assert_same(m.scan(text), o.scan(text))
    assert m.path_taken(text.size) == "sparse"
    m.close()


def test_carried_state_and_cuts(gpu):
    """Streaming contract in sparse mode: cuts inside matches, 1..3 bytes after a match start,
    carried last_state seeds the walker at position 0."""
    pats = SMALL_SETS["nested"] + [b"needle-in-haystack", b"haystack"]
    a, o = build(pats)
    m = Matcher(a, 0, max_text=1 << 16)
    m.set_mode("sparse")
    rng = np.random.default_rng(9)
    text = np.frombuffer(b"abc", dtype=np.uint8)[rng.integers(0, 3, size=40000)].copy()
    for at in range(500, 39000, 1500):
        text[at:at + 18] = np.frombuffer(b"needle-in-haystack", dtype=np.uint8)
    whole = o.scan(text)
    for piece in (64, 67, 100, 1000, 4097):
        pos_all, pat_all, state = [], [], 0
        for start in range(0, text.size, piece):
            seg = text[start:start + piece]
            pos, pat, state = m.scan(seg, state)
            pos_all.append(pos.astype(np.int64) + start)
            pat_all.append(pat)
        assert np.array_equal(np.concatenate(pos_all), whole[0].astype(np.int64)), piece
        assert np.array_equal(np.concatenate(pat_all), whole[1]), piece
        assert state == whole[2]
    m.close()


def test_shard_halo_in_sparse_mode(gpu):
    """acm_scan_shard_async: records ending in the halo are dropped, offsets shifted."""
    name = "clamav2000"
    o = fixtures.oracle_for(name)
    pats = fixtures.patterns_of(name)
    text = fixtures.text_for({"kind": "clamav", "n": 1 << 19, "seed": 77, "n_plant": 800}, pats)
    whole = o.scan(text)
    path, hx, ml = fixtures.set_source(name)
    a = Automaton()
    a.load_file(path, hx, ml)
    a.compile()
    m = Matcher(a, 0, max_text=text.size)
    halo = a.max_pattern_len - 1
    a.close()
    m.set_mode("sparse")
    parts = 4
    per = text.size // parts + 3
    pos_all, pat_all = [], []
    for r in range(parts):
        lo, hi = r * per, min(text.size, (r + 1) * per)
        h = min(halo, lo)
        d = DeviceArray.from_numpy(text[lo - h:hi])
        m.scan_async(d, hi - lo + h, halo=h, offset_shift=lo - h)
        pos, pat, _ = m.fetch()
        assert m.path_taken(hi - lo + h) == "sparse"
        d.free()
        pos_all.append(pos)
        pat_all.append(pat)
    assert np.array_equal(np.concatenate(pos_all), whole[0])
    assert np.array_equal(np.concatenate(pat_all), whole[1])
    m.close()


def test_beyond_the_fused_block_limit(gpu):
    """129 MiB: tiles of 32 KiB, a wave of the bulk pass takes several sub-blocks per tile; and the
    chain pipeline's separate scan launch over more than 8192 block totals."""
    name = "clamav2000"
    o = fixtures.oracle_for(name)
    pats = fixtures.patterns_of(name)
    n = (129 << 20) + 12345
    text = fixtures.text_for({"kind": "clamav", "n": n, "seed": 3, "n_plant": 20000}, pats)
    path, hx, ml = fixtures.set_source(name)
    a = Automaton()
    a.load_file(path, hx, ml)
    a.compile()
    m = Matcher(a, 0, max_text=n)
    a.close()
    m.set_mode("sparse")
    exp = o.scan(text)
    assert_same(m.scan(text), exp)
    assert m.path_taken(n) == "sparse"
    m.set_mode("chain")          # and the chain pipeline's separate scan launch over > 8192 block totals
    m.set_chain_bytes(32)
    assert_same(m.scan(text), exp)
    m.close()


def test_capacity_overflow_in_sparse_mode(gpu):
    """Planes smaller than the result: the count is still exact, the first capacity-2 records and
    the state in the last cell are there, fetch reports ACM_ERR_CAPACITY (same contract as the
    chain pipeline, compactarray.cl:49-55 cell layout)."""
    from gpu_pattern_matching_amd import AcmError
    name = "clamav2000"
    o = fixtures.oracle_for(name)
    pats = fixtures.patterns_of(name)
    text = fixtures.text_for({"kind": "clamav", "n": 1 << 20, "seed": 5, "n_plant": 1000}, pats)
    exp = o.scan(text)
    assert exp[0].size > 200
    path, hx, ml = fixtures.set_source(name)
    a = Automaton()
    a.load_file(path, hx, ml)
    a.compile()
    m = Matcher(a, 0, max_text=text.size, plane_capacity=100)
    a.close()
    m.set_mode("sparse")
    d = DeviceArray.from_numpy(text)
    m.scan_async(d, text.size)
    with pytest.raises(AcmError) as e:
        m.fetch()
    assert e.value.code == -8
    assert m.path_taken(text.size) == "sparse"
    pat = m.pat_plane.to_numpy(np.int32, 100)
    off = m.off_plane.to_numpy(np.int32, 100)
    assert pat[0] == exp[0].size
    assert np.array_equal(off[1:99], exp[0][:98].astype(np.int32))
    assert np.array_equal(pat[1:99], exp[1][:98])
    assert pat[99] == exp[2]
    d.free()
    m.close()


@pytest.mark.parametrize("seed", range(12))
def test_random_pattern_sets(gpu, seed):
    """Random sets (3..40 bytes, small alphabets so that prefixes, suffixes and whole patterns
    collide a lot; duplicates allowed) x random and pattern-laced texts, carried-in states: sparse
    = chain = oracle, head-only and all-patterns."""
    rng = np.random.default_rng(1000 + seed)
    sigma = int(rng.integers(2, 6))
    npat = int(rng.integers(1, 40))
    pats = []
    for _ in range(npat):
        ln = int(rng.integers(3, 41 if rng.random() < 0.3 else 9))
        if pats and rng.random() < 0.4:        # extend or cut an earlier one
            base = pats[int(rng.integers(len(pats)))]
            cut = int(rng.integers(1, len(base) + 1))
            body = (base[:cut] if rng.random() < 0.5 else base[-cut:]) + bytes(rng.integers(97, 97 + sigma, size=ln, dtype=np.uint8))
            body = body[:max(3, ln)] if len(body) >= 3 else body + b"aaa"
            pats.append(bytes(body))
        else:
            pats.append(bytes(rng.integers(97, 97 + sigma, size=ln, dtype=np.uint8)))
    a, o = build(pats)
    m = Matcher(a, 0, max_text=1 << 17, plane_capacity=1 << 18)
    assert m.sparse_eligible()
    for n in (64, 1000, 70001, 1 << 17):
        text = rng.integers(97, 97 + sigma + 1, size=n, dtype=np.uint8)   # one letter outside the alphabet
        for _ in range(n // 200):
            p = np.frombuffer(pats[int(rng.integers(len(pats)))], dtype=np.uint8)
            at = int(rng.integers(0, max(1, n - p.size)))
            text[at:at + p.size] = p[:n - at]
        init = int(rng.integers(0, o.num_states)) if n < 70001 else 0
        exp, exp_all = o.scan(text, init), o.scan_all(text, init)
        for mode in ("sparse", "chain"):
            m.set_mode(mode)
            assert_same(m.scan(text, init), exp)
            if n <= 1000:
                assert_same(m.scan_all(text, init, out_capacity=exp_all[0].size + 2), exp_all)
    m.close()
    a.close()
    o.close()


def test_gibibyte_buffer(gpu):
    """A single 1 GiB + 12345 byte buffer (the API takes up to 2 GiB - 17): offsets above 2^30,
    65 000+ blocks, signatures planted across every kind of internal boundary -- both pipelines
    against the oracle's serial scan."""
    name = "clamav2000"
    o = fixtures.oracle_for(name)
    pats = fixtures.patterns_of(name)
    n = (1 << 30) + 12345
    piece = fixtures.text_for({"kind": "clamav", "n": 1 << 26, "seed": 21, "n_plant": 5000}, pats)
    text = np.empty(n, dtype=np.uint8)
    for at in range(0, n, piece.size):          # 16 shifted copies of a 64 MiB piece + the tail
        m_ = min(piece.size, n - at)
        text[at:at + m_] = np.roll(piece, at >> 20)[:m_]
    sig = np.frombuffer(pats[11], dtype=np.uint8)
    for at in (n - sig.size, (1 << 30) - 7, (1 << 29) + (1 << 15) - 3, (1 << 28) - 1):
        text[at:at + sig.size] = sig
    exp = o.scan(text, cap=1 << 21)
    assert exp[0].size > 70000 and int(exp[0][-1]) == n - 1
    path, hx, ml = fixtures.set_source(name)
    a = Automaton()
    a.load_file(path, hx, ml)
    a.compile()
    m = Matcher(a, 0, max_text=n, plane_capacity=1 << 21)
    a.close()
    for mode in ("sparse", "chain"):
        m.set_mode(mode)
        assert_same(m.scan(text), exp)
    m.set_mode("sparse")
    m.scan(text)
    assert m.path_taken(n) == "sparse"
    m.close()


def test_auto_mode_hands_dense_batches_to_the_chain_pipeline(gpu):
    """AUTO: after 16 sparse batches of which at least 8 were dense in matches (here: all; more
    than a record per 128 bytes), the next 64 go to the chain pipeline directly; then the sparse
    pipeline is tried again, 4 batches at a time, and every bad look quadruples the chain
    pipeline's share.  Results are the oracle's throughout; a forced SPARSE mode stays."""
    a, o = build(SMALL_SETS["nested"])
    m = Matcher(a, 0, max_text=1 << 17)
    endless = np.frombuffer(b"abc" * 40000, dtype=np.uint8)      # a record every byte or so
    quiet = np.frombuffer(b"abd" * 40000, dtype=np.uint8)
    exp_endless, exp_quiet = o.scan(endless), o.scan(quiet)
    m.set_mode("auto")
    paths = []
    for i in range(16 + 64 + 4 + 256 + 2):
        assert_same(m.scan(endless), exp_endless)
        paths.append(m.path_taken(endless.size))
    assert paths[:16] == ["sparse"] * 16
    assert paths[16:80] == ["chain"] * 64
    assert paths[80:84] == ["sparse"] * 4                         # trying again, a short look this time
    assert paths[84:340] == ["chain"] * 256                       # still dense: four times as long on the chain pipeline
    assert paths[340:] == ["sparse"] * 2
    m.set_mode("sparse")
    for i in range(20):
        assert_same(m.scan(endless), exp_endless)
        assert m.path_taken(endless.size) == "sparse"
    m.close()
    a, o = build(SMALL_SETS["nested"])
    m = Matcher(a, 0, max_text=1 << 17)                           # a quiet text stays on the sparse pipeline
    for i in range(40):
        assert_same(m.scan(quiet), exp_quiet)
        assert m.path_taken(quiet.size) == "sparse"
    m.close()


def test_auto_mode_counts_sample_heavy_batches(gpu):
    """AUTO also leaves the sparse pipeline for texts without a single match whose sampled 3-grams
    keep hitting the pattern set (real binaries: common 3-grams of code and tables): more than a
    flagged sample per 48 bytes is more than the check kernel's helper waves keep up with."""
    pats = synth.load_hex_patterns(os.path.join(orc.DATA, "clamav", "15000.txt"), 400)
    a, o = build(pats)
    n = 1 << 20
    rng = np.random.default_rng(5)
    text = rng.integers(0, 256, size=n, dtype=np.uint8)
    for p in range(8, n - 8, 8):                  # at every sampled position bytes 1..4 of some signature: its 3-gram
        g = np.frombuffer(pats[(p >> 3) % len(pats)][1:5], dtype=np.uint8)   # at offset 2 with the byte in front of it
        text[p - 1:p + 3] = g
    exp = o.scan(text)
    m = Matcher(a, 0, max_text=n)
    m.set_mode("auto")
    paths = []
    for i in range(16 + 4):
        assert_same(m.scan(text), exp)
        paths.append(m.path_taken(n))
    assert paths[:16] == ["sparse"] * 16 and paths[16:] == ["chain"] * 4
    m.close()


@pytest.mark.parametrize("shortest", [9, 13])
def test_six_byte_filter_keys(gpu, shortest):
    """Where every pattern has W + 5 bytes the filter looks at 6 bytes of a sample instead of 3 (shortest
    pattern 9: stride 4; 13: stride 8).  Same planes as the oracle's, on text with signatures planted
    whole, cut at either end, and sharing their first bytes."""
    allp = synth.load_hex_patterns(os.path.join(orc.DATA, "clamav", "15000.txt"), 1200)
    pats = [p[:shortest + (i % 40)] for i, p in enumerate(allp) if len(p) >= shortest + 40]
    pats = list(dict.fromkeys(pats))[:600]
    assert min(len(p) for p in pats) == shortest
    a, o = build(pats)
    m = Matcher(a, 0, max_text=1 << 21)
    assert m.set_mode("sparse") == "sparse"
    rng = np.random.default_rng(shortest)
    for n in (4096, 100003, (1 << 21) - 5):
        text = rng.integers(0, 256, size=n, dtype=np.uint8)
        for _ in range(n // 300):
            p = np.frombuffer(pats[int(rng.integers(len(pats)))], dtype=np.uint8)
            cut = int(rng.integers(0, 3))
            piece = p if cut == 0 else p[:max(3, p.size - int(rng.integers(1, 6)))] if cut == 1 else p[int(rng.integers(1, 4)):]
            at = int(rng.integers(0, max(1, n - piece.size)))
            text[at:at + piece.size] = piece[:n - at]
        assert_same(m.scan(text), o.scan(text))
    m.close()


def test_six_byte_keys_text_ends_inside_pattern(gpu):
    """6-byte filter keys, stride 8 (shortest pattern 13, D = 10): a text that ends 10..12 bytes inside a
    pattern, at every alignment of its start against the stride.  The one sample of such a path has its key
    cut off by the end of the text, so no follower is made and the last state comes from the tail walk
    (sparse.hip side_walks: W + 4 bytes, not D - 1).  The state must be the oracle's, and a second segment
    that starts with the rest of the pattern must report it from that carried state."""
    allp = synth.load_hex_patterns(os.path.join(orc.DATA, "clamav", "15000.txt"), 1200)
    pats = [p[:13 + (i % 40)] for i, p in enumerate(allp) if len(p) >= 13 + 40]
    pats = list(dict.fromkeys(pats))[:600]
    assert min(len(p) for p in pats) == 13
    a, o = build(pats)
    m = Matcher(a, 0, max_text=1 << 16)
    assert m.set_mode("sparse") == "sparse"
    rng = np.random.default_rng(1312)
    checked = 0
    for d in (9, 10, 11, 12, 13, 14):
        for align in range(8):
            P = pats[int(rng.integers(len(pats)))]
            if len(P) <= d:
                continue
            n = 4096 + align + d                     # the pattern starts at 4096 + align
            text = rng.integers(0, 256, size=n, dtype=np.uint8)
            text[n - d:] = np.frombuffer(P[:d], dtype=np.uint8)
            exp = o.scan(text)
            got = m.scan(text)
            assert_same(got, exp)
            assert m.path_taken(text.size) == "sparse"
            rest = np.concatenate([np.frombuffer(P[d:], dtype=np.uint8),
                                   rng.integers(0, 256, size=512, dtype=np.uint8)])
            exp2 = o.scan(rest, exp[2])
            assert (len(P) - d - 1) in exp2[0].tolist()   # the straddling signature is reported
            assert_same(m.scan(rest, got[2]), exp2)
            checked += 1
    assert checked >= 40
    m.close()


def test_real_binary_content(gpu):
    """Not synthetic: 48 MiB out of the middle of the largest ROCm library on the box (code, zero
    pages, tables, strings) x 2000 and 15000 signatures.  Whatever path the data makes the sparse
    pipeline take, piece by piece, the planes are the oracle's."""
    import glob
    import os
    libs = sorted(glob.glob("/opt/rocm/lib/*.so*"), key=lambda f: os.path.getsize(f) if os.path.isfile(f) else 0)
    if not libs or os.path.getsize(libs[-1]) < (64 << 20):
        pytest.skip("no large library to read")
    size = os.path.getsize(libs[-1])
    with open(libs[-1], "rb") as fh:
        fh.seek((size // 2) & ~4095)
        data = np.frombuffer(fh.read(48 << 20), dtype=np.uint8)
    for name in ("clamav2000", "clamav15000"):
        o = fixtures.oracle_for(name)
        path, hx, ml = fixtures.set_source(name)
        a = Automaton()
        a.load_file(path, hx, ml)
        a.compile()
        m = Matcher(a, 0, max_text=16 << 20, plane_capacity=1 << 22)
        a.close()
        state, paths = 0, []
        for mode in ("auto", "sparse"):
            m.set_mode(mode)
            state = 0
            for at in range(0, data.size, 16 << 20):      # streamed in 16 MiB pieces, state carried
                piece = data[at:at + (16 << 20)]
                exp = o.scan(piece, state, cap=1 << 22)
                got = m.scan(piece, state)
                assert_same(got, exp)
                paths.append(m.path_taken(piece.size))
                state = exp[2]
        assert set(paths) <= {"sparse", "chain"}
        print(name, os.path.basename(libs[-1]), paths)
        m.close()


def _planes(pat, off, cap, stream=None):
    p, o = pat.to_numpy(np.int32, cap, stream=stream), off.to_numpy(np.int32, cap, stream=stream)
    m = int(p[0])
    return o[1:1 + m].astype(np.uint32), p[1:1 + m].copy(), int(p[m + 1])


@pytest.mark.parametrize("group", [1, 2, 3, 4, 8, 16])
def test_grouped_batches(gpu, group):
    """acm_scan_batches_async puts consecutive sparse batches of one size into one set of launches:
    the planes of every batch are those of scanning it alone -- different texts, different carried-in
    states, halos; batches that share a workspace, differ in size or take the chain pipeline are
    enqueued on their own, in order."""
    pats = synth.load_hex_patterns(os.path.join(orc.DATA, "clamav", "15000.txt"), 400)
    a, o = build(pats)
    n = (1 << 20) + 40
    m = Matcher(a, 0, max_text=n)
    assert m.set_mode("sparse") == "sparse"
    assert m.set_max_group(group) == group
    rng = np.random.default_rng(group)
    nb = 7
    texts, inits = [], []
    for k in range(nb):
        t = rng.integers(0, 256, size=n, dtype=np.uint8)
        for _ in range(300):
            p = np.frombuffer(pats[int(rng.integers(len(pats)))], dtype=np.uint8)
            at = int(rng.integers(0, n - p.size))
            t[at:at + p.size] = p
        if k % 3 == 1:   # a signature that began in the previous buffer
            p = np.frombuffer(pats[k], dtype=np.uint8)
            t[:p.size - 5] = p[5:]
            inits.append(o.scan(p[:5].tobytes())[2])
        else:
            inits.append(0)
        texts.append(t)
    ws_bytes = m.lib.acm_scan_workspace_bytes(m.dfa, n)
    cap = 1 << 14
    d_texts = [DeviceArray.from_numpy(t) for t in texts]
    wss = [DeviceArray(ws_bytes) for _ in range(nb)]
    planes = [(DeviceArray(cap * 4), DeviceArray(cap * 4)) for _ in range(nb)]
    sizes = [n, n, n, n - 16, n, n, n]                   # batch 3 breaks the run
    ws_of = [0, 1, 2, 3, 4, 4, 5]                        # 4 and 5 share a workspace: never together
    batches = [m.make_batch(d_texts[k], sizes[k], m.stream, planes[k][0], planes[k][1], cap, (wss[ws_of[k]], ws_bytes),
                            init_state=inits[k], halo=(64 if k == 2 else 0), offset_shift=(1000 if k == 2 else 0))
               for k in range(nb)]
    for rep in range(2):
        m.enqueue_many(batches)
        for k in range(nb):
            got = _planes(planes[k][0], planes[k][1], cap, m.stream)
            epos, epat, elast = o.scan(texts[k][:sizes[k]], init_state=inits[k])
            if k == 2:
                keep = epos >= 64
                epos, epat = epos[keep] + 1000, epat[keep]
            assert_same(got, (epos, epat, elast))
    # a run of seventeen batches that can all share launches: full groups and a rest
    wss2 = [DeviceArray(ws_bytes) for _ in range(17)]
    planes2 = [(DeviceArray(cap * 4), DeviceArray(cap * 4)) for _ in range(17)]
    run = [m.make_batch(d_texts[k % nb], n, m.stream, planes2[k][0], planes2[k][1], cap, (wss2[k], ws_bytes),
                        init_state=inits[k % nb]) for k in range(17)]
    m.enqueue_many(run)
    for k in range(17):
        got = _planes(planes2[k][0], planes2[k][1], cap, m.stream)
        assert_same(got, o.scan(texts[k % nb], init_state=inits[k % nb]))
    m.close()


def test_workspaces_reused_across_block_sizes(gpu):
    """The check kernel takes blocks of 16 tiles in a launch group of four or more batches and of 8 when a batch is
    launched alone: the rows of a workspace are laid out differently by the two.  The same workspaces used one way,
    the other and back give the planes of scanning each text alone every time."""
    pats = synth.load_hex_patterns(os.path.join(orc.DATA, "clamav", "15000.txt"), 400)
    a, o = build(pats)
    n = (1 << 20) + 8
    m = Matcher(a, 0, max_text=n)
    assert m.set_mode("sparse") == "sparse"
    assert m.set_max_group(16) == 16
    rng = np.random.default_rng(99)
    texts = []
    for k in range(5):
        t = rng.integers(0, 256, size=n, dtype=np.uint8)
        for _ in range(200 + 100 * k):
            p = np.frombuffer(pats[int(rng.integers(len(pats)))], dtype=np.uint8)
            at = int(rng.integers(0, n - p.size))
            t[at:at + p.size] = p
        texts.append(t)
    exp = [o.scan(t) for t in texts]
    ws_bytes = m.lib.acm_scan_workspace_bytes(m.dfa, n)
    cap = 1 << 14
    d_texts = [DeviceArray.from_numpy(t) for t in texts]
    wss = [DeviceArray(ws_bytes) for _ in range(5)]
    planes = [(DeviceArray(cap * 4), DeviceArray(cap * 4)) for _ in range(5)]

    def batch(k, w):
        return m.make_batch(d_texts[k], n, m.stream, planes[k][0], planes[k][1], cap, (wss[w], ws_bytes))

    def check(ks):
        for k in ks:
            assert_same(_planes(planes[k][0], planes[k][1], cap, m.stream), exp[k])
            planes[k][0].fill(0, stream=m.stream)

    m.enqueue_many([batch(k, k) for k in range(5)])                 # one group of five: blocks of 16 tiles
    check(range(5))
    for k in range(5):                                              # each alone, on ANOTHER batch's workspace: blocks of 8
        m.enqueue(batch(k, (k + 2) % 5))
        check([k])
    m.enqueue_many([batch(k, (k + 1) % 5) for k in range(4)])       # a group of four again
    check(range(4))
    m.enqueue_many([batch(k, k) for k in range(3)])                 # a group of three: blocks of 8
    check(range(3))
    m.close()


def test_grouped_batches_large_tiles(gpu):
    """A launch group of texts large enough for tiles of several sub-blocks (40 MiB + 3: 16 KiB tiles, the
    bulk kernel's waves go through two sub-blocks per tile and ten tiles per batch), length not a
    multiple of 16."""
    pats = synth.load_hex_patterns(os.path.join(orc.DATA, "clamav", "15000.txt"), 2000)
    a, o = build(pats)
    n = (40 << 20) + 3
    m = Matcher(a, 0, max_text=n)
    assert m.set_mode("sparse") == "sparse"
    ws_bytes = m.lib.acm_scan_workspace_bytes(m.dfa, n)
    cap = 1 << 14
    texts = [synth.clamav_corpus(n, 40 + k, pats, 3000) for k in range(3)]
    long_one = next(q for q in pats if len(q) >= 60)
    texts[1][n - 40:] = np.frombuffer(long_one, dtype=np.uint8)[:40]      # a signature cut by the end of the text
    d_texts = [DeviceArray.from_numpy(t) for t in texts]
    wss = [DeviceArray(ws_bytes) for _ in range(3)]
    planes = [(DeviceArray(cap * 4), DeviceArray(cap * 4)) for _ in range(3)]
    m.enqueue_many([m.make_batch(d_texts[k], n, m.stream, planes[k][0], planes[k][1], cap, (wss[k], ws_bytes))
                    for k in range(3)])
    for k in range(3):
        assert_same(_planes(planes[k][0], planes[k][1], cap, m.stream), o.scan(texts[k]))
    m.close()


def test_grouped_batches_mixed_reports(gpu):
    """The batches of a launch group keep what is theirs: one reports pattern indices, its neighbour --
    same text -- the final states (all-patterns reporting); offsets agree, the state plane is that of a
    scan enqueued alone."""
    from gpu_pattern_matching_amd import _lib
    pats = synth.load_hex_patterns(os.path.join(orc.DATA, "clamav", "15000.txt"), 500)
    a, o = build(pats)
    n = 1 << 20
    text = synth.clamav_corpus(n, 3, pats, 400)
    m = Matcher(a, 0, max_text=n)
    assert m.set_mode("sparse") == "sparse"
    ws_bytes = m.lib.acm_scan_workspace_bytes(m.dfa, n)
    cap = 1 << 12
    d = DeviceArray.from_numpy(text)
    wss = [DeviceArray(ws_bytes) for _ in range(3)]
    planes = [(DeviceArray(cap * 4), DeviceArray(cap * 4)) for _ in range(3)]
    alone = m.make_batch(d, n, m.stream, planes[2][0], planes[2][1], cap, (wss[2], ws_bytes), report=_lib.REPORT_STATE)
    m.enqueue(alone)
    want_states = _planes(planes[2][0], planes[2][1], cap, m.stream)
    m.enqueue_many([m.make_batch(d, n, m.stream, planes[0][0], planes[0][1], cap, (wss[0], ws_bytes)),
                    m.make_batch(d, n, m.stream, planes[1][0], planes[1][1], cap, (wss[1], ws_bytes),
                                 report=_lib.REPORT_STATE)])
    heads = _planes(planes[0][0], planes[0][1], cap, m.stream)
    states = _planes(planes[1][0], planes[1][1], cap, m.stream)
    assert_same(heads, o.scan(text))
    assert np.array_equal(states[0], heads[0]) and states[2] == heads[2]
    assert np.array_equal(states[1], want_states[1]) and not np.array_equal(states[1], heads[1])
    m.close()
