"""Parity of the HIP scan pipeline (through the C ABI) with the oracle and the golden
vectors produced by the reference's acsmx.c.  Bit-exact: offsets, pattern indices, order,
count and final state (reference numbering).
"""
import hashlib
import json
import os

import numpy as np
import pytest

import fixtures
import orc
from gpu_pattern_matching_amd import AcmError, Automaton, DeviceArray, Matcher
from gpu_pattern_matching_amd._lib import check

pytestmark = pytest.mark.gpu

with open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")) as _f:
    GOLDEN = json.load(_f)["sets"]

_matchers = {}


def matcher_for(name, max_text=1 << 20):
    if name not in _matchers:
        if len(_matchers) >= 2:       # keep device memory bounded (15000 sigs = 0.7 GB)
            _matchers.pop(next(iter(_matchers))).close()
        path, hx, max_len = fixtures.set_source(name)
        a = Automaton()
        a.load_file(path, hx, max_len)
        a.compile()
        _matchers[name] = Matcher(a, 0, max_text=max_text)
        a.close()
    m = _matchers[name]
    m.reserve(max_text)
    m.set_chain_bytes(0)
    m.set_chains_per_lane(4)
    return m


def assert_same(got, exp):
    assert got[0].size == exp[0].size, "record count %d != %d" % (got[0].size, exp[0].size)
    assert np.array_equal(got[0], exp[0]), "offsets differ"
    assert np.array_equal(got[1], exp[1]), "pattern indices differ"
    assert got[2] == exp[2], "final state %d != %d" % (got[2], exp[2])


GOLDEN_CASES = [(name, i) for name in GOLDEN for i in range(len(GOLDEN[name]["texts"]))]


@pytest.mark.parametrize("name,idx", GOLDEN_CASES,
                         ids=["%s-%s" % (n, fixtures.spec_id(GOLDEN[n]["texts"][i])) for n, i in GOLDEN_CASES])
@pytest.mark.parametrize("mode", ["auto", "chain"])
def test_golden_vectors(gpu, name, idx, mode):
    """Every golden vector, at BASELINE's full 32 MiB sizes too, by digest (no oracle needed),
    through the pipeline the library picks for the set and through the chain pipeline."""
    spec = GOLDEN[name]["texts"][idx]
    pats = fixtures.patterns_of(name) if spec["kind"] in ("clamav", "repeat") else None
    text = fixtures.text_for(spec, pats)
    assert hashlib.sha256(text.tobytes()).hexdigest() == spec["sha256"]
    m = matcher_for(name, max_text=max(text.size, 1 << 20))
    m.set_mode(mode)
    pos, pat, last = m.scan(text)
    m.set_mode("auto")
    assert pos.size == spec["count"]
    assert "%016x" % orc.records_digest(pos, pat) == spec["records_digest"]
    assert last == spec["final_state"]
    assert [[int(a), int(b)] for a, b in zip(pos[:8], pat[:8])] == spec["first_records"]
    assert (np.diff(pos.astype(np.int64)) > 0).all()      # strictly increasing offsets


@pytest.mark.parametrize("chain", [16, 32, 64, 128, 256])
@pytest.mark.parametrize("name", ["tests", "sentiment", "clamav2000", "clamav2000_m12"])
def test_chain_length_invariance(gpu, name, chain):
    """The result must not depend on how the text is cut into chains, nor on how many chains a
    lane interleaves in the walk kernel."""
    o = fixtures.oracle_for(name)
    pats = fixtures.patterns_of(name)
    if name == "sentiment":
        text = fixtures.text_for({"kind": "words", "n": 300001, "seed": 4}, None)
    else:
        text = fixtures.text_for({"kind": "clamav", "n": 300001, "seed": 4, "n_plant": 300}, pats)
    m = matcher_for(name)
    m.set_chain_bytes(chain)
    exp = o.scan(text)
    for cpl in (2, 4):
        assert m.set_chains_per_lane(cpl) == cpl
        assert_same(m.scan(text), exp)


@pytest.mark.parametrize("n", [0, 1, 2, 15, 16, 17, 63, 64, 65, 127, 129, 4095, 4096, 4097, 8191,
                               65535, 65537])
def test_ragged_lengths(gpu, n):
    o = fixtures.oracle_for("tests")
    base = np.fromfile(os.path.join(orc.DATA, "ref_tests", "input.txt"), dtype=np.uint8)
    text = np.tile(base, 8)[:n]
    m = matcher_for("tests")
    for cpl in (2, 4):
        m.set_chains_per_lane(cpl)
        for chain in (16, 64):
            m.set_chain_bytes(chain)
            assert_same(m.scan(text), o.scan(text))


def test_streaming_last_state_contract(gpu):
    """databuf.c:622 / ahomatch.cl:42-43: the state after one buffer seeds the next; cutting a
    text anywhere (also inside a match) and carrying last_state gives the one-shot answer."""
    name = "clamav2000"
    o = fixtures.oracle_for(name)
    pats = fixtures.patterns_of(name)
    text = fixtures.text_for({"kind": "clamav", "n": 1 << 18, "seed": 12, "n_plant": 400}, pats)
    whole = o.scan(text)
    m = matcher_for(name)
    rng = np.random.default_rng(12)
    cuts = sorted(set(int(c) for c in rng.integers(1, text.size - 1, size=5)) |
                  {int(whole[0][3]) - 2, int(whole[0][10])})   # two cuts inside matches
    pos_all, pat_all, state, start = [], [], 0, 0
    for cut in cuts + [text.size]:
        pos, pat, state = m.scan(text[start:cut], state)
        pos_all.append(pos.astype(np.int64) + start)
        pat_all.append(pat)
        start = cut
    assert np.array_equal(np.concatenate(pos_all), whole[0].astype(np.int64))
    assert np.array_equal(np.concatenate(pat_all), whole[1])
    assert state == whole[2]


def test_deep_states_everywhere(gpu):
    """Worst case for the boundary resolve: every chain starts deep inside a pattern."""
    name = "clamav2000"
    o = fixtures.oracle_for(name)
    pats = fixtures.patterns_of(name)
    m = matcher_for(name)
    longest = max(range(len(pats)), key=lambda i: len(pats[i]))
    for pid in (5, longest):
        p = pats[pid]
        text = np.frombuffer((p * (200000 // len(p) + 1))[:200000], dtype=np.uint8)
        for cpl in (2, 4):
            m.set_chains_per_lane(cpl)
            for chain in (16, 64, 256):
                m.set_chain_bytes(chain)
                assert_same(m.scan(text), o.scan(text))


def test_every_position_matches(gpu):
    """Maximum result density: a one-byte pattern over a constant text (m == n)."""
    a = Automaton()
    a.add(b"a", 7)
    a.add(b"aa", 8)
    a.compile()
    o = orc.Oracle()
    o.add(b"a", 7)
    o.add(b"aa", 8)
    o.compile()
    text = np.full(100000, ord("a"), dtype=np.uint8)
    m = Matcher(a, 0, max_text=text.size)
    for cpl in (2, 4):
        m.set_chains_per_lane(cpl)
        for chain in (16, 256):
            m.set_chain_bytes(chain)
            got = m.scan(text)
            assert got[0].size == text.size
            assert_same(got, o.scan(text))
    m.close()


def test_idempotent_and_async_reuse(gpu):
    """Scanning the same resident buffer repeatedly (the bench loop) never changes the answer."""
    name = "sentiment"
    o = fixtures.oracle_for(name)
    text = fixtures.text_for({"kind": "words", "n": 1 << 19, "seed": 2}, None)
    exp = o.scan(text)
    m = matcher_for(name)
    d = DeviceArray.from_numpy(text)
    for _ in range(5):
        m.scan_async(d, text.size)
    assert_same(m.fetch(), exp)
    d.free()


def test_capacity_overflow_is_reported(gpu):
    name = "sentiment"
    text = fixtures.text_for({"kind": "words", "n": 1 << 16, "seed": 2}, None)
    exp = fixtures.oracle_for(name).scan(text)
    path, hx, ml = fixtures.set_source(name)
    a = Automaton()
    a.load_file(path, hx, ml)
    a.compile()
    m = Matcher(a, 0, max_text=text.size, plane_capacity=100)
    d = DeviceArray.from_numpy(text)
    m.scan_async(d, text.size)
    with pytest.raises(AcmError) as e:
        m.fetch()
    assert e.value.code == -8
    pat = m.pat_plane.to_numpy(np.int32, 100)
    off = m.off_plane.to_numpy(np.int32, 100)
    assert pat[0] == exp[0].size                       # full count still reported
    assert np.array_equal(off[1:99], exp[0][:98].astype(np.int32))   # first capacity-2 records
    assert pat[99] == exp[2]                           # state in the last cell
    m.close()


def test_bad_arguments(gpu, lib):
    m = matcher_for("tests")
    d = DeviceArray(64)
    with pytest.raises(AcmError):
        m.scan_async(d, 16, init_state=10 ** 6)        # not a state
    with pytest.raises(AcmError):
        m.scan_async(d.ptr + 1, 16)                    # misaligned text
    with pytest.raises(AcmError) as e:
        m.scan_async(d, 1 << 31, workspace=(m.ws, m.ws_bytes))   # beyond the 2 GiB - 17 buffer limit
    assert e.value.code == -5
    with pytest.raises(AcmError):
        m.scan_async(d, 16, report=7)                  # not an ACM_REPORT_* value
    d.free()


@pytest.mark.parametrize("mode", ["chain", "auto"])
def test_graph_replay_tracks_buffer_contents(gpu, mode):
    """With acm_scan_set_graphs on, a scan that repeats with the same buffers is replayed as a HIP
    graph from its third enqueue on: the replay reads what is in the buffers then, not what
    was there at capture, and gives what separate launches give."""
    name = "clamav2000"
    o = fixtures.oracle_for(name)
    pats = fixtures.patterns_of(name)
    m = matcher_for(name, max_text=1 << 20)
    m.set_mode(mode)
    n = 1 << 20
    d = DeviceArray(n)
    try:
        for graphs in (True, False):
            assert m.set_graphs(graphs) == graphs
            for seed in range(5):
                text = fixtures.text_for({"kind": "clamav", "n": n, "seed": 40 + seed, "n_plant": 100 * seed},
                                         pats)
                check(m.lib.acm_rt_memcpy_h2d(d.ptr, text.ctypes.data, n, m.stream), "h2d")
                m.scan_async(d, n)
                assert_same(m.fetch(), o.scan(text))
    finally:
        m.set_graphs(False)
        m.set_mode("auto")
        d.free()


@pytest.mark.parametrize("name", ["tests", "sentiment", "clamav2000"])
def test_all_patterns_reporting(gpu, name):
    """scan with the final states in the pattern plane, expand every state's
    match list on the device -> one record per pattern ending at each offset, in list order.
    The default planes (head only) stay what the reference reports."""
    o = fixtures.oracle_for(name)
    pats = fixtures.patterns_of(name)
    if name == "sentiment":
        text = fixtures.text_for({"kind": "words", "n": 200003, "seed": 8}, None)
    else:
        text = fixtures.text_for({"kind": "clamav", "n": 200003, "seed": 8, "n_plant": 300}, pats)
    m = matcher_for(name)
    modes = ["chain", "sparse"] if m.sparse_eligible() else ["chain"]
    exp_all, exp_head = o.scan_all(text), o.scan(text)
    assert exp_all[0].size >= exp_head[0].size
    if name == "sentiment":
        assert exp_all[0].size > exp_head[0].size      # nested words: more than one pattern per offset
    try:
        for mode in modes:
            m.set_mode(mode)
            assert_same(m.scan_all(text), exp_all)
            assert_same(m.scan(text), exp_head)
            for init in (0, exp_head[2]):
                assert_same(m.scan_all(text[:5000], init), o.scan_all(text[:5000], init))
    finally:
        m.set_mode("auto")


def test_all_patterns_nested_set_sparse(gpu):
    """Match lists longer than one entry through the sparse pipeline (patterns of >= 3 bytes that
    are suffixes of each other, and a duplicate)."""
    pats = [b"abcabc", b"bcabc", b"cabc", b"abc", b"abc", b"bca", b"xabcabc"]
    a = Automaton()
    o = orc.Oracle()
    for i, p in enumerate(pats):
        a.add(p, 10 + i)
        o.add(p, 10 + i)
    a.compile()
    o.compile()
    m = Matcher(a, 0, max_text=1 << 16)
    assert m.sparse_eligible()
    rng = np.random.default_rng(3)
    text = np.frombuffer(b"abcx", dtype=np.uint8)[rng.integers(0, 4, size=60000)]
    exp = o.scan_all(text)
    assert exp[0].size > o.scan(text)[0].size
    for mode in ("chain", "sparse"):
        m.set_mode(mode)
        assert_same(m.scan_all(text, out_capacity=exp[0].size + 2), exp)
    with pytest.raises(AcmError) as e:
        m.scan_all(text, out_capacity=100)             # too small for the expansion: reported
    assert e.value.code == -8
    m.close()
    a.close()
