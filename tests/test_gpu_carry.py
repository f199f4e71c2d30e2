"""The final state of a scan handed to the next one ON THE DEVICE (acm_scan_batch.d_init_plane): a text cut
into pieces, every piece enqueued with the planes of the piece in front as its source of the start state, no
host read in between -- the records of all pieces are those of one serial scan of the whole text
(the reference carries the state through the host: databuf.c:622, ahomatch.cl:42-43).  All three
pipelines: sparse, the chain pipeline's LDS-resident walk, its cold-plane kernels."""
import os

import numpy as np
import pytest

import fixtures
import orc
import synth
from gpu_pattern_matching_amd import Automaton, DeviceArray, Matcher

pytestmark = pytest.mark.gpu


def _planes(pat, off, cap, stream=None):
    p = pat.to_numpy(np.int32, cap, stream=stream)
    q = off.to_numpy(np.int32, cap, stream=stream)
    m = int(p[0])
    return q[1:1 + m].astype(np.uint32), p[1:1 + m].copy(), int(p[m + 1])


def stream_pieces(m, text, cuts, cap):
    """scan text[cuts[i]:cuts[i+1]] for all i, each piece starting where the piece in front ended (device hand-over)"""
    sizes = [cuts[i + 1] - cuts[i] for i in range(len(cuts) - 1)]
    ws_bytes = m.lib.acm_scan_workspace_bytes(m.dfa, max(max(sizes), 1))
    ws = DeviceArray(ws_bytes)
    d_pieces = [DeviceArray.from_numpy(np.ascontiguousarray(text[cuts[i]:cuts[i + 1]])) if sizes[i] else DeviceArray(16)
                for i in range(len(sizes))]
    planes = [(DeviceArray(cap * 4), DeviceArray(cap * 4)) for _ in sizes]
    batches = []
    for i, n in enumerate(sizes):
        batches.append(m.make_batch(d_pieces[i], n, m.stream, planes[i][0], planes[i][1], cap, (ws, ws_bytes),
                                    init_plane=planes[i - 1][0] if i else None, init_plane_capacity=cap if i else 0))
    m.enqueue_many(batches)       # one call, nothing read back in between
    pos, pat, last = [], [], 0
    for i in range(len(sizes)):
        o, p, last = _planes(planes[i][0], planes[i][1], cap, m.stream)
        pos.append(o.astype(np.int64) + cuts[i])
        pat.append(p)
    for b in d_pieces + [ws] + [x for pr in planes for x in pr]:
        b.free()
    return np.concatenate(pos).astype(np.uint32), np.concatenate(pat), last


def check(m, o, text, cuts, cap):
    got = stream_pieces(m, text, cuts, cap)
    exp = o.scan(text)
    assert got[0].size == exp[0].size
    assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1]) and got[2] == exp[2]


def test_sparse_pipeline_signatures_across_the_cuts(gpu):
    name = "clamav2000"
    o = fixtures.oracle_for(name)
    pats = fixtures.patterns_of(name)
    n = 1 << 20
    text = synth.clamav_corpus(n, 77, pats, 600)
    cuts = [0, 300000, 300001, 300001, 655360, 655360 + 40, n]       # a one-byte piece, an empty one, a 40-byte one
    long_one = np.frombuffer(next(p for p in pats if len(p) >= 100), dtype=np.uint8)
    for c in cuts[1:-1]:                                              # a signature across every cut
        at = max(0, c - 50)
        text[at:at + long_one.size] = long_one[:n - at]
    path, hx, ml = fixtures.set_source(name)
    a = Automaton()
    a.load_file(path, hx, ml)
    a.compile()
    m = Matcher(a, 0, max_text=n)
    a.close()
    for mode in ("sparse", "chain"):
        m.set_mode(mode)
        check(m, o, text, cuts, 1 << 13)
    m.close()


@pytest.mark.parametrize("lds", [True, False])
def test_chain_pipeline_words_across_the_cuts(gpu, monkeypatch, lds):
    if not lds:
        monkeypatch.setenv("ACM_SCAN_NO_LDSWALK", "1")
    sent = os.path.join(orc.DATA, "sentiment", "patterns_categorical.txt")
    a = Automaton()
    a.load_file(sent, False, -1)
    a.compile()
    o = orc.Oracle()
    o.load(sent)
    o.compile()
    words = open(os.path.join(orc.DATA, "sentiment", "top5000_words.txt")).read().split()
    text = synth.word_corpus(400000, 31, words)
    rng = np.random.default_rng(8)
    cuts = sorted(set([0, text.size] + [int(x) for x in rng.integers(1, text.size, size=9)]))
    m = Matcher(a, 0, max_text=text.size)
    assert m.lds_resident() == lds
    m.set_mode("chain")
    check(m, o, text, cuts, 1 << 16)
    m.close()
    a.close()
    o.close()
