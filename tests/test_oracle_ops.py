"""CPU restatements of the result post-processing ops (oracle side).

The reference's only executable checks for these are DATABUF_TEST's invariants
(databuf.c:935-1021): exclusive prefix sum == running sum; compaction returns
the total in cell 0 and the elements in order.
"""
import numpy as np

import orc


def test_exclusive_scan_running_sum():
    rng = np.random.default_rng(1)
    a = rng.integers(0, 129, size=5000).astype(np.int32)
    out = orc.exclusive_scan(a)
    assert out[0] == 0
    assert np.array_equal(out[1:], np.cumsum(a)[:-1])


def test_compaction_databuf_test_invariants():
    # databuf.c:935-1019: random per-chunk counts r < max_results, bucket cell j of chunk i
    # holds a running counter; after compaction dst[0] == total and dst[i+1] == i.
    rng = np.random.default_rng(2)
    chunks, max_results = 777, 129
    src = np.zeros(max_results * chunks + 1, dtype=np.int32)
    count = 0
    for i in range(chunks):
        r = int(rng.integers(0, max_results))
        src[i] = r
        for j in range(r):
            src[(j + 1) * chunks + i] = count
            count += 1
    src[max_results * chunks] = 4242  # last-state trailer
    prefix = orc.exclusive_scan(src[:chunks])
    dst = orc.compact_array(src, prefix, chunks, max_results, count + 2)
    assert dst[0] == count
    assert np.array_equal(dst[1:count + 1], np.arange(count))
    assert dst[count + 1] == 4242


def test_bitonic_network_sorts():
    rng = np.random.default_rng(3)
    for length in (2, 8, 512, 1024, 4096):
        batch = max(1, 1024 // length)
        k = rng.integers(0, 1 << 32, size=batch * length, dtype=np.uint64).astype(np.uint32)
        v = np.arange(batch * length, dtype=np.uint32)
        for d in (0, 1):
            rc, ko, vo = orc.bitonic_sort(k, v, batch, length, d)
            assert rc == 0
            for b in range(batch):
                seg = ko[b * length:(b + 1) * length]
                exp = np.sort(k[b * length:(b + 1) * length])
                assert np.array_equal(seg, exp if d else exp[::-1])
            assert np.array_equal(k[vo], ko)  # values travelled with their keys


def test_bitonic_rejects_bad_shapes():
    k = np.arange(6, dtype=np.uint32)
    assert orc.bitonic_sort(k, k, 1, 6, 1)[0] == -1        # not a power of two
    k = np.arange(256, dtype=np.uint32)
    assert orc.bitonic_sort(k, k, 1, 256, 1)[0] == -1      # batch*len % 512 != 0 (<= 512 case)
    assert orc.bitonic_sort(k[:1], k[:1], 1, 1, 1)[0] == 0  # too short: nothing to do


def test_sentinels_sink_to_the_end():
    # the sort's intended use (old/ocl_aho_grep.c.20200110:113-121): -1 sentinels last
    keys = np.array([5, 0xFFFFFFFF, 3, 0xFFFFFFFF, 9, 1, 0xFFFFFFFF, 7] * 64, dtype=np.uint32)
    vals = np.arange(keys.size, dtype=np.uint32)
    rc, ko, vo = orc.bitonic_sort(keys, vals, 1, keys.size, 1)
    assert rc == 0
    n_real = int((keys != 0xFFFFFFFF).sum())
    assert (ko[:n_real] != 0xFFFFFFFF).all() and (ko[n_real:] == 0xFFFFFFFF).all()


def test_bucketize_and_walk():
    pos = np.array([3, 10, 17, 18, 40, 41, 42, 43], dtype=np.uint32)
    pat = np.arange(8, dtype=np.int32) + 100
    indices = np.array([0, 16, 32, 48], dtype=np.int32)
    sizes = np.array([16, 16, 12, 16], dtype=np.int32)
    res, res2 = orc.bucketize(pos, pat, indices, sizes, 4, 77)
    chunks = 4
    assert res[:chunks].tolist() == [2, 2, 4, 0]
    assert res[1 * chunks + 0] == 100 and res2[1 * chunks + 0] == 3
    assert res[2 * chunks + 1] == 103 and res2[2 * chunks + 1] == 18
    # chunk 2 has 4 records but only max_results-1 = 3 cells
    assert [res[(k + 1) * chunks + 2] for k in range(3)] == [104, 105, 106]
    assert res[4 * chunks] == 77
