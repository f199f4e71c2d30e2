"""HIP post-processing ops vs their CPU restatements: exclusive scan (ocl_prefix_sum),
bucket compaction (compactarray.cl), bitonic key/value sort (BitonicSort.cl), bucketize."""
import numpy as np
import pytest

import orc
from gpu_pattern_matching_amd import api

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 2, 255, 256, 1023, 1024, 1025, 4096, 100000, (1 << 20) + 3, 3000001])
def test_exclusive_scan(gpu, n):
    rng = np.random.default_rng(n)
    a = rng.integers(0, 129, size=n).astype(np.int32)
    out, total = api.exclusive_scan(a)
    assert np.array_equal(out, orc.exclusive_scan(a))
    assert total == int(a.sum())


def test_exclusive_scan_empty(gpu):
    out, total = api.exclusive_scan(np.zeros(0, np.int32))
    assert out.size == 0 and total == 0


@pytest.mark.parametrize("chunks,max_results", [(1, 17), (777, 129), (32768, 16), (5000, 2)])
def test_compact_buckets_databuf_test(gpu, chunks, max_results):
    """DATABUF_TEST (databuf.c:935-1019) on the device: cell 0 = total, cells in order, trailer."""
    rng = np.random.default_rng(chunks)
    src = np.zeros(max_results * chunks + 1, dtype=np.int32)
    count = 0
    for i in range(chunks):
        r = int(rng.integers(0, max_results))
        src[i] = r
        for j in range(r):
            src[(j + 1) * chunks + i] = count
            count += 1
    src[max_results * chunks] = 31337
    prefix, total = api.exclusive_scan(src[:chunks])
    assert total == count
    dst = api.compact_buckets(src, prefix, chunks, max_results, count + 2)
    exp = orc.compact_array(src, prefix, chunks, max_results, count + 2)
    assert np.array_equal(dst, exp)
    assert dst[0] == count and np.array_equal(dst[1:count + 1], np.arange(count)) and dst[count + 1] == 31337


def test_compact_buckets_overflowing_chunk(gpu):
    """Q14: a chunk that counted more matches than it has cells leaves holes, as in the reference."""
    chunks, max_results = 8, 4
    src = np.zeros(max_results * chunks + 1, dtype=np.int32)
    src[:chunks] = [1, 9, 0, 2, 3, 0, 7, 1]
    for i in range(chunks):
        for j in range(min(src[i], max_results - 1)):
            src[(j + 1) * chunks + i] = 100 * i + j
    prefix, _ = api.exclusive_scan(src[:chunks])
    total = int(src[:chunks].sum())
    dst = api.compact_buckets(src, prefix, chunks, max_results, total + 2)
    assert np.array_equal(dst, orc.compact_array(src, prefix, chunks, max_results, total + 2))


@pytest.mark.parametrize("length,batch", [(2, 256), (8, 64), (64, 16), (512, 1), (512, 4), (1024, 2),
                                          (2048, 1), (4096, 1), (4096, 3), (1 << 16, 1), (1 << 20, 1)])
@pytest.mark.parametrize("direction", [0, 1])
def test_bitonic_sort_matches_reference_network(gpu, length, batch, direction):
    rng = np.random.default_rng(length * 7 + batch)
    # few distinct keys => many ties: the value order then depends on the exact network
    k = rng.integers(0, 50, size=batch * length).astype(np.uint32)
    k[rng.random(k.size) < 0.2] = 0xFFFFFFFF
    v = np.arange(k.size, dtype=np.uint32)
    rc, ko, vo = api.bitonic_sort(k, v, batch, length, direction)
    erc, eko, evo = orc.bitonic_sort(k, v, batch, length, direction)
    assert rc == 0 and erc == 0
    assert np.array_equal(ko, eko)
    assert np.array_equal(vo, evo)
    for b in range(batch):
        seg = ko[b * length:(b + 1) * length].astype(np.int64)
        assert (np.diff(seg) >= 0).all() if direction else (np.diff(seg) <= 0).all()


def test_bitonic_sort_rejects_non_power_of_two(gpu):
    k = np.arange(6, dtype=np.uint32)
    assert api.bitonic_sort(k, k, 1, 6, 1)[0] == -1
    rc, ko, _ = api.bitonic_sort(k[:1], k[:1], 1, 1, 1)     # too short: nothing happens
    assert rc == 0 and ko[0] == 0


def test_bucketize(gpu):
    rng = np.random.default_rng(9)
    n, B, R = 1 << 16, 256, 5
    pos = np.sort(rng.choice(n, size=3000, replace=False)).astype(np.int32)
    pat = rng.integers(0, 1000, size=pos.size).astype(np.int32)
    m = pos.size
    pat_plane = np.concatenate([[m], pat, [77]]).astype(np.int32)
    off_plane = np.concatenate([[m], pos, [77]]).astype(np.int32)
    chunks = n // B
    indices = (np.arange(chunks) * B).astype(np.int32)
    sizes = np.full(chunks, B, dtype=np.int32)
    sizes[-1] = 100                                        # short tail chunk
    r, r2 = api.bucketize(pat_plane, off_plane, indices, sizes, R)
    er, er2 = orc.bucketize(pos.astype(np.uint32), pat, indices, sizes, R, 77)
    # cells beyond a chunk's count are unspecified in the reference; compare defined cells
    assert np.array_equal(r[:chunks], er[:chunks]) and np.array_equal(r2[:chunks], er2[:chunks])
    for i in range(chunks):
        for k in range(min(int(er[i]), R - 1)):
            assert r[(k + 1) * chunks + i] == er[(k + 1) * chunks + i]
            assert r2[(k + 1) * chunks + i] == er2[(k + 1) * chunks + i]
    assert r[R * chunks] == 77
