"""The multi-GPU host logic behind the C ABI (multi.cpp): shard plan and plane merge, against the
Python restatement the sharding tests use (gpu_pattern_matching_amd/sharding.py).  CPU only: no
device entry point is called."""
import ctypes as C

import numpy as np

from gpu_pattern_matching_amd import _lib, sharding


def test_shard_plan_matches_python(lib):
    for n in (0, 1, 1000, 33554432, (1 << 31) - 17):
        for world in (1, 2, 3, 8):
            for L in (0, 1, 4, 159, 4095):
                for rank in range(world):
                    p = _lib.ShardPlan()
                    assert lib.acm_shard_plan_for(n, world, rank, L, C.byref(p)) == 0
                    want = sharding.shard_plan(n, world, rank, L)
                    got = {"begin": p.begin, "end": p.end, "halo": p.halo, "load_begin": p.load_begin,
                           "load_bytes": p.load_bytes, "offset_shift": p.offset_shift}
                    assert got == want, (n, world, rank, L)
    p = _lib.ShardPlan()
    assert lib.acm_shard_plan_for(100, 2, 2, 5, C.byref(p)) < 0      # rank out of range
    assert lib.acm_shard_plan_for(100, 0, 0, 5, C.byref(p)) < 0


def test_merge_planes_matches_python(lib):
    rng = np.random.default_rng(3)
    cap, world = 64, 4
    planes = np.zeros((world, 2, cap), dtype=np.int32)
    base = 0
    for r in range(world):
        m = int(rng.integers(0, cap - 1))
        planes[r, :, 0] = m
        planes[r, 0, 1:1 + m] = rng.integers(0, 1000, size=m)
        planes[r, 1, 1:1 + m] = base + np.sort(rng.integers(0, 5000, size=m))
        planes[r, :, m + 1] = 100 + r
        base += 5000
    want_off, want_pat, want_last = sharding.merge_gathered([planes[r] for r in range(world)])
    all_pat = np.ascontiguousarray(planes[:, 0, :])
    all_off = np.ascontiguousarray(planes[:, 1, :])
    out_pat = np.zeros(world * cap, dtype=np.int32)
    out_off = np.zeros(world * cap, dtype=np.int32)
    last = C.c_long()
    total = lib.acm_merge_planes(all_pat.ctypes.data, all_off.ctypes.data, world, cap, out_pat.ctypes.data,
                                 out_off.ctypes.data, out_pat.size, C.byref(last))
    assert total == want_off.size and last.value == want_last
    assert np.array_equal(out_pat[:total], want_pat) and np.array_equal(out_off[:total].astype(np.uint32), want_off)
    # count only, and the two ways it can fail
    assert lib.acm_merge_planes(all_pat.ctypes.data, all_off.ctypes.data, world, cap, None, None, 0, None) == total
    assert lib.acm_merge_planes(all_pat.ctypes.data, all_off.ctypes.data, world, cap, out_pat.ctypes.data,
                                out_off.ctypes.data, max(total - 1, 0), None) == (_lib.ACM_ERR_CAPACITY if total else 0)
    all_pat[1, 0] = cap          # a rank claims more records than its planes hold
    assert lib.acm_merge_planes(all_pat.ctypes.data, all_off.ctypes.data, world, cap, None, None, 0, None) == _lib.ACM_ERR_CAPACITY
