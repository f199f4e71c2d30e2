"""Multi-process path on CPU (gloo, world size 2): shard plan, halo handling, fixed-capacity
gather and merge -- the host logic bench.py runs over RCCL.  The scan itself is the oracle here
(no GPU in this test); on the GPU the same plan feeds acm_scan_shard_async."""
import os
import socket
import sys

import numpy as np
import pytest

import fixtures
from gpu_pattern_matching_amd import sharding

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_plan_covers_text_once():
    n, L = 1000003, 159
    for world in (1, 2, 3, 8):
        covered = 0
        for r in range(world):
            p = sharding.shard_plan(n, world, r, L)
            assert p["begin"] == covered
            covered = p["end"]
            assert p["halo"] == min(L - 1, p["begin"])
            assert p["load_begin"] == p["begin"] - p["halo"]
            assert p["load_bytes"] == p["end"] - p["begin"] + p["halo"]
        assert covered == n


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, name, n, seed, q):
    try:
        sys.path.insert(0, HERE)
        sys.path.insert(0, os.path.dirname(HERE))
        import torch
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        o = fixtures.oracle_for(name)
        pats = fixtures.patterns_of(name)
        text = fixtures.text_for({"kind": "clamav", "n": n, "seed": seed, "n_plant": 200}, pats)
        # a signature planted right across the shard border
        border = n // world
        text[border - 4:border - 4 + len(pats[3])] = np.frombuffer(pats[3], dtype=np.uint8)
        plan = sharding.shard_plan(n, world, rank, o.max_pattern_len)
        local = text[plan["load_begin"]:plan["end"]]
        pos, pat, last = o.scan(local, 0)                 # the scan a GPU rank would do
        keep = pos >= plan["halo"]                        # drop records ending in the halo
        pos, pat = pos[keep].astype(np.int64) + plan["offset_shift"], pat[keep]
        cap = 4096
        planes = torch.zeros((2, cap), dtype=torch.int32)
        m = pos.size
        planes[0, 0] = planes[1, 0] = m
        planes[0, 1:1 + m] = torch.from_numpy(pat.astype(np.int32))
        planes[1, 1:1 + m] = torch.from_numpy(pos.astype(np.int32))
        planes[0, m + 1] = planes[1, m + 1] = last
        got = sharding.gather_planes(planes, dist, dst=0)
        if rank == 0:
            offs, pids, last_state = sharding.merge_gathered(got)
            epos, epat, elast = o.scan(text, 0)
            ok = (np.array_equal(offs, epos) and np.array_equal(pids, epat) and last_state == elast
                  and (border + len(pats[3]) - 5) in epos.tolist())
            q.put(("ok" if ok else "mismatch", int(epos.size)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        q.put(("error: %r" % (e,), 0))


@pytest.mark.timeout(300)
def test_two_rank_shard_scan_gather_equals_serial():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, "clamav2000_m12", 300000, 77, q))
             for r in range(2)]
    for p in procs:
        p.start()
    status, count = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
    assert status == "ok", status
    assert count > 100
