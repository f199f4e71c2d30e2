"""The multi-GPU path where only one GPU is available: RCCL really runs (communicator of one
rank), the shard form of the scan and the gather go through the C ABI, and bench.py's collective
path runs over torch.distributed's nccl backend at world size 1.  What a second GPU would add is
more ranks in the same calls; the sharding arithmetic across ranks is covered on the CPU
(test_sharding_gloo.py, test_multi_abi.py)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import fixtures
from gpu_pattern_matching_amd import Automaton, DeviceArray, Matcher, _lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def test_gather_planes_over_rccl(gpu, lib):
    """Two 'ranks' worth of shards scanned on the one device, each gathered through
    acm_gather_planes on a one-rank RCCL communicator (ncclSend/ncclRecv to itself inside a
    group), merged by acm_merge_planes: the records of the serial scan of the whole text."""
    rccl = C.CDLL("librccl.so")
    uid = NcclUniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, NcclUniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    name = "clamav2000"
    o = fixtures.oracle_for(name)
    pats = fixtures.patterns_of(name)
    text = fixtures.text_for({"kind": "clamav", "n": 1 << 20, "seed": 91, "n_plant": 700}, pats)
    path, hx, ml = fixtures.set_source(name)
    a = Automaton()
    a.load_file(path, hx, ml)
    a.compile()
    L = a.max_pattern_len
    cap = 4096
    m = Matcher(a, 0, max_text=text.size, plane_capacity=cap)
    a.close()
    world = 2
    host_pat = np.zeros((world, cap), dtype=np.int32)
    host_off = np.zeros((world, cap), dtype=np.int32)
    for rank in range(world):
        plan = _lib.ShardPlan()
        assert lib.acm_shard_plan_for(text.size, world, rank, L, C.byref(plan)) == 0
        d = DeviceArray.from_numpy(text[plan.load_begin:plan.load_begin + plan.load_bytes])
        m.scan_async(d, plan.load_bytes, halo=plan.halo, offset_shift=plan.offset_shift)
        all_pat, all_off = DeviceArray(cap * 4), DeviceArray(cap * 4)
        if rank == 0:      # the fixed-capacity gather ...
            rc = lib.acm_gather_planes(comm, 0, 1, 0, m.pat_plane.ptr, m.off_plane.ptr, cap, all_pat.ptr, all_off.ptr, m.stream)
            assert rc == 0, lib.acm_last_error()
        else:              # ... and the one that gathers the record counts first and sends count + 2 cells
            all_pat.fill(0xEE, stream=m.stream)
            all_off.fill(0xEE, stream=m.stream)
            d_counts = DeviceArray(64)
            counts = (C.c_int32 * 1)()
            rc = lib.acm_gather_planes_sized(comm, 0, 1, 0, m.pat_plane.ptr, m.off_plane.ptr, cap, all_pat.ptr, all_off.ptr,
                                             d_counts.ptr, counts, m.stream)
            assert rc == 0, lib.acm_last_error()
            got = all_pat.to_numpy(np.int32, cap, stream=m.stream)
            assert counts[0] == got[0] and 0 < counts[0] < cap - 2
            assert (got[counts[0] + 2:] == np.int32(-286331154)).all()     # 0xEEEEEEEE: nothing behind the trailer cell was sent
            d_counts.free()
        host_pat[rank] = all_pat.to_numpy(np.int32, cap, stream=m.stream)
        host_off[rank] = all_off.to_numpy(np.int32, cap, stream=m.stream)
        for b in (d, all_pat, all_off):
            b.free()
    out_pat = np.zeros(world * cap, dtype=np.int32)
    out_off = np.zeros(world * cap, dtype=np.int32)
    last = C.c_long()
    total = lib.acm_merge_planes(host_pat.ctypes.data, host_off.ctypes.data, world, cap, out_pat.ctypes.data,
                                 out_off.ctypes.data, out_pat.size, C.byref(last))
    epos, epat, elast = o.scan(text)
    assert total == epos.size and last.value == elast
    assert np.array_equal(out_off[:total].astype(np.uint32), epos) and np.array_equal(out_pat[:total], epat)
    m.close()
    rccl.ncclCommDestroy(comm)


@pytest.mark.parametrize("gather", ["torch", "abi"])
def test_bench_collective_path_at_world_size_one(gpu, gather):
    """bench.py with the nccl process group initialised at world size 1: init, the gather of the
    planes tensor on its own stream behind the workers' scans, the barriers of the timed blocks --
    the code the driver's multi-GPU run executes -- with parity checked by bench.py itself.  gather = abi: the planes
    go through acm_gather_planes on a communicator bench.py makes for itself, and the parity check reads the
    gathered copy."""
    env = dict(os.environ, ACM_BENCH_NCCL1="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "24", "--warmup", "4", "--repeats", "2",
                        "--texts", "3", "--sub", "", "--no-cpu-baseline", "--no-e2e", "--gather", gather], capture_output=True, text=True,
                       timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    rec = json.loads(p.stdout.strip().splitlines()[-1])
    assert rec["parity"].startswith("bit-exact") and rec["config"]["pipeline"] == "sparse"
    assert rec["config"]["gather"].startswith("acm_gather_planes" if gather == "abi" else "torch")


@pytest.mark.parametrize("ranks,scaling", [(2, "weak"), (2, "strong"), (4, "weak")])
def test_bench_ranks_rehearsal(gpu, ranks, scaling):
    """Two ranks of bench.py on the one GPU (ACM_BENCH_BACKEND=gloo: both on cuda:0, the gather goes
    through the host): shard plan, halo, offset shift, merged parity against the oracle's scan of
    the whole logical text -- the multi-rank logic of the driver's N > 1 runs, minus RCCL (which
    the world-size-1 tests above exercise)."""
    env = dict(os.environ, ACM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(29541 + ranks), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "16", "--warmup", "4",
           "--repeats", "2", "--texts", "2", "--scaling", scaling]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == ranks and rec["scaling"] == scaling
    assert rec["parity"].startswith("bit-exact"), rec["parity"]
    want = (32 << 20) * (ranks if scaling == "weak" else 1)
    assert rec["config"]["text_bytes_per_step"] == want
