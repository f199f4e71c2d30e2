#!/usr/bin/env python3
"""bench.py -- the reference's headline workload on MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): input GB/s scanned (+ matches/s), 32 MB text x N ClamAV signatures.
A step = one pass of the scan pipeline over one 32 MiB batch already resident in HBM: the
library's sparse pipeline for signature sets whose shortest pattern has 3 bytes (strided 3-gram
sieve -> exact prefix check -> trie-path followers -> ordered emit; three kernels), its chain
pipeline otherwise (--mode forces one).  Like the reference, which keeps -w worker threads in
flight on one device, each with its own queue and buffers (ocl_aho_grep.c:37-144, :498-502),
the steps of a block are dealt over --workers HIP streams with private scratch and handed to
the library with one acm_scan_batches_async call (--issue threads: one call per stream, from a
host thread each); the library puts up to --group consecutive sparse batches of a stream into
one set of kernel launches (--group 1: a set of launches per step).  The steps rotate over
--texts distinct 32 MiB texts (64 = 2 GiB by default: eight times the 256 MiB Infinity Cache, so
that with every batch in flight the text a step reads comes from HBM -- sub-records included).  The timed region is --repeats blocks of --steps steps, each block
bracketed by a barrier + device synchronisation; the line reports the median block.

With N > 1 GPUs the logical text of a step is N x 32 MiB (weak scaling) or one 32 MiB text cut
N ways (--scaling strong); rank g scans shard g (+ an (L-1)-byte halo), the DFA is replicated,
and the compact match planes of all steps of a block go to rank 0 in ONE RCCL gather inside
the timed block.

Prints ONE JSON line on rank 0.  The headline is BASELINE.json configs[1] (clamav2000); at N = 1
the line also carries sub-records for configs[2] (clamav15000) and configs[4] (sentiment), an
end-to-end figure that includes the host-to-device copy, and the CPU baseline.
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SHARD = 32 << 20
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
L3_BYTES = 256 << 20        # Infinity Cache


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


class Workload:
    """Pattern set + corpus generator of one BASELINE.json config."""

    def __init__(self, name, plant):
        import synth
        self.name = name
        self.plant = plant
        data = os.path.join(ROOT, "tests", "data")
        if name.startswith("clamav"):
            self.sigs = int(name[len("clamav"):])
            self.pats = synth.load_hex_patterns(os.path.join(data, "clamav", "15000.txt"), self.sigs)
            self.pattern_file = None
            self.what = ("32 MiB per GPU of seeded uniform bytes + %d planted signatures x first %d ClamAV sigs"
                         % (plant, self.sigs))
            self.cap = 1 << 13      # records never exceed the planted signatures (uniform bytes match nothing)
            while self.cap < plant + 1024:
                self.cap *= 2
        elif name == "sentiment":
            self.sigs = None
            self.pats = None
            self.pattern_file = os.path.join(data, "sentiment", "patterns_categorical.txt")
            self.words = open(os.path.join(data, "sentiment", "top5000_words.txt")).read().split()
            self.what = ("32 MiB of space-separated words, half of them from the sentiment vocabulary, x the "
                         "4376 categorical sentiment patterns (apps/patterns.txt)")
            self.cap = 1 << 21      # ~0.93 M records per 32 MiB
        else:
            raise SystemExit("unknown workload " + name)

    def automaton(self):
        from gpu_pattern_matching_amd import Automaton
        aut = Automaton()
        if self.pats is not None:
            for i, p in enumerate(self.pats):
                aut.add(p, i)
        else:
            aut.load_file(self.pattern_file, False, -1)
        aut.compile()
        return aut

    def oracle(self):
        import orc  # tests/orc.py -> oracle/ (checker + CPU baseline only)
        o = orc.Oracle()
        if self.pats is not None:
            for i, p in enumerate(self.pats):
                o.add(p, i)
        else:
            o.load(self.pattern_file)
        o.compile()
        return o

    def text(self, seed, n=SHARD):
        import synth
        if self.pats is not None:
            return synth.clamav_corpus(n, seed, self.pats, self.plant)
        return synth.word_corpus(n, seed, self.words)


class Pool:
    """--workers host threads: n - 1 parked between blocks (a thread per block would cost more than a
    step) plus the caller, which takes the last share itself instead of sleeping through the block."""

    def __init__(self, n):
        self.n = n
        self.go = threading.Barrier(n)
        self.done = threading.Barrier(n)
        self.job = None
        self.err = []
        self.threads = [threading.Thread(target=self._run, args=(i,), daemon=True) for i in range(n - 1)]
        for t in self.threads:
            t.start()

    def _run(self, i):
        while True:
            self.go.wait()
            if self.job is None:
                return
            try:
                self.job(i)
            except Exception as e:      # surfaced by run()
                self.err.append(e)
            self.done.wait()

    def run(self, job):
        self.job = job
        self.go.wait()
        try:
            job(self.n - 1)
        except Exception as e:
            self.err.append(e)
        self.done.wait()
        if self.err:
            raise self.err[0]

    def close(self):
        self.job = None
        self.go.wait()


def rccl_communicator(torch, dist, rank, world, dev):
    """--gather abi: an RCCL communicator of the caller's own, as a C or Go host of libacmatch.so would make it
    (ncclGetUniqueId on rank 0, handed round -- here through the process group --, ncclCommInitRank), for
    acm_gather_planes."""
    import ctypes as C

    class NcclUniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    rccl = C.CDLL("librccl.so")
    uid = NcclUniqueId()
    if rank == 0 and rccl.ncclGetUniqueId(C.byref(uid)) != 0:
        raise RuntimeError("ncclGetUniqueId failed")
    t = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).to(dev)
    dist.broadcast(t, src=0)
    C.memmove(C.byref(uid), bytes(t.cpu().numpy().tobytes()), 128)
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, NcclUniqueId, C.c_int]
    if rccl.ncclCommInitRank(C.byref(comm), world, uid, rank) != 0:
        raise RuntimeError("ncclCommInitRank failed")
    return comm


def run_workload(ctx, wl, steps, warmup, repeats, ntexts, workers, verify, headline):
    """Everything for one workload; returns the record (rank 0) or None."""
    import torch
    import torch.distributed as dist
    from gpu_pattern_matching_amd import Matcher, sharding
    from gpu_pattern_matching_amd._lib import check
    rank, world, dev, args = ctx["rank"], ctx["world"], ctx["dev"], ctx["args"]
    strong = args.scaling == "strong" and world > 1

    t0 = time.perf_counter()
    aut = wl.automaton()
    t_compile = time.perf_counter() - t0
    L, states = aut.max_pattern_len, aut.num_states
    matcher = Matcher(aut, ctx["local_rank"], max_text=16, plane_capacity=2)
    matcher.set_mode(args.mode)
    if args.chain_bytes:
        matcher.set_chain_bytes(args.chain_bytes)
    aut.close()
    log(rank, "%s: %d states, L=%d, compile %.2fs, device %.1f MB, sparse eligible: %s" % (
        wl.name, states, L, t_compile, matcher.device_bytes / 1e6, matcher.sparse_eligible()))
    if workers <= 0:
        # automatic: three.  The sparse pipeline's batches go eight to a launch and three streams of such
        # groups keep the bulk kernel busy; a fourth adds 3-5 % on long blocks (4.3-4.5 against 4.1-4.3 TB/s),
        # nothing on 20-step blocks, and costs 50-step blocks 4 % -- and the bulk kernels of the streams queue
        # behind each other, so every launch then takes a third longer from dispatch to end.  The chain
        # pipeline's LDS-resident walk fills every CU by itself (all of its LDS): two streams, one's walk kernel and
        # the other's scatter kernel in its shadow (a third only adds scatter workgroups that keep walk
        # workgroups waiting for a CU: 940 against 1060 GB/s).
        workers = 2 if matcher.lds_resident() and args.mode != "sparse" and not matcher.sparse_eligible() else 3

    # ---- texts.  Logical text i = world shards of 32 MiB (weak) or one 32 MiB text (strong); rank r
    #      loads its range plus the halo in front of it -------------------------------------------
    # ntexts distinct texts (2 GiB at the default of 64: a text's next use is far beyond the 256 MiB
    # Infinity Cache, whatever the number of batches in flight).  The word corpus is slow to generate, so
    # beyond `unique` base texts the rest are the base texts rotated by a large odd number of bytes:
    # different content at every address, checked against the oracle like any other text.
    unique = ntexts if wl.pats is not None else min(ntexts, 8)
    base_cache = {}

    def shard(i, r):
        b, turn = i % unique, i // unique
        if (b, r) not in base_cache:
            base_cache[(b, r)] = wl.text(7 + b * 64 + r)
        t = base_cache[(b, r)]
        return t if turn == 0 else np.roll(t, turn * 1048583)

    def logical(i):
        return shard(i, 0) if (world == 1 or strong) else np.concatenate([shard(i, r) for r in range(world)])

    total_bytes = SHARD if strong else SHARD * world
    plan = sharding.shard_plan(total_bytes, world, rank, L)
    n_local = plan["load_bytes"]
    d_texts, h_texts = [], []
    for i in range(ntexts):
        if strong:
            mine = shard(i, 0)[plan["load_begin"]:plan["end"]]
        else:
            mine = shard(i, rank)
            if plan["halo"]:
                mine = np.concatenate([shard(i, rank - 1)[-plan["halo"]:], mine])
        assert mine.size == n_local
        t = torch.zeros((n_local + 15) // 16 * 16, dtype=torch.uint8, device=dev)
        t[:n_local] = torch.from_numpy(mine).to(dev)
        d_texts.append(t)
        if i < 2:
            h_texts.append(mine)
    if ntexts > 16:
        base_cache.clear()       # (regenerated for the texts the check looks at)
    # a text is read again after ntexts - 1 others: HBM only if that is well beyond the Infinity Cache
    residency = ("HBM (%d MiB of texts = %.1f x the Infinity Cache)" % (ntexts * n_local >> 20, ntexts * n_local / L3_BYTES)
                 if ntexts * n_local >= 4 * L3_BYTES else
                 "Infinity Cache in part or whole (%d MiB of texts < 4 x 256 MiB)" % (ntexts * n_local >> 20))

    ws_bytes = matcher.lib.acm_scan_workspace_bytes(matcher.dfa, n_local)
    cap = wl.cap
    W = max(1, workers)
    K = max(W, steps)
    streams = [torch.cuda.Stream(device=dev) for _ in range(W)]
    # a worker's consecutive steps alternate between G workspaces: acm_scan_batches_async then puts
    # up to G of them into one set of launches (--group 1: every step has its own three launches)
    G = max(1, min(args.group, 16))
    G = matcher.lib.acm_scan_set_max_group(matcher.dfa, G)
    wss = [[torch.empty(ws_bytes, dtype=torch.uint8, device=dev) for _ in range(G)] for _ in range(W)]
    pe = max(1, args.profile_every)
    # launch groups exist for the sparse pipeline and for the chain pipeline's LDS-resident walk; a workload on
    # the cold-plane chain kernels is dealt to the workers step by step (else a 20-step block would put 16
    # consecutive steps on one stream)
    groups_apply = matcher.group_capable()
    Geff = G if (args.issue != "main" and groups_apply) else 1
    if args.deal != "groups":
        # a block shorter than W full groups: every worker an equal share (20 steps: 7 + 7 + 6 instead of 16 + 4,
        # +3 %; the chain pipeline's walk kernels run one after the other whatever the stream, each with the
        # scatter of the group in front in its shadow: 10 + 10 steps overlap, 16 + 4 hardly, +2 %)
        Geff = max(1, min(Geff, -(-K // W)))

    # The steps are dealt to the workers a launch group at a time: steps 0 .. Geff-1 to worker 0, the next
    # Geff to worker 1, ...  (a short block then ends with ONE short group, not with one per worker)
    def worker_of(k):
        return (k // Geff) % W

    # planes of every step of a block: [K, 2, cap] when they all have to survive until the gather /
    # the check, else a ring of R slots PER WORKER (two streams never write one slot; the slots of a
    # launch group are distinct; the last ncheck steps of a block -- the ones that get checked -- survive)
    ncheck = max(1, min(ntexts, args.check_texts))      # the last steps of a block whose planes are checked
    R = max(Geff, ncheck, 2)
    ring = not (world > 1 or K * cap * 8 <= (512 << 20)) and W * R < K
    slots = W * R if ring else K

    def slot_of(k):
        if not ring:
            return k
        local = (k // (Geff * W)) * Geff + k % Geff      # the step's number among its worker's
        return worker_of(k) * R + local % R

    planes = torch.zeros((slots, 2, cap), dtype=torch.int32, device=dev)
    gathered = [torch.empty_like(planes) for _ in range(world)] if (world > 1 and rank == 0) else None
    gather_stream = torch.cuda.Stream(device=dev)
    abi_buf = None
    if ctx.get("rccl_comm") is not None:
        if gathered is None and rank == 0:       # (world size 1 rehearsal: the gather still runs, into a scratch copy)
            gathered = [torch.empty_like(planes)]
        if rank == 0:
            abi_buf = [torch.empty(world * planes.numel() // 2, dtype=torch.int32, device=dev) for _ in range(2)]

    def timed(k):
        """is step k one of worker 0's whose kernels are timed?  Whole launch groups, every pe-th of them."""
        return worker_of(k) == 0 and (k // (Geff * W)) % pe == 0

    timed_steps = sum(1 for k in range(K) if timed(k))

    def batches(profile):
        out = []
        for k in range(K):
            w, p = worker_of(k), planes[slot_of(k)]
            out.append(matcher.make_batch(d_texts[k % ntexts], n_local, streams[w].cuda_stream, p[0], p[1], cap,
                                          (wss[w][k % G], ws_bytes), halo=plan["halo"],
                                          offset_shift=plan["offset_shift"],
                                          profile=profile and timed(k)))
        return out

    plain, profiled = batches(False), batches(True)
    native = args.issue == "native" or (args.issue == "threads" and W == 1)   # one worker: the caller is the thread
    pool = Pool(W) if (W > 1 and args.issue == "threads") else None
    enq, enq_many = matcher.lib.acm_scan_batch_async, matcher.lib.acm_scan_batches_async
    dfa = matcher.dfa
    Batch = type(plain[0])
    # the steps of a block as arrays for acm_scan_batches_async: all of them in step order (native:
    # consecutive entries on one stream are what the library groups), or worker w's share (threads)

    # N > 1: a block's planes go to rank 0 in up to three gathers, each behind the scans it carries and
    # beside the scans of the next piece (a piece = whole rounds of launch groups: a contiguous range of
    # steps, i.e. of plane slots)
    def pieces(count, n=3):
        per_round = max(1, Geff * W)
        rounds = (count + per_round - 1) // per_round
        out, lo = [], 0
        for i in range(n):
            hi = min(count, ((rounds * (i + 1) + n - 1) // n) * per_round)
            if hi > lo:
                out.append((lo, hi))
                lo = hi
        return out

    whole = {(id(bs), c): (Batch * c)(*bs[:c]) for bs in (plain, profiled) for c in {K, min(K, max(warmup, W))}}
    chunked = native and dist.is_initialized() and not ring
    piecewise = {id(bs): [(lo, hi, (Batch * (hi - lo))(*bs[lo:hi])) for lo, hi in pieces(K)]
                 for bs in (plain, profiled)} if chunked else {}
    mine_of = [[k for k in range(K) if worker_of(k) == w] for w in range(W)]
    share = {id(bs): [(Batch * max(1, len(mine_of[w])))(*[bs[k] for k in mine_of[w]]) for w in range(W)]
             for bs in (plain, profiled)}

    def issue(bs, count):
        """count steps, step k on worker (stream) worker_of(k).  threads: every worker's steps are enqueued by
        its own host thread with one acm_scan_batches_async call; native: one call from this thread
        enqueues them all in step order; main: this thread, one FFI call per step."""
        def job(w):
            mine = sum(1 for k in mine_of[w] if k < count)
            rc = enq_many(dfa, share[id(bs)][w], mine) if mine else 0
            if rc:
                check(rc, "acm_scan_batches_async")
        t = time.perf_counter()
        if native:
            rc = enq_many(dfa, whole[(id(bs), count)], count)
            if rc:
                check(rc, "acm_scan_batches_async")
        elif pool is not None:
            pool.run(job)
        else:
            for k in range(count):
                rc = enq(dfa, C.byref(bs[k]))
                if rc:
                    check(rc, "acm_scan_batch_async")
        return time.perf_counter() - t

    def fence():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    def gather(lo=0, hi=None):
        if not dist.is_initialized():
            return
        hi = planes.shape[0] if hi is None else hi
        mine = planes[lo:hi]
        for s in streams:                               # behind every worker's scans so far
            gather_stream.wait_stream(s)
        with torch.cuda.stream(gather_stream):
            if ctx.get("rccl_comm") is not None:
                # --gather abi: the library's own gather on a communicator of its own (include/acmatch.h,
                # acm_gather_planes: ncclSend/ncclRecv in one group).  The piece's planes go as ONE pair of
                # "planes" of half its cells each -- the call moves bytes, it does not look at them.
                half = mine.numel() // 2
                flat = mine.view(-1)
                if rank == 0:
                    allp = abi_buf[0][:world * half]
                    allo = abi_buf[1][:world * half]
                rc = matcher.lib.acm_gather_planes(ctx["rccl_comm"], rank, world, 0, flat.data_ptr(), flat.data_ptr() + half * 4,
                                                   half, allp.data_ptr() if rank == 0 else None,
                                                   allo.data_ptr() if rank == 0 else None, gather_stream.cuda_stream)
                if rc:
                    check(rc, "acm_gather_planes")
                if rank == 0 and gathered is not None:
                    for r, g in enumerate(gathered):
                        gv = g[lo:hi].view(-1)
                        gv[:half].copy_(allp[r * half:(r + 1) * half])
                        gv[half:].copy_(allo[r * half:(r + 1) * half])
            elif ctx["backend"] == "nccl":
                into = [g[lo:hi] for g in gathered] if gathered is not None else [torch.empty_like(mine)]
                dist.gather(mine, gather_list=into if rank == 0 else None, dst=0)
            else:                                       # rehearsal on one GPU: through the host
                gather_stream.synchronize()
                host = mine.cpu()
                bufs = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
                dist.gather(host, gather_list=bufs, dst=0)
                if rank == 0 and gathered is not None:
                    for g, h in zip(gathered, bufs):
                        g[lo:hi].copy_(h)

    def block(bs):
        """the K steps of a block and the gather of their planes; returns the host's enqueue seconds"""
        if not chunked:
            t = issue(bs, K)
            gather()
            return t
        t = 0.0
        for lo, hi, arr in piecewise[id(bs)]:
            t0 = time.perf_counter()
            rc = enq_many(dfa, arr, hi - lo)
            if rc:
                check(rc, "acm_scan_batches_async")
            t += time.perf_counter() - t0
            gather(lo, hi)
        return t

    issue(plain, min(K, max(warmup, W)))
    gather()
    fence()
    blocks, host_issue = [], []
    matcher.profile_read()
    for rep in range(max(1, repeats)):
        fence()
        t0 = time.perf_counter()
        host_issue.append(block(profiled))      # (worker 0's launch groups carry HIP events in every block)
        fence()
        blocks.append(time.perf_counter() - t0)
    k1_ms, k2_ms, pipe_ms, launches = matcher.profile_read()
    elapsed = statistics.median(blocks)
    host_planes = planes.cpu().numpy()      # the planes of the last timed block
    gathered_host = [g.cpu().numpy() for g in gathered] if gathered is not None else None
    # the same block with a set of launches per step (no launch groups), for comparison: three blocks
    ungrouped = None
    if headline and world == 1 and Geff > 1 and matcher.sparse_eligible() and args.mode != "chain" and not args.no_extra:
        matcher.lib.acm_scan_set_max_group(matcher.dfa, 1)
        ub = []
        for rep in range(3):
            fence()
            t0 = time.perf_counter()
            block(plain)
            fence()
            ub.append(time.perf_counter() - t0)
        matcher.lib.acm_scan_set_max_group(matcher.dfa, G)
        ungrouped = statistics.median(ub)
    # one batch in flight (outside the timed region): what the kernels take when they have the GPU
    # to themselves
    scratch = torch.zeros((2, cap), dtype=torch.int32, device=dev)
    for i in range(0 if args.no_extra else 12):
        matcher.enqueue(matcher.make_batch(d_texts[i % ntexts], n_local, streams[0].cuda_stream, scratch[0], scratch[1],
                                           cap, (wss[0][0], ws_bytes), halo=plan["halo"],
                                           offset_shift=plan["offset_shift"], profile=True))
        torch.cuda.synchronize()
    s_k1, s_k2, s_pipe, s_n = matcher.profile_read()
    # ... and one launch group of Geff batches in flight, alone
    g_k1 = g_k2 = g_pipe = 0.0
    g_n = 0
    if Geff > 1 and not args.no_extra:
        gsc = torch.zeros((Geff, 2, cap), dtype=torch.int32, device=dev)
        for i in range(6):
            matcher.enqueue_many([matcher.make_batch(d_texts[(i * Geff + j) % ntexts], n_local, streams[0].cuda_stream,
                                                     gsc[j][0], gsc[j][1], cap, (wss[0][j], ws_bytes), halo=plan["halo"],
                                                     offset_shift=plan["offset_shift"], profile=True)
                                  for j in range(Geff)])
            torch.cuda.synchronize()
        g_k1, g_k2, g_pipe, g_n = matcher.profile_read()
    path = ("sparse (void: ACM_SIEVE_SKIP)" if os.environ.get("ACM_SIEVE_SKIP") else
            matcher.path_taken(n_local, streams[0].cuda_stream, workspace=(wss[0][0].data_ptr(), ws_bytes)))
    matcher_lds = matcher.lds_resident()

    red = dev if ctx["backend"] == "nccl" else torch.device("cpu")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- end to end with the host-to-device copy: the reference's Mbps definition counts bytes
    #      from the files to the results (ocl_aho_grep.c:580,598,614,628-630) ------------------------
    e2e = None
    if headline and world == 1 and not args.no_e2e:
        pinned = [torch.from_numpy(np.ascontiguousarray(h)).pin_memory() for h in h_texts]
        stage = [torch.zeros_like(d_texts[0]) for _ in range(W)]
        ne = 72            # 2.25 GiB behind a warm-up of 6 steps (24 cold steps measured the link's ramp: 37 of its 57 GB/s)
        We = min(W, 3)     # (copies over PCIe: three streams keep it busy, a fourth only adds contention)

        def run(count, scans):
            for k in range(count):
                w = k % We
                with torch.cuda.stream(streams[w]):
                    stage[w][:n_local].copy_(pinned[k % len(pinned)], non_blocking=True)
                if scans:
                    matcher.scan_async(stage[w], n_local, 0, streams[w].cuda_stream, scratch[0], scratch[1], cap,
                                       workspace=(wss[w][0], ws_bytes))
            torch.cuda.synchronize()

        run(6, True)
        t0 = time.perf_counter()
        run(ne, True)
        dt = time.perf_counter() - t0
        # ... and the copies alone, same buffers and streams: what the link gives (the ceiling of the figure above)
        run(6, False)
        t0 = time.perf_counter()
        run(ne, False)
        dt_copy = time.perf_counter() - t0
        e2e = {"value": round(SHARD * ne / dt / 1e9, 2), "unit": "GB/s", "steps": ne,
               "h2d_only": round(SHARD * ne / dt_copy / 1e9, 2),
               "what": "pinned host buffer -> hipMemcpyAsync -> scan, %d streams; PCIe included, file I/O not; "
                       "h2d_only: the same copies without the scans" % We}

    # ---- parity: the planes of the last steps of the last block, every distinct text once -----------
    owner = {slot_of(k): k for k in range(K)}            # the last step that wrote each slot
    last_k = [k for k in range(max(0, K - ncheck), K) if owner[slot_of(k)] == k]
    m_local = int(host_planes[slot_of(K - 1), 0, 0])
    m_total = m_local
    if world > 1:
        t = torch.tensor([m_local], dtype=torch.int64, device=red)
        dist.all_reduce(t)
        m_total = int(t.item())
    parity, cpu, out = "not checked", None, None
    if rank == 0:
        if verify or (headline and not args.no_cpu_baseline):
            o = wl.oracle()
            if verify:
                ok, nrec = True, 0
                for k in last_k:
                    if gathered_host is not None:
                        offs, pids, last_state = sharding.merge_gathered([g[slot_of(k)] for g in gathered_host])
                    else:
                        offs, pids, last_state = sharding.merge_gathered([host_planes[slot_of(k)]])
                    epos, epat, elast = o.scan(logical(k % ntexts), cap=max(cap * world, 1 << 16))
                    good = np.array_equal(epos, offs) and np.array_equal(epat, pids) and elast == last_state
                    if not good:
                        log(rank, "PARITY MISMATCH %s step %d: gpu %d records / oracle %d" % (
                            wl.name, k, offs.size, epos.size))
                    ok &= good
                    nrec += epos.size
                parity = ("bit-exact vs oracle serial scan on %d distinct texts (%d records)" % (len(last_k), nrec)
                          if ok else "MISMATCH")
            if headline and world == 1 and not args.no_cpu_baseline:
                mine = h_texts[0]
                best, reps, t_start = 1e30, 0, time.perf_counter()
                while time.perf_counter() - t_start < args.cpu_seconds or reps < 2:
                    t1 = time.perf_counter()
                    o.scan_count(mine)
                    best = min(best, time.perf_counter() - t1)
                    reps += 1
                ncpu = os.cpu_count() or 1
                t1 = time.perf_counter()
                o.scan_threads(mine, ncpu)
                t_all = time.perf_counter() - t1
                cpu = {"value": round(SHARD / best / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
                       "sample": "oracle/acref.c serial walk of the reference-format int32 table over one 32 MiB "
                                 "text of the workload, best of %d passes (%.1f s of CPU work)" % (
                                     reps, time.perf_counter() - t_start),
                       "all_cores": {"value": round(SHARD / t_all / 1e9, 4), "cores": ncpu}}
            o.close()

        # The roofline is quoted for the kernel that reads the text (the bulk pass: k_sieve, or
        # k_spec_walk of the chain pipeline): algorithmic bytes = N x 1 B of text + 8 B per record
        #  over that kernel's average duration, HIP events on the worker's own stream
        # inside the timed region.
        alg_batch = n_local + 8 * m_local
        # batches one launch of the bulk kernel processes (the timed groups' average: the last group
        # of a block may be short)
        grouped = (timed_steps * max(1, repeats) / max(launches, 1)) if (groups_apply and launches) else 1
        alg_bytes = int(alg_batch * grouped)
        kname = "k_sieve" if path == "sparse" else ("k_lds_walk" if matcher_lds else "k_spec_walk")
        L1 = max(launches, 1)
        k_s = k1_ms / 1e3 / L1
        achieved = alg_bytes / k_s / 1e9 if k_s > 0 else 0.0
        value = total_bytes * K / elapsed / 1e9
        traffic, stage_traffic = None, {}
        tfile = os.path.join(ROOT, "profiles", "r3_traffic_%s.json" % wl.name)
        if not os.path.exists(tfile):
            tfile = os.path.join(ROOT, "profiles", "r2_traffic_%s.json" % wl.name)
        if os.path.exists(tfile):
            for name, rec in json.load(open(tfile)).items():
                if name.startswith("k_"):     # (the short names; the file also has the full ones and the copies)
                    stage_traffic[name] = round(rec["hbm_bytes_per_launch"], 1)
            if kname == "k_spec_walk" and "k_halo_walk" in stage_traffic and "k_spec_walk" not in stage_traffic:
                kname = "k_halo_walk"     # (halo mode with the text loaded up front: its own kernel)
            if kname == "k_lds_walk" and "k_lds_walk" not in stage_traffic:
                stage_traffic = {}        # (counters of the kernels this workload no longer runs)
            traffic = stage_traffic.get(kname)
            if traffic is not None:
                traffic = round(traffic * grouped, 1)     # (the counters were collected with one batch per launch)
        out = {
            "value": round(value, 3),
            "steps": K,
            "ms_per_step": round(elapsed / K * 1e3, 5),
            "timed_region_ms": round(elapsed * 1e3, 3),
            "blocks_ms": [round(b * 1e3, 3) for b in blocks],
            "config": {
                "workload": "%s (%d states, max len %d); scan -> ordered compact (offset, pattern) planes%s" % (
                    wl.what, states, L, ", gathered to rank 0" if world > 1 else ""),
                "text_bytes_per_step": total_bytes,
                "distinct_texts": ntexts,
                "text_residency": residency,
                "pipeline": path,
                "workers": W,
                "batches_in_flight": min(K, W * Geff),
                "host_threads": W if pool is not None else 1,
                "issue": args.issue,
                "batches_per_launch_group": Geff,      # what a launch really carries (a 20-step block on 3 workers: 7)
                "max_launch_group": G if (groups_apply and args.issue != "main") else 1,
                "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                "parallelism": "text sharded %d-way (%s), DFA replicated" % (world, "strong" if strong else "weak"),
                "gather": ("acm_gather_planes (own RCCL communicator)" if ctx.get("rccl_comm") is not None
                           else "torch.distributed gather (RCCL)" if dist.is_initialized() else "none (one rank)"),
            },
            "host_enqueue_us_per_step": round(statistics.median(host_issue) / K * 1e6, 2),
            "matches_per_step": m_total,
            "matches_per_s": round(m_total * K / elapsed, 1),
            "frac_of_hbm_peak": round(value / world / HBM_PEAK_GBS, 5),
            "parity": parity,
            "roofline": {
                "bound": "hbm",
                "kernel": kname,
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes,
                "batches_per_launch": round(grouped, 2),
                "kernel_us": round(k_s * 1e6, 2),
                "rest_of_pipeline_us": round(k2_ms / L1 * 1e3, 2),
                "pipeline_us": round(pipe_ms / L1 * 1e3, 2),
                "launches_timed": launches,
                "stage_traffic": stage_traffic or None,
                "note": "HIP events on the worker's own stream around the kernels of every %d-th launch group of worker 0 "
                        "in every timed block (1: all of them); with %d workers a kernel shares the GPU with the other workers' "
                        "kernels.  traffic = the bulk kernel's counted HBM bytes for ONE batch (profiles/, collected "
                        "with --group 1) x batches_per_launch; stage_traffic: per batch" % (pe, W),
            },
        }
        if s_n:
            sk = s_k1 / 1e3 / s_n
            out["roofline_one_batch_in_flight"] = {
                "kernel_us": round(sk * 1e6, 2),
                "rest_of_pipeline_us": round(s_k2 / s_n * 1e3, 2),
                "pipeline_us": round(s_pipe / s_n * 1e3, 2),
                "achieved": round(alg_batch / sk / 1e9, 2),
                "frac": round(alg_batch / sk / 1e9 / HBM_PEAK_GBS, 5),
                "launches": s_n,
            }
        if g_n and groups_apply:
            gk = g_k1 / 1e3 / g_n
            out["roofline_one_group_in_flight"] = {
                "batches_per_launch": Geff,
                "kernel_us": round(gk * 1e6, 2),
                "rest_of_pipeline_us": round(g_k2 / g_n * 1e3, 2),
                "pipeline_us": round(g_pipe / g_n * 1e3, 2),
                "achieved": round(alg_batch * Geff / gk / 1e9, 2),
                "frac": round(alg_batch * Geff / gk / 1e9 / HBM_PEAK_GBS, 5),
                "launches": g_n,
            }
        if ungrouped is not None:
            out["one_launch_set_per_step"] = {
                "value": round(total_bytes * K / ungrouped / 1e9, 3), "unit": "GB/s",
                "ms_per_step": round(ungrouped / K * 1e3, 5),
                "what": "the same block with acm_scan_set_max_group(1): three kernel launches per step instead of per "
                        "group of up to %d steps" % G}
        if e2e is not None:
            out["e2e_with_h2d"] = e2e
        if cpu is not None:
            out["cpu_baseline"] = cpu
    if pool is not None:
        pool.close()
    matcher.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps steps; the median is reported")
    ap.add_argument("--workload", default="clamav2000", choices=["clamav2000", "clamav10000", "clamav15000", "sentiment"])
    ap.add_argument("--sub", default="clamav15000,sentiment",
                    help="workloads reported as sub-records of the line (N = 1 only); '' for none")
    ap.add_argument("--texts", type=int, default=64,
                    help="distinct 32 MiB texts the steps rotate over (64 = 2 GiB: no text is read twice within 256 MiB of traffic)")
    ap.add_argument("--check-texts", type=int, default=16,
                    help="the planes of the last this-many steps of the last block (distinct texts) are compared with the oracle")
    ap.add_argument("--sub-texts", type=int, default=0, help="texts of the sub-record workloads (0: as --texts)")
    ap.add_argument("--plant", type=int, default=4096)
    ap.add_argument("--workers", type=int, default=0,
                    help="HIP streams the steps are dealt over (0: automatic = 3)")
    ap.add_argument("--chain-bytes", type=int, default=0, help="chain pipeline: bytes per chain (0: automatic)")
    ap.add_argument("--group", type=int, default=16,
                    help="batches of one worker that go into one set of kernel launches (acm_scan_set_max_group): 1..16")
    ap.add_argument("--deal", default="even", choices=["groups", "even"],
                    help="a block shorter than workers x group steps: full groups first (groups) or equal shares (even)")
    ap.add_argument("--gather", default="torch", choices=["torch", "abi"],
                    help="N > 1: the block's planes go to rank 0 through torch.distributed's gather (RCCL) or through "
                         "the library's acm_gather_planes on an RCCL communicator of bench.py's own")
    ap.add_argument("--issue", default="native", choices=["native", "threads", "main"],
                    help="who enqueues the steps of a block: one acm_scan_batches_async call from the main thread, one "
                         "host thread per worker (each with one such call), or the main thread step by step")
    ap.add_argument("--mode", default="auto", choices=["auto", "chain", "sparse"],
                    help="scan pipeline (acm_scan_set_mode); auto = sparse when every signature has >= 3 bytes")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: 32 MiB per GPU (weak) or one 32 MiB text cut N ways (strong)")
    ap.add_argument("--profile-every", type=int, default=1,
                    help="HIP events around the kernels of every K-th launch group (or step) of worker 0 in the last timed block")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="only the warm-up and the timed blocks: no one-batch / one-group / ungrouped side measurements "
                         "(for a profiler run whose per-kernel averages should be those of the timed region)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    args = ap.parse_args()

    # one hardware queue per batch in flight: with HIP's default of 4 queues per process the
    # streams of the workers (+ torch's own) share queues and serialise
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
        args.gpus = world

    import torch
    import torch.distributed as dist
    # ACM_BENCH_BACKEND=gloo: rehearsal on a one-GPU box (every rank on cuda:0, gather through the
    # host).  ACM_BENCH_NCCL1=1: initialise RCCL even at world size 1, so that the collective path
    # runs where only one GPU is available.
    backend = os.environ.get("ACM_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1 or os.environ.get("ACM_BENCH_NCCL1"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from gpu_pattern_matching_amd import build
    build.build()          # no-op when libacmatch.so is current; raises if it cannot be built
    ctx = {"rank": rank, "world": world, "local_rank": local_rank, "dev": torch.device("cuda", local_rank),
           "backend": backend, "args": args, "rccl_comm": None}
    if args.gather == "abi" and dist.is_initialized() and backend == "nccl":
        ctx["rccl_comm"] = rccl_communicator(torch, dist, rank, world, ctx["dev"])

    head = run_workload(ctx, Workload(args.workload, args.plant), args.steps, args.warmup, args.repeats,
                        max(1, args.texts), args.workers, not args.no_verify, True)
    subs = {}
    if world == 1 and args.sub:
        for name in args.sub.split(","):
            if not name or name == args.workload:
                continue
            rec = run_workload(ctx, Workload(name, args.plant), args.steps, args.warmup, args.repeats,
                               max(1, args.sub_texts or args.texts), args.workers, not args.no_verify, False)
            if rec is not None:
                subs[name] = rec

    bad = False
    if rank == 0:
        out = {
            "metric": "input_GB_per_s_scanned",
            "value": head["value"],
            "unit": "GB/s",
            "n_gpus": world,
            "steps": head["steps"],
            "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "repeats": max(1, args.repeats),
        }
        out.update({k: v for k, v in head.items() if k not in ("value", "ms_per_step")})
        if subs:
            out["sub_records"] = subs
        print(json.dumps(out), flush=True)
        bad = head["parity"] == "MISMATCH" or any(s["parity"] == "MISMATCH" for s in subs.values())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if bad:
        sys.exit(1)


if __name__ == "__main__":
    main()
