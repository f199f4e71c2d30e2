#!/usr/bin/env python3
"""bench.py -- the reference's headline workload on MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): input GB/s scanned (+ matches/s), 32 MB text x N ClamAV signatures.
A step = one pass of the scan pipeline (walk -> probe -> resolve -> prefix sum -> scatter) over
one 32 MiB batch already resident in HBM (for signature sets whose shortest pattern has 3 bytes,
as here, the library's sparse pipeline: trigram filter -> candidate walks -> prefix max -> count ->
scatter; --mode chain forces the general one).  Like the reference, which keeps -w worker threads in
flight on one device, each with its own queue and buffers (ocl_aho_grep.c:37-144, :498-502),
steps are issued round-robin on --workers HIP streams with private scratch, so the
latency-bound tail of one batch overlaps the walk of the next.  With N > 1 GPUs the logical
text is N x 32 MiB, rank g scans shard g (+ an (L-1)-byte halo), the DFA is replicated and
the compact match planes of every group of --gather-rounds x --workers steps go to rank 0 in one
RCCL gather (double-buffered: it overlaps the next group's scans).
Weak scaling: 32 MiB per GPU.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SHARD = 32 << 20
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--sigs", type=int, default=2000, choices=[2000, 10000, 15000])
    ap.add_argument("--max-len", type=int, default=-1, help="-m pattern length limit")
    ap.add_argument("--chain", type=int, default=0, help="chain bytes (0 = auto)")
    ap.add_argument("--plant", type=int, default=4096)
    ap.add_argument("--workers", type=int, default=4, help="batches in flight (HIP streams)")
    ap.add_argument("--chain-walks", action="store_true",
                    help="chain the walk kernels of consecutive batches with events (acm_scan_batch_async); "
                         "measured slower than letting the streams run free on MI355X")
    ap.add_argument("--mode", default="auto", choices=["auto", "chain", "sparse"],
                    help="scan pipeline (acm_scan_set_mode); auto = sparse when every signature has >= 3 bytes")
    ap.add_argument("--graphs", action="store_true",
                    help="replay the HIP graph the library captures for a repeating batch instead of "
                         "launching every kernel of every step (acm_scan_set_graphs; measured neutral)")
    ap.add_argument("--profile-every", type=int, default=8,
                    help="record the library's HIP events (kernel durations for the roofline) on every K-th "
                         "timed step; those steps are launched kernel by kernel, the others replay the graph")
    ap.add_argument("--gather-rounds", type=int, default=4,
                    help="N > 1: rounds of --workers steps whose match planes travel to rank 0 in one gather")
    ap.add_argument("--cpl", type=int, default=0, help="chains per lane in the walk (2 or 4; 0 = default)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    args = ap.parse_args()

    # one hardware queue per batch in flight: with HIP's default of 4 queues per process the
    # streams of 4 workers (+ torch's own) share queues and serialise (measured: 43 vs 27 us/step)
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
    # rehearsal mode for a one-GPU box: ACM_BENCH_BACKEND=gloo puts every rank on cuda:0 and
    # moves the gather through host memory; the real multi-GPU run uses RCCL ("nccl")
    backend = os.environ.get("ACM_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import synth  # tests/synth.py: seeded corpus
    from gpu_pattern_matching_amd import Automaton, Matcher, build, sharding
    from gpu_pattern_matching_amd._lib import check
    build.build()          # no-op when libacmatch.so is current; raises if it cannot be built
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.current_stream().cuda_stream

    # ---- automaton: first --sigs ClamAV signatures (clamav_sample_sigs/<N>.txt are prefixes) ----
    sig_path = os.path.join(ROOT, "tests", "data", "clamav", "15000.txt")
    pats = synth.load_hex_patterns(sig_path, args.sigs, args.max_len)
    t0 = time.perf_counter()
    aut = Automaton()
    for i, p in enumerate(pats):
        aut.add(p, i)
    aut.compile()
    t_compile = time.perf_counter() - t0
    L = aut.max_pattern_len
    states = aut.num_states
    matcher = Matcher(aut, local_rank, max_text=16, plane_capacity=2, stream=stream)
    matcher.set_chain_bytes(args.chain)
    matcher.set_mode(args.mode)
    graphs = matcher.set_graphs(args.graphs)
    if args.cpl:
        matcher.set_chains_per_lane(args.cpl)
    aut.close()
    log(rank, "automaton: %d sigs, %d states, L=%d, compile %.2fs, device %.1f MB, hot rows %d" % (
        len(pats), states, L, t_compile, matcher.device_bytes / 1e6, matcher.hot_rows))

    # ---- text: logical text = world shards of 32 MiB; shard r = seeded corpus(seed 7 + r) ------
    def shard_text(r):
        return synth.clamav_corpus(SHARD, 7 + r, pats, args.plant)

    total_bytes = SHARD * world
    plan = sharding.shard_plan(total_bytes, world, rank, L)
    mine = shard_text(rank)
    if plan["halo"]:
        mine = np.concatenate([shard_text(rank - 1)[-plan["halo"]:], mine])
    n_local = mine.size
    assert n_local == plan["load_bytes"]
    d_text = torch.zeros((n_local + 15) // 16 * 16, dtype=torch.uint8, device=dev)
    d_text[:n_local] = torch.from_numpy(mine).to(dev)
    ws_bytes = matcher.lib.acm_scan_workspace_bytes(matcher.dfa, n_local)
    cap = 1 << 14                                   # cells per plane; 3.8k records expected
    while cap < 2 * args.plant + 1024:
        cap *= 2
    W = max(1, args.workers)

    class Worker:
        def __init__(self):
            self.stream = torch.cuda.Stream(device=dev)
            self.ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            self.scanned = torch.cuda.Event()       # recorded behind the worker's latest scan

    workers = [Worker() for _ in range(W)]
    # Match planes of a group of G = --gather-rounds x W consecutive steps sit in one tensor and
    # travel to rank 0 in ONE gather: a collective per step would cost more host time than the
    # scan it follows.  Two such tensors, so the gather of one group overlaps the scans of the next.
    G = W * max(1, args.gather_rounds)
    planes = [torch.zeros((G, 2, cap), dtype=torch.int32, device=dev) for _ in range(2)]
    gathered = [[torch.empty((G, 2, cap), dtype=torch.int32, device=dev) for _ in range(world)]
                for _ in range(2)] if (world > 1 and rank == 0) else [None, None]
    gather_stream = torch.cuda.Stream(device=dev)
    gather_done = [None, None]                      # event behind the last gather of each tensor
    open_group = {"buf": None, "slots": set()}      # scans issued but not gathered yet
    # the walk kernels of consecutive batches are chained by events (batch k+1's walk starts when
    # batch k's walk is done): only the walks serialise, everything behind them overlaps
    walk_done = []
    for _ in range(2 * W):
        e = C.c_void_p()
        check(matcher.lib.acm_rt_event_create(C.byref(e)), "acm_rt_event_create")
        walk_done.append(e)
    issued = [0]
    torch.cuda.synchronize()

    def flush():
        """Gather the planes of the group that is open (all of its scans are enqueued)."""
        buf = open_group["buf"]
        open_group["buf"], open_group["slots"] = None, set()
        if buf is None or world == 1:
            return
        if backend == "nccl":
            for wk in workers:                      # behind every worker's latest scan
                gather_stream.wait_event(wk.scanned)
            with torch.cuda.stream(gather_stream):
                dist.gather(planes[buf], gather_list=gathered[buf] if rank == 0 else None, dst=0)
                done = torch.cuda.Event()
                done.record(gather_stream)
            gather_done[buf] = done
        else:   # rehearsal: through the host
            for wk in workers:
                wk.stream.synchronize()
            host = planes[buf].cpu()
            bufs = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
            dist.gather(host, gather_list=bufs, dst=0)
            if rank == 0:
                for g, h in zip(gathered[buf], bufs):
                    g.copy_(h)

    def step(k):
        w, buf, slot = k % W, (k // G) & 1, k % G
        wk = workers[w]
        if open_group["buf"] is not None and (open_group["buf"] != buf or slot in open_group["slots"]):
            flush()                                 # a new group begins
        if gather_done[buf] is not None:            # the gather that last read this tensor:
            wk.stream.wait_event(gather_done[buf])  # the worker's stream waits for it, not the host
        p = planes[buf][slot]
        i = issued[0]
        issued[0] += 1
        chain = args.chain_walks and W > 1
        wait = walk_done[(i - 1) % len(walk_done)] if (chain and i > 0) else None
        matcher.scan_async(d_text, n_local, 0, wk.stream.cuda_stream, p[0], p[1], cap, halo=plan["halo"],
                           offset_shift=plan["offset_shift"], workspace=(wk.ws, ws_bytes),
                           wait_before_walk=wait,
                           record_after_walk=walk_done[i % len(walk_done)] if chain else None)
        if world > 1:
            wk.scanned.record(wk.stream)
        open_group["buf"] = buf
        open_group["slots"].add(slot)
        if len(open_group["slots"]) == G:
            flush()

    def drain():
        flush()
        gather_stream.synchronize()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    drain()
    fence()
    pe = max(1, args.profile_every)
    t0 = time.perf_counter()
    for k in range(args.steps):
        if k % pe == 0:
            matcher.profile(True)
            step(k)
            matcher.profile(False)
        else:
            step(k)
    t_issued = time.perf_counter() - t0      # host time to enqueue everything (not a result)
    drain()
    fence()
    elapsed = time.perf_counter() - t0
    k1_ms, k2_ms, pipe_ms, launches = matcher.profile_read()
    # the same kernels with ONE batch in flight (outside the timed region): how long the walk
    # takes when it has the GPU to itself, for reading the roofline beside the shared figure
    solo_k1_ms = solo_k2_ms = solo_pipe_ms = 0.0
    solo_n = 0
    matcher.profile(True)
    if W > 1:
        for k in range(0, 10 * W, W):
            step(k)
            drain()
            torch.cuda.synchronize()
        solo_k1_ms, solo_k2_ms, solo_pipe_ms, solo_n = matcher.profile_read()
    matcher.profile(False)
    path = matcher.path_taken(n_local, workers[0].stream.cuda_stream, workspace=(workers[0].ws.data_ptr(), ws_bytes))

    red_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- results of the last step -------------------------------------------------------------
    k_last = max(args.steps - 1, 0)
    slot_last, last = k_last % G, (k_last // G) & 1
    local = planes[last][slot_last].cpu().numpy()
    m_local = int(local[0, 0])
    m_total = m_local
    if world > 1:
        t = torch.tensor([m_local], dtype=torch.int64, device=red_dev)
        dist.all_reduce(t)
        m_total = int(t.item())

    out = None
    if rank == 0:
        if world > 1:
            offs, pids, last_state = sharding.merge_gathered([g[slot_last] for g in gathered[last]])
        else:
            offs, pids, last_state = sharding.merge_gathered([local])
        assert offs.size == m_total

        parity = "not checked"
        cpu = None
        if not args.no_verify or not args.no_cpu_baseline:
            import orc  # tests/orc.py -> oracle/ (checker + CPU baseline only)
            o = orc.Oracle()
            for i, p in enumerate(pats):
                o.add(p, i)
            o.compile()
            whole = mine if world == 1 else np.concatenate([shard_text(r) for r in range(world)])
            if not args.no_verify:
                epos, epat, elast = o.scan(whole)
                ok = (np.array_equal(epos, offs) and np.array_equal(epat, pids) and elast == last_state)
                parity = "bit-exact vs oracle serial scan (%d records, final state %d)" % (
                    epos.size, elast) if ok else "MISMATCH"
                if not ok:
                    log(rank, "PARITY MISMATCH: gpu %d records / oracle %d" % (offs.size, epos.size))
            if world == 1 and not args.no_cpu_baseline:
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
                cpu = {"value": round(SHARD / best / 1e9, 4), "unit": "GB/s", "cores": 1,
                       "kind": "port",
                       "sample": "oracle/acref.c serial walk of the reference-format int32 table over "
                                 "the same 32 MiB text, best of %d passes (%.1f s of CPU work)" % (
                                     reps, time.perf_counter() - t_start),
                       "all_cores": {"value": round(SHARD / t_all / 1e9, 4), "cores": ncpu}}
            o.close()

        # The roofline is quoted for the longest kernel of the pipeline that ran.  Every kernel of a
        # pipeline is a stage over the same N text positions, so the units one launch processes are
        # the N bytes of the batch and the algorithmic bytes are SURVEY 8(d)'s per-scan figure,
        # N x 1 B of text + 8 B per record, whichever stage is the slowest (chain: k_spec_walk reads
        # the text; sparse: k_sparse_filter reads the text and hands one candidate bit per byte to
        # k_sparse_walk, which is latency bound on a few ten thousand dependent table walks).
        L1 = max(launches, 1)
        alg_bytes = n_local + 8 * m_local
        if path == "chain":
            kernels = [("k_spec_walk", k1_ms, solo_k1_ms)]
        else:
            kernels = [("k_sieve", k1_ms, solo_k1_ms), ("k_sieve_emit", k2_ms, solo_k2_ms)]
        kname, kms, ksolo_ms = max(kernels, key=lambda t: t[1])
        walk_s = kms / 1e3 / L1
        achieved = alg_bytes / walk_s / 1e9 if walk_s > 0 else 0.0
        value = total_bytes * args.steps / elapsed / 1e9
        # HBM bytes per launch of the walk kernel from the PMC passes of this same command
        # (tests/run_pmc.sh -> tests/pmc_summarize.py -> profiles/r1_traffic.json); counters
        # cannot be collected inside a timed run, so this is read back, never estimated
        traffic = None
        stage_traffic = {}
        tfile = os.path.join(ROOT, "profiles", "r1_traffic_chain.json" if path == "chain" else "r1_traffic.json")
        if os.path.exists(tfile) and args.sigs == 2000 and args.max_len < 0:
            for name, rec in json.load(open(tfile)).items():
                for k in kernels:
                    if k[0] in name:
                        stage_traffic[k[0]] = round(rec["hbm_bytes_per_launch"], 1)
            traffic = stage_traffic.get(kname)
        out = {
            "metric": "input_GB_per_s_scanned",
            "value": round(value, 3),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "32 MiB per GPU of seeded uniform bytes + %d planted signatures x first %d "
                            "ClamAV sigs%s (%d states, max len %d); scan -> ordered compact "
                            "(offset, pattern) planes, gathered to rank 0" % (
                                args.plant, args.sigs,
                                "" if args.max_len < 0 else " (-m %d)" % args.max_len, states, L),
                "text_bytes_per_gpu": SHARD,
                "signatures": args.sigs,
                "pipeline": path,
                "hip_graphs": graphs,
                "chain_bytes": matcher.set_chain_bytes(args.chain) or "auto",
                "batches_in_flight": W,
                "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                "parallelism": "text sharded %d-way, DFA replicated" % world,
            },
            "host_enqueue_us_per_step": round(t_issued / args.steps * 1e6, 2),
            "matches_per_step": m_total,
            "matches_per_s": round(m_total * args.steps / elapsed, 1),
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
                "kernel_us": round(walk_s * 1e6, 2),
                "pipeline_us": round(pipe_ms / max(launches, 1) * 1e3, 2),
                "kernels_us": {k[0]: round(k[1] / L1 * 1e3, 2) for k in kernels},
                "stages": [{"kernel": k[0], "us": round(k[1] / L1 * 1e3, 2),
                            "achieved": round(alg_bytes / (k[1] / 1e3 / L1) / 1e9, 1) if k[1] > 0 else None,
                            "frac": round(alg_bytes / (k[1] / 1e3 / L1) / 1e9 / HBM_PEAK_GBS, 4) if k[1] > 0 else None,
                            "traffic": stage_traffic.get(k[0])} for k in kernels],
                "launches_timed": launches,
                "note": "HIP events on the batch's own stream around the kernels of every %d-th step of the "
                        "timed region; with %d batches in flight a kernel shares the GPU with the other "
                        "batches' kernels" % (pe, W),
            },
        }
        if solo_n:
            sw = ksolo_ms / 1e3 / solo_n
            out["roofline_one_batch_in_flight"] = {
                "kernel_us": round(sw * 1e6, 2),
                "kernels_us": {k[0]: round(k[2] / solo_n * 1e3, 2) for k in kernels},
                "pipeline_us": round(solo_pipe_ms / solo_n * 1e3, 2),
                "achieved": round(alg_bytes / sw / 1e9, 2),
                "frac": round(alg_bytes / sw / 1e9 / HBM_PEAK_GBS, 5),
                "launches": solo_n,
            }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    matcher.close()
    if out is not None and out["parity"] == "MISMATCH":
        sys.exit(1)


if __name__ == "__main__":
    main()
