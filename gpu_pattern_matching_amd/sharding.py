"""Multi-GPU host logic: shard a byte stream over ranks, gather the records.

The reference is single-device (ocl_aho_grep.c:498-499 hands the same -D to
every worker); this is the MI355X-native extension SURVEY section 8(e) describes:

  * rank g owns text[g*N/G, (g+1)*N/G); it also reads the max_pattern_len-1
    bytes in front of its range (the halo), starts there from state 0 and drops
    records that end inside the halo => identical to the serial scan, because
    the DFA state depends only on the last max_pattern_len bytes;
  * the DFA is replicated, no data-path collective during the scan;
  * one exchange: every rank contributes its compact planes (count in cell 0)
    to rank 0 with a fixed-capacity gather over RCCL/xGMI (torch.distributed
    backend "nccl"; "gloo" on CPU in the tests).  Ranks are position ordered,
    so concatenating the per-rank records gives the globally ordered list.

Pure host logic + torch.distributed plumbing; the scan itself is passed in.
"""
import numpy as np


def shard_range(n, world, rank):
    """[begin, end) of rank's share of an n-byte text."""
    return n * rank // world, n * (rank + 1) // world


def halo_bytes(max_pattern_len, begin):
    """bytes of left context rank needs: max_pattern_len - 1, clipped at the text start."""
    return min(max(max_pattern_len - 1, 0), begin)


def shard_plan(n, world, rank, max_pattern_len):
    """dict describing what rank loads and how its offsets map back to the whole text."""
    begin, end = shard_range(n, world, rank)
    halo = halo_bytes(max_pattern_len, begin)
    return {
        "begin": begin, "end": end, "halo": halo,
        "load_begin": begin - halo,          # first byte the rank needs
        "load_bytes": end - begin + halo,    # bytes scanned (halo included)
        "offset_shift": begin - halo,        # local offset + shift = global offset
    }


def gather_planes(planes, dist, dst=0, group=None):
    """Gather every rank's planes tensor [2, cap] int32 to dst; returns list (dst) or None."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return [planes]
    import torch
    bufs = [torch.empty_like(planes) for _ in range(world)] if rank == dst else None
    dist.gather(planes, gather_list=bufs, dst=dst, group=group)
    return bufs


def merge_gathered(gathered):
    """Concatenate per-rank compact planes (host side, rank order = position order).

    gathered: list of array-likes [2, cap] (row 0 = pattern plane, row 1 = offset plane,
    cell 0 = count, cell count+1 = last state).  Returns (offsets, patterns, last_state).
    """
    offs, pats, last = [], [], 0
    for g in gathered:
        a = np.asarray(g.cpu() if hasattr(g, "cpu") else g)
        m = int(a[0, 0])
        if m + 2 > a.shape[1]:
            raise OverflowError("rank contributed %d records but planes hold %d" % (m, a.shape[1] - 2))
        pats.append(a[0, 1:1 + m])
        offs.append(a[1, 1:1 + m])
        last = int(a[0, m + 1])
    return (np.concatenate(offs).astype(np.uint32) if offs else np.zeros(0, np.uint32),
            np.concatenate(pats).astype(np.int32) if pats else np.zeros(0, np.int32), last)
