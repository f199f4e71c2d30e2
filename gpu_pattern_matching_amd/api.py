"""Thin Python front-end over the native boundary (acm_* in include/acmatch.h).

Used by tests/ and bench.py.  Everything that computes runs in libacmatch.so
on the GPU; this module only moves bytes and keeps handles alive.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import AcmError, check  # noqa: F401


def _ptr(x):
    """device pointer of a DeviceArray / int / torch tensor (has data_ptr)."""
    if x is None:
        return None
    if isinstance(x, DeviceArray):
        return x.ptr
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return int(x)


class DeviceArray:
    """A hipMalloc'ed block; numpy in, numpy out."""

    def __init__(self, nbytes, device=None):
        self.lib = _lib.load()
        if device is not None:
            check(self.lib.acm_rt_set_device(device), "acm_rt_set_device")
        p = C.c_void_p()
        check(self.lib.acm_rt_malloc(C.byref(p), nbytes), "acm_rt_malloc")
        self.ptr = p.value
        self.nbytes = nbytes

    @classmethod
    def from_numpy(cls, a, pad_to=16, stream=None):
        a = np.ascontiguousarray(a)
        nb = (a.nbytes + pad_to - 1) // pad_to * pad_to if pad_to else a.nbytes
        d = cls(max(nb, 16))
        if nb > a.nbytes:
            check(d.lib.acm_rt_memset(d.ptr + a.nbytes, 0, nb - a.nbytes, stream), "acm_rt_memset")
        if a.nbytes:
            check(d.lib.acm_rt_memcpy_h2d(d.ptr, a.ctypes.data, a.nbytes, stream), "acm_rt_memcpy_h2d")
        check(d.lib.acm_rt_stream_sync(stream), "acm_rt_stream_sync")
        return d

    def to_numpy(self, dtype, count, offset_bytes=0, stream=None):
        out = np.empty(count, dtype=dtype)
        if out.nbytes:
            check(self.lib.acm_rt_memcpy_d2h(out.ctypes.data, self.ptr + offset_bytes, out.nbytes,
                                             stream), "acm_rt_memcpy_d2h")
        check(self.lib.acm_rt_stream_sync(stream), "acm_rt_stream_sync")
        return out

    def fill(self, byte, stream=None):
        check(self.lib.acm_rt_memset(self.ptr, byte, self.nbytes, stream), "acm_rt_memset")

    def free(self):
        if self.ptr:
            self.lib.acm_rt_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Automaton:
    """Host automaton: patterns -> acsmx-compatible DFA (acm_automaton_*)."""

    def __init__(self):
        self.lib = _lib.load()
        self.h = self.lib.acm_automaton_new()
        if not self.h:
            raise MemoryError("acm_automaton_new")

    def add(self, pattern: bytes, iid: int = 0):
        check(self.lib.acm_automaton_add(self.h, pattern, len(pattern), iid), "acm_automaton_add")

    def load_file(self, path, hex=False, max_len=-1):
        n = self.lib.acm_automaton_load_file(self.h, str(path).encode(), int(hex), int(max_len))
        if n < 0:
            check(n, "acm_automaton_load_file")
        return n

    def compile(self):
        check(self.lib.acm_automaton_compile(self.h), "acm_automaton_compile")
        return self

    @property
    def num_patterns(self):
        return self.lib.acm_automaton_num_patterns(self.h)

    @property
    def num_states(self):
        return self.lib.acm_automaton_num_states(self.h)

    @property
    def max_pattern_len(self):
        return self.lib.acm_automaton_max_pattern_len(self.h)

    def byte_classes(self):
        """(number of byte classes, byte -> class map as a 256-entry uint8 array); 256 classes: no compression"""
        m = np.zeros(256, dtype=np.uint8)
        return int(self.lib.acm_automaton_byte_classes(self.h, m.ctypes.data_as(C.c_void_p))), m

    def reference_table(self):
        """The reference-format table [states, 2, 256] int32 (acsmx.c:640-658)."""
        t = np.zeros((self.num_states, 2, 256), dtype=np.int32)
        check(self.lib.acm_automaton_export_reference_table(
            self.h, t.ctypes.data_as(C.POINTER(C.c_int32))), "export_reference_table")
        return t

    def pattern(self, i):
        iid, n, nxt = C.c_int32(), C.c_int32(), C.c_int32()
        p = C.c_void_p()
        check(self.lib.acm_automaton_pattern(self.h, i, C.byref(iid), C.byref(n), C.byref(p),
                                             C.byref(nxt)), "acm_automaton_pattern")
        data = C.string_at(p.value, n.value) if n.value else b""
        return data, iid.value, nxt.value

    def state_matches(self, ref_state):
        """Every pattern index ending where the walk enters ref_state, list order."""
        buf = (C.c_int32 * 4096)()
        n = self.lib.acm_automaton_state_matches(self.h, ref_state, buf, 4096)
        if n < 0:
            raise ValueError("bad state %r" % (ref_state,))
        return list(buf[:n])

    def state_output(self, ref_state):
        return self.lib.acm_automaton_state_output(self.h, ref_state)

    def close(self):
        if self.h:
            self.lib.acm_automaton_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Matcher:
    """Device DFA + scratch + result planes for texts up to max_text bytes."""

    def __init__(self, automaton, device=0, max_text=1 << 20, plane_capacity=None, stream=None):
        self.lib = _lib.load()
        self.device = device
        self.stream = stream
        h = C.c_void_p()
        check(self.lib.acm_dfa_upload(automaton.h, device, C.byref(h)), "acm_dfa_upload")
        self.dfa = h.value
        self.max_text = 0
        self.ws = self.pat_plane = self.off_plane = None
        self.reserve(max_text, plane_capacity)

    def reserve(self, max_text, plane_capacity=None):
        cap = plane_capacity if plane_capacity is not None else max_text + 2
        if max_text <= self.max_text and cap <= getattr(self, "plane_capacity", 0):
            return
        for b in (self.ws, self.pat_plane, self.off_plane):
            if b is not None:
                b.free()
        self.max_text = max_text
        self.plane_capacity = max(cap, 2)
        self.ws_bytes = self.lib.acm_scan_workspace_bytes(self.dfa, max_text)
        self.ws = DeviceArray(self.ws_bytes, self.device)
        self.pat_plane = DeviceArray(self.plane_capacity * 4)
        self.off_plane = DeviceArray(self.plane_capacity * 4)

    @property
    def hot_rows(self):
        return self.lib.acm_dfa_hot_rows(self.dfa)

    @property
    def device_bytes(self):
        return self.lib.acm_dfa_device_bytes(self.dfa)

    def set_chain_bytes(self, s):
        return self.lib.acm_scan_set_chain_bytes(self.dfa, s)

    def set_chains_per_lane(self, c):
        return self.lib.acm_scan_set_chains_per_lane(self.dfa, c)

    MODES = {"auto": 0, "chain": 1, "sparse": 2}
    PATHS = {1: "chain", 2: "sparse", 0xDEAD: "failed"}

    def set_mode(self, mode):
        """'auto' | 'chain' | 'sparse' (acm_scan_set_mode); returns the mode in use."""
        got = self.lib.acm_scan_set_mode(self.dfa, self.MODES[mode])
        return [k for k, v in self.MODES.items() if v == got][0]

    def set_graphs(self, enable):
        """Replay repeating scans as HIP graphs (acm_scan_set_graphs); returns the setting in use."""
        return bool(self.lib.acm_scan_set_graphs(self.dfa, int(enable)))

    def sparse_eligible(self):
        return bool(self.lib.acm_scan_sparse_eligible(self.dfa))

    def lds_resident(self):
        """the chain pipeline walks this set with the whole automaton in LDS (lds_walk.hip)"""
        return bool(self.lib.acm_scan_lds_resident(self.dfa))

    def group_capable(self):
        """consecutive batches of one size share their launches in the current mode (acm_scan_batches_async)"""
        return bool(self.lib.acm_scan_group_capable(self.dfa))

    def path_taken(self, n, stream=None, workspace=None):
        """Which pipeline produced the planes of the last n-byte scan (synchronises)."""
        st = stream if stream is not None else self.stream
        ws_ptr = workspace[0] if workspace is not None else self.ws.ptr
        rc = self.lib.acm_scan_path_taken(self.dfa, _ptr(ws_ptr), n, st)
        if rc < 0:
            check(rc, "acm_scan_path_taken")
        return self.PATHS[rc]

    def scan_async(self, d_text, n, init_state=0, stream=None, pat_plane=None, off_plane=None,
                   plane_capacity=None, halo=0, offset_shift=0, workspace=None, wait_before_walk=None,
                   record_after_walk=None, report=0):
        """Enqueue one scan of device text; nothing is synchronised.

        halo/offset_shift: shard form (acm_scan_shard_async).  workspace: (ptr, nbytes) of a
        caller-owned scratch block instead of the matcher's own.  wait_before_walk /
        record_after_walk: hipEvent_t handles chaining the walk kernels of batches that are in
        flight on different streams (acm_scan_batch_async).
        """
        if wait_before_walk is not None or record_after_walk is not None or report:
            if workspace is None and n > self.max_text:
                raise ValueError("text of %d bytes exceeds reserved %d" % (n, self.max_text))
            st = stream if stream is not None else self.stream
            ws_ptr, ws_bytes = workspace if workspace is not None else (self.ws.ptr, self.ws_bytes)
            b = _lib.ScanBatch(_ptr(d_text), n, halo, offset_shift, init_state, _ptr(ws_ptr), ws_bytes,
                               _ptr(pat_plane) if pat_plane is not None else self.pat_plane.ptr,
                               _ptr(off_plane) if off_plane is not None else self.off_plane.ptr,
                               plane_capacity if plane_capacity is not None else self.plane_capacity,
                               st, wait_before_walk, record_after_walk, report)
            check(self.lib.acm_scan_batch_async(self.dfa, C.byref(b)), "acm_scan_batch_async")
            return
        if workspace is None and n > self.max_text:
            raise ValueError("text of %d bytes exceeds reserved %d" % (n, self.max_text))
        st = stream if stream is not None else self.stream
        ws_ptr, ws_bytes = workspace if workspace is not None else (self.ws.ptr, self.ws_bytes)
        check(self.lib.acm_scan_shard_async(self.dfa, _ptr(d_text), n, halo, offset_shift, init_state,
                                      _ptr(ws_ptr), ws_bytes,
                                      _ptr(pat_plane) if pat_plane is not None else self.pat_plane.ptr,
                                      _ptr(off_plane) if off_plane is not None else self.off_plane.ptr,
                                      plane_capacity if plane_capacity is not None
                                      else self.plane_capacity, st), "acm_scan_async")

    def make_batch(self, d_text, n, stream, pat_plane, off_plane, plane_capacity, workspace, init_state=0, halo=0,
                   offset_shift=0, report=0, profile=False, init_plane=None, init_plane_capacity=0):
        """A reusable acm_scan_batch for enqueue(): a worker that scans with the same buffers over and
        over builds its batches once and pays one foreign call per scan."""
        return _lib.ScanBatch(_ptr(d_text), n, halo, offset_shift, init_state, _ptr(workspace[0]), workspace[1],
                              _ptr(pat_plane), _ptr(off_plane), plane_capacity, stream, None, None, report,
                              1 if profile else 0, _ptr(init_plane) if init_plane is not None else None,
                              init_plane_capacity)

    def enqueue(self, batch):
        rc = self.lib.acm_scan_batch_async(self.dfa, C.byref(batch))
        if rc:
            check(rc, "acm_scan_batch_async")

    def enqueue_many(self, batches):
        """acm_scan_batches_async: the batches in order with one foreign call; consecutive sparse
        batches of one size on one stream with their own workspaces and planes share their launches."""
        arr = (_lib.ScanBatch * len(batches))(*batches)
        rc = self.lib.acm_scan_batches_async(self.dfa, arr, len(batches))
        if rc:
            check(rc, "acm_scan_batches_async")

    def set_max_group(self, batches):
        return int(self.lib.acm_scan_set_max_group(self.dfa, int(batches)))

    def fetch(self, stream=None):
        """(offsets u32[], patterns i32[], last_state) of the last scan."""
        st = stream if stream is not None else self.stream
        head = self.pat_plane.to_numpy(np.int32, 1, stream=st)
        m = int(head[0])
        stored = min(m, self.plane_capacity - 2)
        pat = self.pat_plane.to_numpy(np.int32, stored + 2, stream=st)
        off = self.off_plane.to_numpy(np.int32, stored + 2, stream=st)
        if m > stored:
            raise AcmError(_lib.ACM_ERR_CAPACITY, "Matcher.fetch",
                           "%d matches but planes hold %d" % (m, stored))
        return off[1:1 + m].astype(np.uint32), pat[1:1 + m].copy(), int(pat[m + 1])

    def scan_all(self, text, init_state=0, out_capacity=None):
        """All-patterns reporting: scan with the final states in the pattern
        plane, expand every state's match list on the device (acm_expand_matches_async), download.
        Returns (offsets, patterns, last_state) with one record per pattern ending at each offset."""
        t = np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray)) \
            else np.ascontiguousarray(text, dtype=np.uint8)
        self.reserve(max(t.size, 1))
        d = DeviceArray.from_numpy(t, stream=self.stream)
        cap = out_capacity if out_capacity is not None else 8 * self.plane_capacity
        max_records = self.plane_capacity - 2
        ws_bytes = self.lib.acm_expand_workspace_bytes(max_records)
        ws, pat, off = DeviceArray(ws_bytes), DeviceArray(cap * 4), DeviceArray(cap * 4)
        try:
            self.scan_async(d, t.size, init_state, report=_lib.REPORT_STATE)
            check(self.lib.acm_expand_matches_async(self.dfa, self.pat_plane.ptr, self.off_plane.ptr, max_records,
                                                    pat.ptr, off.ptr, cap, ws.ptr, ws_bytes, self.stream),
                  "acm_expand_matches_async")
            m = int(pat.to_numpy(np.int32, 1, stream=self.stream)[0])
            if m > cap - 2:
                raise AcmError(_lib.ACM_ERR_CAPACITY, "Matcher.scan_all", "%d records but planes hold %d" % (m, cap - 2))
            p = pat.to_numpy(np.int32, m + 2, stream=self.stream)
            o = off.to_numpy(np.int32, m + 2, stream=self.stream)
            return o[1:1 + m].astype(np.uint32), p[1:1 + m].copy(), int(p[m + 1])
        finally:
            for b in (d, ws, pat, off):
                b.free()

    def scan(self, text, init_state=0):
        """Scan host bytes: upload, scan, download."""
        t = np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray)) \
            else np.ascontiguousarray(text, dtype=np.uint8)
        self.reserve(max(t.size, 1))
        d = DeviceArray.from_numpy(t, stream=self.stream)
        try:
            self.scan_async(d, t.size, init_state)
            return self.fetch()
        finally:
            d.free()

    def profile(self, enable):
        check(self.lib.acm_scan_profile_enable(self.dfa, int(enable)), "acm_scan_profile_enable")

    def profile_read(self):
        """(first kernel ms, second kernel ms, pipeline ms, launches) since the last read."""
        f, s2, p, n = C.c_double(), C.c_double(), C.c_double(), C.c_int()
        check(self.lib.acm_scan_profile_read(self.dfa, C.byref(f), C.byref(s2), C.byref(p), C.byref(n)),
              "acm_scan_profile_read")
        return f.value, s2.value, p.value, n.value

    def close(self):
        for b in (self.ws, self.pat_plane, self.off_plane):
            if b is not None:
                b.free()
        self.ws = self.pat_plane = self.off_plane = None
        if self.dfa:
            self.lib.acm_dfa_release(self.dfa)
            self.dfa = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# --------------------------------------------------------------- device ops

def exclusive_scan(values, stream=None):
    """int32 exclusive prefix sum on the device -> (scan, total)."""
    lib = _lib.load()
    a = np.ascontiguousarray(values, dtype=np.int32)
    d_in = DeviceArray.from_numpy(a, pad_to=0) if a.size else DeviceArray(16)
    d_out = DeviceArray(max(a.nbytes, 16))
    d_tot = DeviceArray(16)
    wsb = lib.acm_exclusive_scan_workspace_bytes(a.size)
    ws = DeviceArray(max(wsb, 16))
    check(lib.acm_exclusive_scan_i32(d_in.ptr, d_out.ptr, a.size, d_tot.ptr, ws.ptr, wsb, stream),
          "acm_exclusive_scan_i32")
    out = d_out.to_numpy(np.int32, a.size, stream=stream)
    tot = int(d_tot.to_numpy(np.int32, 1, stream=stream)[0])
    for d in (d_in, d_out, d_tot, ws):
        d.free()
    return out, tot


def compact_buckets(src, prefix, length, max_results, dst_cells, stream=None):
    lib = _lib.load()
    d_src = DeviceArray.from_numpy(np.ascontiguousarray(src, dtype=np.int32), pad_to=0)
    d_pre = DeviceArray.from_numpy(np.ascontiguousarray(prefix, dtype=np.int32), pad_to=0)
    d_dst = DeviceArray(dst_cells * 4)
    d_dst.fill(0)
    check(lib.acm_compact_buckets(d_dst.ptr, d_src.ptr, d_pre.ptr, length, max_results, stream),
          "acm_compact_buckets")
    out = d_dst.to_numpy(np.int32, dst_cells, stream=stream)
    for d in (d_src, d_pre, d_dst):
        d.free()
    return out


def bitonic_sort(keys, vals, batch, length, direction, stream=None):
    lib = _lib.load()
    k = np.ascontiguousarray(keys, dtype=np.uint32)
    v = np.ascontiguousarray(vals, dtype=np.uint32)
    d_k = DeviceArray.from_numpy(k, pad_to=0)
    d_v = DeviceArray.from_numpy(v, pad_to=0)
    d_ko = DeviceArray(max(k.nbytes, 16))
    d_vo = DeviceArray(max(v.nbytes, 16))
    check(lib.acm_rt_memcpy_d2d(d_ko.ptr, d_k.ptr, k.nbytes, stream), "d2d")
    check(lib.acm_rt_memcpy_d2d(d_vo.ptr, d_v.ptr, v.nbytes, stream), "d2d")
    rc = lib.acm_bitonic_sort_u32(d_ko.ptr, d_vo.ptr, d_k.ptr, d_v.ptr, batch, length, direction,
                                  stream)
    ko = d_ko.to_numpy(np.uint32, k.size, stream=stream)
    vo = d_vo.to_numpy(np.uint32, v.size, stream=stream)
    for d in (d_k, d_v, d_ko, d_vo):
        d.free()
    return rc, ko, vo


def bucketize(pat_plane, off_plane, indices, sizes, max_results, stream=None):
    lib = _lib.load()
    ind = np.ascontiguousarray(indices, dtype=np.int32)
    siz = np.ascontiguousarray(sizes, dtype=np.int32)
    chunks = ind.size
    d_p = DeviceArray.from_numpy(np.ascontiguousarray(pat_plane, dtype=np.int32), pad_to=0)
    d_o = DeviceArray.from_numpy(np.ascontiguousarray(off_plane, dtype=np.int32), pad_to=0)
    d_i = DeviceArray.from_numpy(ind, pad_to=0)
    d_s = DeviceArray.from_numpy(siz, pad_to=0)
    cells = max_results * chunks + 1
    d_r = DeviceArray(cells * 4)
    d_r2 = DeviceArray(cells * 4)
    d_r.fill(0)
    d_r2.fill(0)
    check(lib.acm_bucketize(d_p.ptr, d_o.ptr, d_i.ptr, d_s.ptr, chunks, max_results, d_r.ptr,
                            d_r2.ptr, min(np.asarray(pat_plane).size, np.asarray(off_plane).size), stream),
          "acm_bucketize")
    r = d_r.to_numpy(np.int32, cells, stream=stream)
    r2 = d_r2.to_numpy(np.int32, cells, stream=stream)
    for d in (d_p, d_o, d_i, d_s, d_r, d_r2):
        d.free()
    return r, r2


def device_info(device=0):
    lib = _lib.load()
    name = C.create_string_buffer(256)
    cus, lds = C.c_int(), C.c_int()
    mem = C.c_size_t()
    check(lib.acm_rt_device_info(device, name, 256, C.byref(cus), C.byref(mem), C.byref(lds)),
          "acm_rt_device_info")
    return {"name": name.value.decode(), "cus": cus.value, "mem_bytes": mem.value,
            "lds_per_cu": lds.value}
