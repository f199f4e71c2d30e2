"""ctypes front-ends for the checker libraries (test infrastructure only).

* ``Oracle``   -> oracle/liboracle.so   (our CPU restatement, oracle/acref.c)
* ``RefAcsmx`` -> oracle/_ref/libacsmx_ref.so (the reference's own acsmx.c,
  compiled where it lies; only buildable where /root/reference exists, the
  prebuilt .so travels to the GPU box)

Nothing under gpu_pattern_matching_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
DATA = os.path.join(ROOT, "tests", "data")

_u8p = C.POINTER(C.c_ubyte)
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)


def _np_ptr(a, ty):
    return a.ctypes.data_as(ty)


def build_oracle(force=False):
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    src = os.path.join(ORACLE_DIR, "acref.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "all"], stdout=subprocess.DEVNULL)
    return so


def build_ref():
    """Build oracle/_ref when the reference tree is present; else use prebuilt."""
    so = os.path.join(ORACLE_DIR, "_ref", "libacsmx_ref.so")
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)
    return so if os.path.exists(so) else None


_olib = None


def olib():
    global _olib
    if _olib is None:
        L = C.CDLL(build_oracle())
        L.orc_new.restype = C.c_void_p
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_add_pattern.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
        L.orc_compile.argtypes = [C.c_void_p]
        L.orc_gen_table.argtypes = [C.c_void_p]
        for f in ("orc_num_states", "orc_num_patterns", "orc_max_pattern_len"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = C.c_int
        L.orc_table.argtypes = [C.c_void_p]
        L.orc_table.restype = _i32p
        L.orc_next_rows.argtypes = [C.c_void_p]
        L.orc_next_rows.restype = _i32p
        L.orc_fail.argtypes = [C.c_void_p]
        L.orc_fail.restype = _i32p
        L.orc_head_index.argtypes = [C.c_void_p, C.c_int]
        L.orc_head_index.restype = C.c_int
        L.orc_match_list.argtypes = [C.c_void_p, C.c_int, _i32p, C.c_int]
        L.orc_match_list.restype = C.c_int
        L.orc_pattern_info.argtypes = [C.c_void_p, C.c_int, _i32p, _i32p, C.c_char_p, C.c_int]
        L.orc_pattern_info.restype = C.c_int
        L.orc_patterns_chain.argtypes = [C.c_void_p, _i32p]
        L.orc_table_digest.argtypes = [_i32p, C.c_int]
        L.orc_table_digest.restype = C.c_uint64
        L.orc_records_digest.argtypes = [_u32p, _i32p, C.c_size_t]
        L.orc_records_digest.restype = C.c_uint64
        L.orc_load_patterns.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
        L.orc_load_patterns.restype = C.c_int
        L.orc_scan_serial.argtypes = [_i32p, _u8p, C.c_size_t, C.c_long, _u32p, _i32p,
                                      C.c_size_t, C.POINTER(C.c_long)]
        L.orc_scan_serial.restype = C.c_size_t
        L.orc_scan_count.argtypes = [_i32p, _u8p, C.c_size_t, C.c_long, C.POINTER(C.c_long)]
        L.orc_scan_count.restype = C.c_size_t
        L.orc_scan_threads.argtypes = [_i32p, _u8p, C.c_size_t, C.c_long, C.c_int, C.c_int,
                                       _u32p, _i32p, C.c_size_t, C.POINTER(C.c_long)]
        L.orc_scan_threads.restype = C.c_size_t
        L.orc_scan_refkernel.argtypes = [_i32p, _u8p, _i32p, _i32p, _i32p, _i32p, C.c_uint,
                                         C.c_ulong, C.c_long, C.c_int, C.c_int]
        L.orc_exclusive_scan.argtypes = [_i32p, _i32p, C.c_size_t]
        L.orc_compact_array.argtypes = [_i32p, _i32p, _i32p, C.c_int, C.c_int]
        L.orc_bitonic_sort.argtypes = [_u32p, _u32p, C.c_uint, C.c_uint, C.c_uint]
        L.orc_bitonic_sort.restype = C.c_int
        L.orc_bucketize.argtypes = [_u32p, _i32p, C.c_size_t, _i32p, _i32p, C.c_int, C.c_int,
                                    C.c_long, _i32p, _i32p]
        L.orc_now.restype = C.c_double
        _olib = L
    return _olib


class Oracle:
    """The CPU restatement: build a DFA the acsmx way and walk it serially."""

    def __init__(self):
        self.L = olib()
        self.h = C.c_void_p(self.L.orc_new())
        self._table = None

    def close(self):
        if self.h:
            self.L.orc_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add(self, pat: bytes, iid: int):
        self.L.orc_add_pattern(self.h, pat, len(pat), iid)

    def load(self, path, hex=False, max_len=-1):
        n = self.L.orc_load_patterns(self.h, path.encode(), int(hex), int(max_len))
        if n < 0:
            raise ValueError("oracle: cannot load patterns from %s" % path)
        return n

    def compile(self):
        self.L.orc_compile(self.h)
        self.L.orc_gen_table(self.h)
        return self

    @property
    def num_states(self):
        return self.L.orc_num_states(self.h)

    @property
    def num_patterns(self):
        return self.L.orc_num_patterns(self.h)

    @property
    def max_pattern_len(self):
        return self.L.orc_max_pattern_len(self.h)

    def pattern(self, i):
        iid = C.c_int32()
        n = C.c_int32()
        buf = C.create_string_buffer(4096)
        self.L.orc_pattern_info(self.h, i, C.byref(iid), C.byref(n), buf, 4096)
        return bytes(buf.raw[: n.value]), iid.value

    def patterns(self):
        return [self.pattern(i) for i in range(self.num_patterns)]

    def table(self):
        """Reference-format table as an int32 view [nstates, 2, 256]."""
        if self._table is None:
            p = self.L.orc_table(self.h)
            self._table = np.ctypeslib.as_array(p, shape=(self.num_states, 2, 256))
        return self._table

    def head_index(self, s):
        return self.L.orc_head_index(self.h, s)

    def match_list(self, s):
        buf = np.zeros(64, dtype=np.int32)
        n = self.L.orc_match_list(self.h, s, _np_ptr(buf, _i32p), 64)
        return buf[: min(n, 64)].tolist()

    def patterns_chain(self):
        out = np.zeros(self.num_patterns, dtype=np.int32)
        self.L.orc_patterns_chain(self.h, _np_ptr(out, _i32p))
        return out

    def table_digest(self):
        return int(self.L.orc_table_digest(self.L.orc_table(self.h), self.num_states))

    def scan(self, text, init_state=0, cap=None):
        """Serial scan -> (pos u32[], pat i32[], final_state)."""
        t = np.frombuffer(text, dtype=np.uint8) if not isinstance(text, np.ndarray) else text
        t = np.ascontiguousarray(t, dtype=np.uint8)
        n = t.size
        cap = n if cap is None else cap
        pos = np.empty(max(cap, 1), dtype=np.uint32)
        pat = np.empty(max(cap, 1), dtype=np.int32)
        fs = C.c_long(0)
        m = self.L.orc_scan_serial(self.L.orc_table(self.h), _np_ptr(t, _u8p), n, init_state,
                                   _np_ptr(pos, _u32p), _np_ptr(pat, _i32p), cap, C.byref(fs))
        if m > cap:
            raise OverflowError("oracle scan: %d matches > cap %d" % (m, cap))
        return pos[:m].copy(), pat[:m].copy(), fs.value

    def scan_all(self, text, init_state=0, cap=None):
        """Serial scan reporting every pattern of each final state's match list."""
        t = np.ascontiguousarray(np.frombuffer(text, dtype=np.uint8) if not isinstance(text, np.ndarray)
                                 else text, dtype=np.uint8)
        cap = 8 * t.size + 16 if cap is None else cap
        pos = np.empty(cap, dtype=np.uint32)
        pat = np.empty(cap, dtype=np.int32)
        fs = C.c_long(0)
        self.L.orc_scan_serial_all.restype = C.c_size_t
        self.L.orc_scan_serial_all.argtypes = [C.c_void_p, _u8p, C.c_size_t, C.c_long, _u32p, _i32p,
                                               C.c_size_t, C.POINTER(C.c_long)]
        m = self.L.orc_scan_serial_all(self.h, _np_ptr(t, _u8p), t.size, init_state, _np_ptr(pos, _u32p),
                                       _np_ptr(pat, _i32p), cap, C.byref(fs))
        if m > cap:
            raise OverflowError("oracle scan_all: %d matches > cap %d" % (m, cap))
        return pos[:m].copy(), pat[:m].copy(), fs.value

    def scan_count(self, text, init_state=0):
        t = np.ascontiguousarray(text, dtype=np.uint8)
        fs = C.c_long(0)
        m = self.L.orc_scan_count(self.L.orc_table(self.h), _np_ptr(t, _u8p), t.size, init_state,
                                  C.byref(fs))
        return m, fs.value

    def scan_threads(self, text, nthreads, init_state=0, cap=None):
        t = np.ascontiguousarray(text, dtype=np.uint8)
        n = t.size
        cap = n if cap is None else cap
        pos = np.empty(max(cap, 1), dtype=np.uint32)
        pat = np.empty(max(cap, 1), dtype=np.int32)
        fs = C.c_long(0)
        m = self.L.orc_scan_threads(self.L.orc_table(self.h), _np_ptr(t, _u8p), n, init_state,
                                    self.max_pattern_len, nthreads, _np_ptr(pos, _u32p),
                                    _np_ptr(pat, _i32p), cap, C.byref(fs))
        return pos[:m].copy(), pat[:m].copy(), fs.value


def records_digest(pos, pat):
    pos = np.ascontiguousarray(pos, dtype=np.uint32)
    pat = np.ascontiguousarray(pat, dtype=np.int32)
    return int(olib().orc_records_digest(_np_ptr(pos, _u32p), _np_ptr(pat, _i32p), pos.size))


def exclusive_scan(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    out = np.empty_like(a)
    olib().orc_exclusive_scan(_np_ptr(a, _i32p), _np_ptr(out, _i32p), a.size)
    return out


def compact_array(src, prefix, length, max_results, dst_size):
    src = np.ascontiguousarray(src, dtype=np.int32)
    prefix = np.ascontiguousarray(prefix, dtype=np.int32)
    dst = np.zeros(dst_size, dtype=np.int32)
    olib().orc_compact_array(_np_ptr(dst, _i32p), _np_ptr(src, _i32p), _np_ptr(prefix, _i32p),
                             length, max_results)
    return dst


def bitonic_sort(key, val, batch, length, direction):
    key = np.array(key, dtype=np.uint32)
    val = np.array(val, dtype=np.uint32)
    rc = olib().orc_bitonic_sort(_np_ptr(key, _u32p), _np_ptr(val, _u32p), batch, length,
                                 direction)
    return rc, key, val


def bucketize(pos, pat, indices, sizes, max_results, last_state):
    pos = np.ascontiguousarray(pos, dtype=np.uint32)
    pat = np.ascontiguousarray(pat, dtype=np.int32)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    sizes = np.ascontiguousarray(sizes, dtype=np.int32)
    chunks = indices.size
    res = np.zeros(max_results * chunks + 1, dtype=np.int32)
    res2 = np.zeros(max_results * chunks + 1, dtype=np.int32)
    olib().orc_bucketize(_np_ptr(pos, _u32p), _np_ptr(pat, _i32p), pos.size,
                         _np_ptr(indices, _i32p), _np_ptr(sizes, _i32p), chunks, max_results,
                         last_state, _np_ptr(res, _i32p), _np_ptr(res2, _i32p))
    return res, res2


def refkernel_scan(table, data, indices, sizes, last_state, max_pat, max_results):
    """Emulate ahomatch.cl chunk semantics (documentation of F2/Q8 only)."""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    sizes = np.ascontiguousarray(sizes, dtype=np.int32)
    chunks = indices.size
    res = np.zeros(max_results * chunks + 1, dtype=np.int32)
    res2 = np.zeros(max_results * chunks + 1, dtype=np.int32)
    tab = np.ascontiguousarray(table, dtype=np.int32)
    olib().orc_scan_refkernel(_np_ptr(tab, _i32p), _np_ptr(data, _u8p), _np_ptr(indices, _i32p),
                              _np_ptr(sizes, _i32p), _np_ptr(res, _i32p), _np_ptr(res2, _i32p),
                              chunks, data.size, last_state, max_pat, max_results)
    return res, res2


# ---------------------------------------------------------------- reference

_rlib = None


def rlib():
    global _rlib
    if _rlib is None:
        so = build_ref()
        if so is None:
            return None
        L = C.CDLL(so)
        L.ref_new.restype = C.c_void_p
        L.ref_add_pattern.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
        L.ref_compile.argtypes = [C.c_void_p]
        for f in ("ref_num_states", "ref_max_pattern_len", "ref_num_patterns"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = C.c_int
        L.ref_row.argtypes = [C.c_void_p, C.c_int, _i32p]
        L.ref_fail.argtypes = [C.c_void_p, C.c_int]
        L.ref_head_index.argtypes = [C.c_void_p, C.c_int]
        L.ref_match_list.argtypes = [C.c_void_p, C.c_int, _i32p, C.c_int]
        L.ref_fill_table.argtypes = [C.c_void_p, _i32p]
        L.ref_scan_serial.argtypes = [C.c_void_p, _u8p, C.c_size_t, C.c_long, _u32p, _i32p,
                                      C.c_size_t, C.POINTER(C.c_long)]
        L.ref_scan_serial.restype = C.c_size_t
        L.ref_patterns_table.argtypes = [C.c_void_p, _i32p, _i32p, _i32p]
        L.ref_hex_to_bytes.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.ref_free.argtypes = [C.c_void_p]
        _rlib = L
    return _rlib


class RefAcsmx:
    """The reference's own acsmx.c (compiled unmodified) behind ref_harness.c."""

    def __init__(self):
        self.L = rlib()
        if self.L is None:
            raise RuntimeError("oracle/_ref not built (no /root/reference here)")
        self.h = C.c_void_p(self.L.ref_new())

    def add(self, pat: bytes, iid: int):
        self.L.ref_add_pattern(self.h, pat, len(pat), iid)

    def compile(self):
        self.L.ref_compile(self.h)
        return self

    @property
    def num_states(self):
        return self.L.ref_num_states(self.h)

    @property
    def max_pattern_len(self):
        return self.L.ref_max_pattern_len(self.h)

    @property
    def num_patterns(self):
        return self.L.ref_num_patterns(self.h)

    def table(self):
        t = np.zeros((self.num_states, 2, 256), dtype=np.int32)
        self.L.ref_fill_table(self.h, _np_ptr(t, _i32p))
        return t

    def match_list(self, s):
        buf = np.zeros(64, dtype=np.int32)
        n = self.L.ref_match_list(self.h, s, _np_ptr(buf, _i32p), 64)
        return buf[: min(n, 64)].tolist()

    def scan(self, text, init_state=0):
        t = np.ascontiguousarray(np.frombuffer(text, dtype=np.uint8)
                                 if not isinstance(text, np.ndarray) else text, dtype=np.uint8)
        cap = t.size
        pos = np.empty(max(cap, 1), dtype=np.uint32)
        pat = np.empty(max(cap, 1), dtype=np.int32)
        fs = C.c_long(0)
        m = self.L.ref_scan_serial(self.h, _np_ptr(t, _u8p), t.size, init_state,
                                   _np_ptr(pos, _u32p), _np_ptr(pat, _i32p), cap, C.byref(fs))
        return pos[:m].copy(), pat[:m].copy(), fs.value

    def patterns_table(self):
        n = self.num_patterns
        iid = np.zeros(n, dtype=np.int32)
        ln = np.zeros(n, dtype=np.int32)
        nxt = np.zeros(n, dtype=np.int32)
        self.L.ref_patterns_table(self.h, _np_ptr(iid, _i32p), _np_ptr(ln, _i32p),
                                  _np_ptr(nxt, _i32p))
        return iid, ln, nxt

    def hex_to_bytes(self, s: str) -> bytes:
        buf = C.create_string_buffer(4096)
        n = self.L.ref_hex_to_bytes(s.encode(), buf, 4096)
        return bytes(buf.raw[:n])


def table_digest(table):
    t = np.ascontiguousarray(table, dtype=np.int32)
    return int(olib().orc_table_digest(_np_ptr(t, _i32p), t.shape[0]))


# ------------------------------------------------------------ pattern sets

def pattern_set(name):
    """(path, hex, max_len) for the named fixture set."""
    sets = {
        "tests": (os.path.join(DATA, "ref_tests", "patterns.txt"), False),
        "tests1": (os.path.join(DATA, "ref_tests", "1", "patterns.txt"), False),
        "tests2": (os.path.join(DATA, "ref_tests", "2", "patterns.txt"), False),
        "tests3": (os.path.join(DATA, "ref_tests", "3", "patterns.txt"), False),
        "sentiment": (os.path.join(DATA, "sentiment", "patterns_categorical.txt"), False),
    }
    if name in sets:
        p, hx = sets[name]
        return p, hx
    raise KeyError(name)


def clamav_file(n, tmpdir):
    """First n ClamAV signatures (2000 and 10000 are line-prefixes of 15000)."""
    src = os.path.join(DATA, "clamav", "15000.txt")
    if n == 15000:
        return src
    dst = os.path.join(str(tmpdir), "clamav_%d.txt" % n)
    if not os.path.exists(dst):
        with open(src, "rb") as f, open(dst, "wb") as g:
            for i, line in enumerate(f):
                if i >= n:
                    break
                g.write(line)
    return dst
