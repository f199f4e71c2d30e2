"""ctypes binding of libacmatch.so (the C ABI declared in include/acmatch.h).

The library is the product: there is no Python or CPU fallback.  If the shared
object is missing, ``load()`` raises; if no HIP device is visible, every
device entry point returns ACM_ERR_NODEV and the wrappers raise ``AcmError``.
"""
import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "libacmatch.so")

ACM_OK = 0
ACM_ERR_CAPACITY = -8

_vp = C.c_void_p
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)


class ScanBatch(C.Structure):
    """struct acm_scan_batch (include/acmatch.h)."""
    _fields_ = [("d_text", _vp), ("n", C.c_size_t), ("halo", C.c_size_t), ("offset_shift", C.c_long),
                ("init_state", C.c_long), ("d_workspace", _vp), ("workspace_bytes", C.c_size_t),
                ("d_pat_plane", _vp), ("d_off_plane", _vp), ("plane_capacity", C.c_size_t),
                ("stream", _vp), ("wait_before_walk", _vp), ("record_after_walk", _vp), ("report", C.c_int),
                ("profile", C.c_int), ("d_init_plane", _vp), ("init_plane_capacity", C.c_size_t)]


class ShardPlan(C.Structure):
    """struct acm_shard_plan (include/acmatch.h)."""
    _fields_ = [("begin", C.c_size_t), ("end", C.c_size_t), ("halo", C.c_size_t), ("load_begin", C.c_size_t),
                ("load_bytes", C.c_size_t), ("offset_shift", C.c_long)]


REPORT_HEAD, REPORT_STATE = 0, 1


class AcmError(RuntimeError):
    def __init__(self, code, where, detail):
        super().__init__("%s: %s (code %d)" % (where, detail, code))
        self.code = code


# every exported symbol: name -> (restype, argtypes); tests check the .so
# exports exactly these (plus the reference-named layer below)
NATIVE_API = {
    "acm_last_error": (C.c_char_p, []),
    "acm_strerror": (C.c_char_p, [C.c_int]),
    "acm_version": (C.c_char_p, []),
    "acm_device_count": (C.c_int, []),
    "acm_automaton_new": (_vp, []),
    "acm_automaton_state_matches": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int32), C.c_int]),
    "acm_automaton_free": (None, [_vp]),
    "acm_automaton_add": (C.c_int, [_vp, C.c_char_p, C.c_int, C.c_int]),
    "acm_automaton_load_file": (C.c_int, [_vp, C.c_char_p, C.c_int, C.c_int]),
    "acm_automaton_compile": (C.c_int, [_vp]),
    "acm_automaton_num_patterns": (C.c_int, [_vp]),
    "acm_automaton_max_pattern_len": (C.c_int, [_vp]),
    "acm_automaton_byte_classes": (C.c_int, [_vp, _vp]),
    "acm_compact_selftest": (C.c_int, [_vp, C.c_uint32, _u32p]),
    "acm_compact_profile": (C.c_int, [_vp, _vp, C.c_size_t, C.POINTER(C.c_uint64)]),
    "acm_automaton_num_states": (C.c_int, [_vp]),
    "acm_automaton_reference_table_bytes": (C.c_size_t, [_vp]),
    "acm_automaton_export_reference_table": (C.c_int, [_vp, _i32p]),
    "acm_automaton_pattern": (C.c_int, [_vp, C.c_int, _i32p, _i32p, C.POINTER(_vp), _i32p]),
    "acm_automaton_state_output": (C.c_int, [_vp, C.c_int]),
    "acm_dfa_upload": (C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    "acm_dfa_release": (None, [_vp]),
    "acm_dfa_device_bytes": (C.c_size_t, [_vp]),
    "acm_dfa_hot_rows": (C.c_int, [_vp]),
    "acm_dfa_device": (C.c_int, [_vp]),
    "acm_scan_workspace_bytes": (C.c_size_t, [_vp, C.c_size_t]),
    "acm_scan_async": (C.c_int, [_vp, _vp, C.c_size_t, C.c_long, _vp, C.c_size_t, _vp, _vp,
                                 C.c_size_t, _vp]),
    "acm_scan_shard_async": (C.c_int, [_vp, _vp, C.c_size_t, C.c_size_t, C.c_long, C.c_long, _vp,
                                       C.c_size_t, _vp, _vp, C.c_size_t, _vp]),
    "acm_scan_batch_async": (C.c_int, [_vp, C.POINTER(ScanBatch)]),
    "acm_scan_batches_async": (C.c_int, [_vp, C.POINTER(ScanBatch), C.c_size_t]),
    "acm_scan_set_max_group": (C.c_int, [_vp, C.c_int]),
    "acm_scan_set_chain_bytes": (C.c_int, [_vp, C.c_int]),
    "acm_scan_set_chains_per_lane": (C.c_int, [_vp, C.c_int]),
    "acm_scan_kernel_count": (C.c_int, []),
    "acm_expand_workspace_bytes": (C.c_size_t, [C.c_size_t]),
    "acm_expand_matches_async": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp]),
    "acm_scan_set_mode": (C.c_int, [_vp, C.c_int]),
    "acm_scan_set_graphs": (C.c_int, [_vp, C.c_int]),
    "acm_scan_sparse_eligible": (C.c_int, [_vp]),
    "acm_scan_lds_resident": (C.c_int, [_vp]),
    "acm_scan_group_capable": (C.c_int, [_vp]),
    "acm_scan_path_taken": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "acm_scan_profile_enable": (C.c_int, [_vp, C.c_int]),
    "acm_scan_profile_read": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                        C.POINTER(C.c_int)]),
    "acm_exclusive_scan_workspace_bytes": (C.c_size_t, [C.c_size_t]),
    "acm_exclusive_scan_i32": (C.c_int, [_vp, _vp, C.c_size_t, _vp, _vp, C.c_size_t, _vp]),
    "acm_compact_buckets": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "acm_bitonic_sort_u32": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint, C.c_uint, C.c_uint, _vp]),
    "acm_bucketize": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, C.c_size_t, _vp]),
    "acm_pack_chunks": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    "acm_remap_offsets": (C.c_int, [_vp, C.c_size_t, _vp, _vp, C.c_int, _vp]),
    "acm_shard_plan_for": (C.c_int, [C.c_size_t, C.c_int, C.c_int, C.c_int, _vp]),
    "acm_gather_planes": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_size_t, _vp, _vp, _vp]),
    "acm_gather_planes_sized": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_size_t, _vp, _vp, _vp, _i32p, _vp]),
    "acm_merge_planes": (C.c_long, [_vp, _vp, C.c_int, C.c_size_t, _vp, _vp, C.c_size_t, _vp]),
    "acm_rt_set_device": (C.c_int, [C.c_int]),
    "acm_rt_malloc": (C.c_int, [C.POINTER(_vp), C.c_size_t]),
    "acm_rt_free": (C.c_int, [_vp]),
    "acm_rt_host_alloc": (C.c_int, [C.POINTER(_vp), C.c_size_t]),
    "acm_rt_host_free": (C.c_int, [_vp]),
    "acm_rt_memcpy_h2d": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "acm_rt_memcpy_d2h": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "acm_rt_memcpy_d2d": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "acm_rt_memset": (C.c_int, [_vp, C.c_int, C.c_size_t, _vp]),
    "acm_rt_stream_create": (C.c_int, [C.POINTER(_vp)]),
    "acm_rt_stream_destroy": (C.c_int, [_vp]),
    "acm_rt_stream_sync": (C.c_int, [_vp]),
    "acm_rt_device_sync": (C.c_int, []),
    "acm_rt_event_create": (C.c_int, [C.POINTER(_vp)]),
    "acm_rt_event_destroy": (C.c_int, [_vp]),
    "acm_rt_event_record": (C.c_int, [_vp, _vp]),
    "acm_rt_event_sync": (C.c_int, [_vp]),
    "acm_rt_event_elapsed_ms": (C.c_int, [_vp, _vp, C.POINTER(C.c_float)]),
    "acm_rt_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int),
                                     C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
}

# the reference's names (include/acmatch.h layer 2); signatures are bound in
# compat.py, here only the list the export test checks
REFERENCE_API = [
    "clinitctx",
    "acsm_new", "acsm_add_pattern", "acsm_compile", "acsm_gen_state_table",
    "acsm_get_patterns_table", "acsm_get_max_pattern_size", "acsm_get_states", "acsm_get_size",
    "acsm_cleanup", "acsm_free",
    "databuf_new", "databuf_add_fd", "databuf_add_fp", "databuf_add_chunk", "databuf_reset",
    "databuf_clear", "databuf_copy_host_to_device", "databuf_copy_device_to_host",
    "databuf_process_results", "databuf_free",
    "ocl_aho_match_init", "ocl_aho_match_close", "ocl_aho_match",
    "ocl_prefix_sum_init", "ocl_prefix_sum_close", "ocl_prefix_sum",
    "ocl_compact_array_init", "ocl_compact_array_close", "ocl_compact_array",
    "ocl_bitonic_sort_init", "ocl_bitonic_sort_close", "ocl_bitonic_sort",
]

_lib = None


def load():
    """Load libacmatch.so; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libacmatch.so is missing (%s). Build it with "
            "`python -m gpu_pattern_matching_amd.build`; there is no fallback path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in NATIVE_API.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, where):
    if code != ACM_OK:
        lib = load()
        detail = lib.acm_last_error().decode() or lib.acm_strerror(code).decode()
        raise AcmError(code, where, detail)
