"""The C-ABI library loads without a GPU and exports every symbol the header declares."""
import ctypes
import os
import re

from gpu_pattern_matching_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "acmatch.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b((?:acm|acsm|databuf|ocl|clinitctx)\w*)\s*\(", src))
    return names


def test_every_declared_symbol_is_exported(lib):
    names = declared_functions()
    assert len(names) > 70
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, "declared in include/acmatch.h but not exported: %s" % missing


def test_binding_tables_cover_the_header(lib):
    names = declared_functions()
    bound = set(_lib.NATIVE_API) | set(_lib.REFERENCE_API)
    assert names == bound, (sorted(names - bound), sorted(bound - names))


def test_loads_without_a_device(lib):
    assert lib.acm_version().startswith(b"acmatch")
    assert lib.acm_device_count() >= 0
    assert lib.acm_strerror(-4) == b"no usable device"


def test_compat_headers_are_shims():
    inc = os.path.join(ROOT, "include", "compat")
    for h in ("acsmx.h", "databuf.h", "ocl_aho_match.h", "ocl_context.h", "ocl_prefix_sum.h",
              "ocl_compact_array.h", "ocl_bitonic_sort.h"):
        text = open(os.path.join(inc, h)).read()
        assert '#include "../acmatch.h"' in text


def test_device_entry_points_fail_loudly_without_gpu(lib):
    """No CPU fallback: with no device the upload reports ACM_ERR_NODEV."""
    if lib.acm_device_count() > 0:
        return
    a = lib.acm_automaton_new()
    lib.acm_automaton_add(a, b"abc", 3, 0)
    assert lib.acm_automaton_compile(a) == 0
    h = ctypes.c_void_p()
    assert lib.acm_dfa_upload(a, 0, ctypes.byref(h)) == -4
    assert b"no HIP device" in lib.acm_last_error()
    lib.acm_automaton_free(a)
