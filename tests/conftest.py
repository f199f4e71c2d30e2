import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(HERE, "golden", "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def lib():
    """libacmatch.so, built in-tree if the sources are newer (hipcc cross-compiles on CPU)."""
    import shutil
    from gpu_pattern_matching_amd import _lib, build
    have_hipcc = bool(os.environ.get("HIPCC") or shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"))
    try:
        build.build()
    except Exception:
        # a stale binary must not turn a compile error green: fall back to an existing library
        # only where there is no compiler at all and nothing newer than it to compile
        if have_hipcc or not os.path.exists(_lib.LIB_PATH) or build._stale(build.LIB, build._deps()):
            raise
    return _lib.load()


@pytest.fixture(scope="session")
def gpu(lib):
    """Fail (not skip) when a gpu-marked test runs without a device."""
    n = lib.acm_device_count()
    assert n > 0, "gpu-marked test but no HIP device is visible"
    return 0
