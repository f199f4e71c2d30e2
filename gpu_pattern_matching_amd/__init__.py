"""gpu_pattern_matching_amd -- MI355X-native Aho-Corasick matcher.

The product is libacmatch.so (HIP/gfx950 kernels + C ABI, see include/acmatch.h);
this package builds it, binds it with ctypes and offers a thin Python front-end
for tests and benchmarks.  There is no CPU fallback.
"""
from ._lib import AcmError, load  # noqa: F401
from .api import Automaton, DeviceArray, Matcher  # noqa: F401

__all__ = ["AcmError", "load", "Automaton", "DeviceArray", "Matcher"]
