"""bpmf_amd -- MI355X-native drop-in for the per-column Gibbs sampler of ExaScience/bpmf.

The compute path is libbpmf_hip.so (hand-written HIP for gfx950 behind the C ABI
of include/bpmf_hip.h); this package is the thin host-side mirror of the
reference's `struct Sys` interface (c++/bpmf.h:112-239) on top of it.  There is
no CPU fallback: without the library and a HIP device the sampler raises.
"""
from ._lib import load_library, library_path, BpmfHipError  # noqa: F401
from .engine import HipEngine  # noqa: F401
from .sys import Sys, HyperParams, gibbs  # noqa: F401

__all__ = ["load_library", "library_path", "BpmfHipError", "HipEngine", "Sys", "HyperParams", "gibbs"]
