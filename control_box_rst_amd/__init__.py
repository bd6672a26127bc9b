"""control_box_rst_amd -- MI355X-native NLP inner loop for control_box_rst's hypergraph OCPs.

Only the hot path lives here: ``csrc/`` (HIP kernels + the C-ABI of include/corbo_hip.h), ``capi`` (ctypes mirror of
that ABI), ``problems`` (descriptors of the BASELINE configurations), ``solver`` (host-side mirror of the
reference's ``LevenbergMarquardtSparse`` setters over the C-ABI) and ``adapter/`` (the C++ class that plugs the
C-ABI into the reference's ``NlpSolverInterface``).  There is no CPU fallback.
"""
from . import capi, problems  # noqa: F401

__all__ = ["capi", "problems"]
