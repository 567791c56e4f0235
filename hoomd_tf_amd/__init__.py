"""hoomd_tf_amd -- MI355X-native drop-in for hoomd-tf's per-particle force/energy path.

Mirrors the reference's Python surface for that path (``hoomd.htf``): ``SimModel``,
``tfcompute``, ``compute_nlist_forces``, ``nlist_rinv``, ``safe_norm``,
``RBFExpansion``, ``WCARepulsion``, ``EDSLayer`` ...  All arithmetic runs in the
hand-written HIP kernels of ``libhtf_amd.so`` (C ABI: include/htf_amd.h).
"""
from . import _lib
from ._lib import NlistOverflowError, SkewedBoxError
from . import ops
from .ops import Potential, Context

__version__ = "0.1.0"
