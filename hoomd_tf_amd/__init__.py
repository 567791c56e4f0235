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
from . import standin
from .simmodel import (SimModel, compute_nlist_forces, compute_positions_forces, nlist_rinv, safe_norm,
                       box_size, wrap_vector, compute_rdf, masked_nlist, reduce_sum, pairwise_unit_forces, Nlist,
                       norm, cast, divide_no_nan, Positions, sort, exp, log, tanh, sqrt, square, pow, abs, minimum, maximum, where,
                       gather, equal, not_equal, erf, erfc, sigmoid, softplus, sin, cos,
                       MolSimModel, find_molecules, MeanTensor)
from .layers import RBFExpansion, WCARepulsion, EDSLayer, PairMLP, SoftRDFCV, LJLayer, Dense
from . import optimizers
from .tensorflowcompute import tfcompute

__version__ = "0.1.0"
