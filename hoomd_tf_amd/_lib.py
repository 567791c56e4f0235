"""ctypes binding of ``libhtf_amd.so`` (the C ABI of ``include/htf_amd.h``).

This is the only place the package touches native code.  There is NO fallback: if
the HIP library is missing the import fails loudly (a product path that silently
ran on the CPU would void every parity claim).
"""
import ctypes as C
import os

# torch must be imported BEFORE libhtf_amd.so is loaded: torch ships its own
# libamdhip64.so.7 and dlopens it by path; loading ours first would pull the system HIP
# runtime as well and leave two runtimes in one process (the second one then sees no
# device).  With torch first, our DT_NEEDED libamdhip64.so.7 binds to the copy that is
# already loaded, so kernels, streams and tensors share one runtime.
import torch  # noqa: F401  (plumbing: device memory + streams)

_HERE = os.path.dirname(os.path.abspath(__file__))
# HTF_AMD_LIB: another build of the same ABI, for same-box A/B runs of kernel variants (tools/)
LIB_PATH = os.environ.get("HTF_AMD_LIB") or os.path.join(_HERE, "libhtf_amd.so")

HTF_OK, HTF_ERR_INVALID, HTF_ERR_DEVICE, HTF_ERR_NLIST_OVERFLOW, HTF_ERR_SKEWED_BOX, HTF_ERR_NOMEM = range(6)
HTF_F32, HTF_F64 = 0, 1
HTF_TF2HOOMD, HTF_HOOMD2TF = 0, 1
POT_NONE, POT_LJ, POT_WCA, POT_RINV_POLY, POT_SIMPLE, POT_PAIR_MLP, POT_GAUSS, POT_LJ_PARAM, POT_TOPK_MLP, POT_JIT = range(10)
JIT_READS_OWN_TYPE = 1
OPT_SGD, OPT_ADAM, OPT_NADAM = range(3)
OPT_STATE_FLOATS = 24
ACT_LINEAR, ACT_TANH = 0, 1
MLP_FP32, MLP_BF16, MLP_SPLIT, MLP_SPLIT16 = 0, 1, 2, 3
MAX_POLY_TERMS = 8


class NlistOverflowError(RuntimeError):
    """'Neighbor list is full!' -- tf.errors.InvalidArgumentError upstream
    (simmodel.py:220-224, test_tensorflow.py:830-848)."""


class SkewedBoxError(RuntimeError):
    """'box is skewed' -- tf.errors.InvalidArgumentError upstream (simmodel.py:195)."""


class Box(C.Structure):
    _fields_ = [("lo", C.c_double * 3), ("hi", C.c_double * 3), ("tilt", C.c_double * 3),
                ("periodic", C.c_int * 3)]


BRICK_MAX_MSG, BRICK_MAX_P = 8, 64   # include/htf_standin.h HTFS_BRICK_MAX_MSG / _MAX_P
BC_N_INT, BC_N_BND, BC_N_CAND, BC_N_ARRIVED, BC_FLAGS, BC_REBUILDS, BC_MSG, BC_CLASS, BC_SLOT, BC_WORDS = 0, 1, 2, 3, 4, 5, 8, 16, 64, 192
BF_LOST, BF_MIG_OVERFLOW, BF_INT_OVERFLOW, BF_BND_OVERFLOW, BF_GHOST_OVERFLOW, BF_HALO_TIMEOUT = 1, 2, 4, 8, 16, 32


class Brick(C.Structure):
    """htfs_brick (include/htf_standin.h)."""
    _fields_ = [("ndim", C.c_int), ("axis", C.c_int * 2), ("p", C.c_int * 2), ("me", C.c_int * 2), ("n_msg", C.c_int),
                ("replica", C.c_int), ("r_ghost", C.c_double), ("cap_int", C.c_uint), ("cap_bnd", C.c_uint),
                ("ghost_cap", C.c_uint * BRICK_MAX_MSG), ("ghost_off", C.c_uint * BRICK_MAX_MSG),
                ("mig_cap", C.c_uint * BRICK_MAX_MSG), ("mig_off", C.c_uint * BRICK_MAX_MSG),
                ("shift", (C.c_double * 3) * BRICK_MAX_MSG), ("mig_shift", (C.c_double * 3) * BRICK_MAX_MSG),
                ("halo_wrap", C.c_int), ("mig_wrap", C.c_int), ("box_lo", C.c_double * 3), ("box_L", C.c_double * 3)]


class Peer(C.Structure):
    """htfs_peer (transport "peer": the halo as direct stores into the neighbors' inboxes)."""
    _fields_ = [("inbox", C.c_void_p * BRICK_MAX_MSG), ("signal", C.c_void_p * BRICK_MAX_MSG), ("my_inbox", C.c_void_p),
                ("my_signal", C.c_void_p), ("state", C.c_void_p), ("spin_limit", C.c_uint)]


MIRROR_MAX = 4


class StepEpilogue(C.Structure):
    """htfs_step_epilogue: the stand-in integrator (and a brick's halo pack) as the one-kernel step's epilogue."""
    _fields_ = [("d_vel", C.c_void_p), ("d_pos_next", C.c_void_p), ("dtype", C.c_int), ("dt", C.c_double), ("box", Box),
                ("brick", C.c_void_p), ("d_row_slots", C.c_void_p), ("d_halo_send", C.c_void_p), ("d_ghost_direct", C.c_void_p)]


MBOX_MAX_MSG, MBOX_MAX_RANKS, IPC_HANDLE_BYTES = 8, 64, 64   # include/htf_standin.h HTFS_MBOX_MAX_MSG / _MAX_RANKS, HTFS_IPC_HANDLE_BYTES


class Mailbox(C.Structure):
    """htfs_mailbox: one rank's view of a message channel between ranks (csrc/mailbox.hip)."""
    _fields_ = [("remote", C.c_void_p * MBOX_MAX_MSG), ("remote_signal", C.c_void_p * MBOX_MAX_MSG), ("mine", C.c_void_p),
                ("my_signal", C.c_void_p), ("state", C.c_void_p), ("spin_limit", C.c_uint), ("half_units", C.c_uint)]


class ReduceBox(C.Structure):
    """htfs_reduce_box: the all-to-all table of htfs_mailbox_allreduce_max_f32."""
    _fields_ = [("remote", C.c_void_p * MBOX_MAX_RANKS), ("mine", C.c_void_p), ("state", C.c_void_p), ("spin_limit", C.c_uint),
                ("world", C.c_int), ("rank", C.c_int)]


class Mirror(C.Structure):
    """htfs_mirror (word ranges the check kernel's last workgroup copies from device to pinned host memory)."""
    _fields_ = [("src", C.c_void_p * MIRROR_MAX), ("dst", C.c_void_p * MIRROR_MAX), ("words", C.c_uint * MIRROR_MAX), ("n", C.c_uint)]


class BrickWork(C.Structure):
    """htfs_brick_work."""
    _fields_ = [("key", C.c_void_p), ("order", C.c_void_p), ("sort_scratch", C.c_void_p), ("start1", C.c_void_p),
                ("start2", C.c_void_p), ("tmp_pos", C.c_void_p), ("tmp_vel", C.c_void_p)]


class PotentialDesc(C.Structure):
    _fields_ = [("kind", C.c_int), ("sigma", C.c_double),
                ("gauss_r0", C.c_double), ("gauss_gap", C.c_double), ("gauss_coef", C.c_double),
                ("lj_w0", C.c_double), ("lj_w1", C.c_double), ("d_theta", C.c_void_p), ("n_terms", C.c_int),
                ("coef", C.c_double * MAX_POLY_TERMS), ("power", C.c_int * MAX_POLY_TERMS),
                ("K", C.c_int), ("H1", C.c_int), ("H2", C.c_int), ("activation", C.c_int),
                ("mlp_precision", C.c_int), ("rbf_low", C.c_double), ("rbf_high", C.c_double),
                ("W1", C.c_void_p), ("b1", C.c_void_p), ("W2", C.c_void_p), ("b2", C.c_void_p),
                ("W3", C.c_void_p), ("b3", C.c_void_p), ("poly_cut", C.c_double),
                ("jit_image", C.c_void_p), ("jit_image_bytes", C.c_size_t), ("jit_flags", C.c_int)]


class OptimizerDesc(C.Structure):
    _fields_ = [("kind", C.c_int), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("epsilon", C.c_float), ("nonneg_mask", C.c_uint), ("l1_reg", C.c_float * 8)]


class Config(C.Structure):
    _fields_ = [("r_cut", C.c_double), ("nneighs", C.c_uint), ("force_mode", C.c_int),
                ("period", C.c_uint), ("batch_size", C.c_uint), ("scalar_dtype", C.c_int),
                ("check_nlist", C.c_int), ("virial", C.c_int), ("max_n", C.c_uint), ("fused", C.c_int)]


class HoomdArrays(C.Structure):
    _fields_ = [("pos", C.c_void_p), ("N", C.c_uint), ("n_ghost", C.c_uint),
                ("n_neigh", C.c_void_p), ("nlist", C.c_void_p), ("head_list", C.c_void_p),
                ("box", Box), ("force", C.c_void_p), ("virial", C.c_void_p),
                ("virial_pitch", C.c_size_t)]


_vp, _u, _i, _d, _sz = C.c_void_p, C.c_uint, C.c_int, C.c_double, C.c_size_t

# name -> (restype, argtypes); must list every HTF_API symbol of include/htf_amd.h
PROTOTYPES = {
    "htf_last_error": (C.c_char_p, []),
    "htf_abi_version": (_i, []),
    "htf_device_count": (_i, []),
    "htf_potential_create": (_i, [C.POINTER(PotentialDesc), C.POINTER(_vp)]),
    "htf_potential_destroy": (None, [_vp]),
    "htf_build_pair_vectors": (_i, [_vp, _i, _vp, _i, _u, _u, _u, _u, _u, C.POINTER(Box), _vp, _vp, _vp, _d, _vp, _vp]),
    "htf_eval_forces": (_i, [_vp, _vp, _i, _u, _u, _vp, _i, _vp, _vp]),
    "htf_eval_forces_typed": (_i, [_vp, _vp, _i, _u, _u, _vp, _i, _vp, _i, _vp, _vp]),
    "htf_jit_available": (_i, []),
    "htf_jit_compile": (_i, [C.c_char_p, C.c_char_p, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, C.c_size_t]),
    "htf_jit_free": (None, [_vp]),
    "htf_fused_forces": (_i, [_vp, _vp, _i, _u, _u, _u, _u, C.POINTER(Box), _vp, _vp, _vp, _d, _vp, _i, _vp, _vp, _vp]),
    "htf_build_eval_forces": (_i, [_vp, _vp, _vp, _i, _u, _u, _u, _u, C.POINTER(Box), _vp, _vp, _vp, _d, _vp, _i, _vp, _vp, _vp]),
    "htf_eval_forces2": (_i, [_vp, _vp, _vp, _i, _u, _u, _vp, _vp, _i, _vp, C.c_float, C.c_float, _u, _vp, _vp]),
    "htf_build_eval_forces2": (_i, [_vp, _vp, _vp, _vp, _i, _u, _u, _u, _u, C.POINTER(Box), _vp, _vp, _vp, _d, _vp, _vp, _i, _vp,
                                    C.c_float, C.c_float, _u, _vp, _vp]),
    "htf_build_eval2_num_partials": (_u, [_u]),
    "htf_eval2_num_partials": (_u, [_u, _u]),
    "htf_reduce_partials": (_i, [_vp, _u, C.c_float, _vp, _vp]),
    "htf_bias_combine": (_i, [_vp, _vp, _vp, _vp, _i, _u, _vp]),
    "htf_potential_num_params": (_i, [_vp]),
    "htf_train_scratch_floats": (_sz, [_vp, _u, _u]),
    "htf_train_pair_grad": (_i, [_vp, _vp, _i, _u, _u, _vp, _i, _vp, _vp, _vp, _vp]),
    "htf_train_pair_grad_list": (_i, [_vp, _vp, _i, _u, _u, C.POINTER(Box), _vp, _vp, _vp, _d, _vp, _i, _vp, _vp, _vp, _vp]),
    "htf_optimizer_step": (_i, [_vp, _u, _vp, C.c_float, _vp, C.POINTER(OptimizerDesc), _vp]),
    "htf_optimizer_step_n": (_i, [_vp, _u, _vp, C.c_float, _vp, C.POINTER(OptimizerDesc), _vp]),
    "htf_potential_refresh": (_i, [_vp, _vp]),
    "htf_add_virial": (_i, [_vp, _vp, _i, _u, _sz, _vp]),
    "htf_add_scalar4": (_i, [_vp, _vp, _i, _u, _vp]),
    "htf_energy_sum": (_i, [_vp, _i, _u, _vp, _vp]),
    "htf_copy_positions": (_i, [_vp, _i, _vp, _i, _u, _u, _i, _vp]),
    "htf_copy3": (_i, [_vp, _i, _vp, _i, _u, _vp]),
    "htf_positions_forces_radial": (_i, [_vp, _i, _u, _i, _i, _d, _vp, _i, _vp]),
    "htf_check_nlist": (_i, [_vp, _i, _u, _u, _vp, _vp]),
    "htf_nlist_rinv": (_i, [_vp, _i, _u, _u, _vp, _vp]),
    "htf_create": (_i, [C.POINTER(Config), C.POINTER(_vp)]),
    "htf_destroy": (None, [_vp]),
    "htf_set_potential": (_i, [_vp, _vp]),
    "htf_resize": (_i, [_vp, _u]),
    "htf_compute_forces": (_i, [_vp, _u, C.POINTER(HoomdArrays), _vp]),
    "htf_compute_forces_rows": (_i, [_vp, _u, C.POINTER(HoomdArrays), _u, _u, _vp]),
    "htf_get_nlist_buffer": (_vp, [_vp]),
    "htf_reset_nlist_buffer": (_i, [_vp, _vp]),
    "htf_get_positions_buffer": (_vp, [_vp]),
    "htf_get_virial_buffer": (_vp, [_vp]),
    "htf_get_batch_capacity": (_u, [_vp]),
    "htf_top_k": (_i, [_vp, _u, _u, _u, _vp, _vp, _vp]),
    "htf_rdf_histogram": (_i, [_vp, _i, _u, _u, C.c_float, C.c_float, _u, _vp, _u, _i, _i, _vp, _vp]),
    "htf_rdf_finalize": (_i, [_vp, _u, C.c_float, C.c_float, _vp, _vp, _vp]),
    "htf_rbf_expansion": (_i, [_vp, _sz, _d, _d, _u, _vp, _vp]),
    "htf_eds_update": (_i, [_vp, _vp, C.c_float, _i, C.c_float, C.c_float, _vp]),
    "htf_wrap_vector": (_i, [_vp, _i, _sz, C.POINTER(Box), _vp, _vp]),
    "htf_halo_available": (_i, []),
    "htf_halo_unique_id": (_i, [_vp]),
    "htf_halo_create": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "htf_halo_destroy": (None, [_vp]),
    "htf_halo_exchange_begin": (_i, [_vp, _vp, _i, _i, _i, _u, _u, _u, _u, _u, _u, _u, _u, _vp]),
    "htf_halo_exchange_end": (_i, [_vp, _vp]),
    "htf_halo_exchange_n": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i]),
    "htf_halo_allreduce_max_f32": (_i, [_vp, _vp, _u, _vp]),
    "htf_halo_comm_info": (_i, [_vp, _vp, _vp, _vp]),
    "htf_profile_enable": (_i, [_vp, _i]),
    "htf_profile_read": (_i, [_vp, C.POINTER(_d), C.POINTER(_d), C.POINTER(_u)]),
}

# HOOMD stand-in entry points (include/htf_standin.h) -- outside the drop-in boundary
STANDIN_PROTOTYPES = {
    "htfs_gather4": (_i, [_vp, _vp, _vp, _i, _u, _vp]),
    "htfs_gather4_tagged": (_i, [_vp, _vp, _vp, _i, _u, _i, _vp]),
    "htfs_cell_sort": (_i, [_vp, _u, _u, _vp, _vp, _vp, _vp]),
    "htfs_gather4_tagged_live": (_i, [_vp, _vp, _vp, _i, _u, _vp, _i, _vp]),
    "htfs_brick_migrate_pack": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "htfs_brick_migrate_merge": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "htfs_brick_pack_halo": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "htfs_brick_nve_halo": (_i, [_vp, _vp, _vp, _vp, _i, _d, C.POINTER(Box), _vp, _vp, _vp, _vp]),
    "htfs_brick_nve_halo_peer": (_i, [_vp, _vp, _vp, _vp, _i, _d, C.POINTER(Box), _vp, _vp, _vp]),
    "htfs_brick_pack_halo_peer": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "htfs_brick_unpack_halo": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "htfs_set_step_epilogue": (_i, [_vp, _i, _vp, _vp]),
    "htfs_use_step_epilogue": (_i, [_vp, _i]),
    "htfs_brick_row_slots": (_i, [_vp, _vp, _vp, _vp]),
    "htfs_shared_alloc": (_i, [_sz, _i, _vp]),
    "htfs_shared_free": (_i, [_vp]),
    "htfs_ipc_export": (_i, [_vp, _vp]),
    "htfs_ipc_import": (_i, [_vp, _vp]),
    "htfs_ipc_close": (_i, [_vp]),
    "htfs_mailbox_push": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "htfs_mailbox_pull": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _u, _vp]),
    "htfs_mailbox_allreduce_max_f32": (_i, [_vp, _vp, _vp, _u, _vp]),
    "htfs_slab_classify": (_i, [_vp, _i, _u, _vp, _i, _i, _d, _vp, _vp]),
    "htfs_segment_copy": (_i, [_vp, _vp, _u, _u, _vp, _vp, _vp, _vp]),
    "htfs_key_sort16": (_i, [_vp, _u, _vp, _vp, _vp, _vp]),
    "htfs_nve_step": (_i, [_vp, _vp, _vp, _i, _u, _d, C.POINTER(Box), _vp]),
    "htfs_max_displacement2": (_i, [_vp, _vp, _i, _u, C.POINTER(Box), _vp, _vp]),
    "htfs_check_displacement2": (_i, [_vp, _vp, _i, _u, C.POINTER(Box), _vp, _vp, _vp, C.POINTER(Mirror), _vp]),
    "htfs_build_nlist": (_i, [_vp, _vp, _i, _u, _u, C.POINTER(Box), _d, C.POINTER(_i * 3), C.POINTER(_i * 3), _vp, _u, _i, _vp, _vp, _vp,
                              _vp, _vp, _vp]),
    "htfs_cell_index": (_i, [_vp, _i, _u, C.POINTER(Box), C.POINTER(_i * 3), _vp, _vp]),
    "htfs_set_gate": (_i, [_vp, _d]),
    "htfs_commit_rebuild": (_i, [_vp, _vp, _i, _u, _vp, _vp]),
    "htfs_rebuild_nlist": (_i, [_vp, _i, _u, C.POINTER(Box), _d, C.POINTER(_i * 3), C.POINTER(_i * 3), _vp, _vp, _vp, _vp, _vp, _u, _i,
                                _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "htfs_rebuild_nlist_ghosts": (_i, [_vp, _i, _u, _u, C.POINTER(Box), _d, C.POINTER(_i * 3), C.POINTER(_i * 3), _vp, _vp, _vp, _vp, _vp, _u, _i,
                                       _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "htfs_check_rebuild_nlist": (_i, [_vp, _i, _u, C.POINTER(Box), _d, C.POINTER(_i * 3), C.POINTER(_i * 3), _vp, _vp, _vp, _vp, _vp, _u, _i,
                                      _vp, _vp, _vp, _vp, _vp, _vp, _d, _vp, _vp, _vp]),
}


ABI_VERSION = 4  # include/htf_amd.h HTF_AMD_ABI_VERSION: the struct layouts the ctypes Structures of this file mirror


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "hoomd_tf_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C hoomd_tf_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in list(PROTOTYPES.items()) + list(STANDIN_PROTOTYPES.items()):
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.htf_abi_version() != ABI_VERSION:
        raise ImportError("hoomd_tf_amd: ABI version mismatch (library %s: %d, this binding: %d); rebuild with "
                          "`make -C hoomd_tf_amd/csrc`" % (LIB_PATH, lib.htf_abi_version(), ABI_VERSION))
    return lib


# ---------------------------------------------------------------------------- the pybind11 binding of the same ABI
# Every call goes through hoomd_tf_amd/_htf_abi.so (csrc/pybind_abi.cc: one template per C signature, pointers as integers)
# when that module has been built -- BASELINE's "thin pybind11 C-ABI" -- and through ctypes otherwise (HTF_BINDING=ctypes |
# pybind11 forces either; HTF_AMD_LIB, an A/B copy of the library, implies ctypes: the module is linked against the default
# one).  The call sites do not change: the facade below turns what they pass -- ints, None, c_void_p, byref(structure),
# ctypes arrays -- into addresses.  The GPU suite passes under either binding (tools/evidence_pass.sh runs both).
def _address(a):
    if a is None:
        return 0
    if type(a) is int:
        return a
    if isinstance(a, C.c_void_p):
        return a.value or 0
    if isinstance(a, (C.Structure, C.Array)):
        return C.addressof(a)
    obj = getattr(a, "_obj", None)  # byref(x)
    if obj is not None:
        return C.addressof(obj)
    if isinstance(a, C._Pointer):
        return C.cast(a, C.c_void_p).value or 0
    if isinstance(a, C._SimpleCData):
        return C.addressof(a)
    return int(a)


def _is_pointer(t):
    return t is _vp or t is C.c_char_p or (isinstance(t, type) and issubclass(t, C._Pointer))


class _PybindLib:
    """Same attribute surface as the ctypes library object, backed by the pybind11 module."""

    def __init__(self, mod):
        self._mod = mod
        for name, (res, args) in list(PROTOTYPES.items()) + list(STANDIN_PROTOTYPES.items()):
            fn = getattr(mod, name)  # AttributeError if the module lacks a declared symbol
            ptr_at = tuple(i for i, t in enumerate(args) if _is_pointer(t))
            setattr(self, name, self._wrap(fn, ptr_at, res is _vp))

    @staticmethod
    def _wrap(fn, ptr_at, returns_pointer):
        def call(*a):
            if ptr_at:
                a = list(a)
                for i in ptr_at:
                    a[i] = _address(a[i])
            r = fn(*a)
            return (r or None) if returns_pointer else r
        return call


def _load_pybind():
    import importlib.util
    path = os.path.join(_HERE, "_htf_abi.so")
    if not os.path.exists(path):
        raise ImportError("hoomd_tf_amd: HTF_BINDING=pybind11 but %s is missing: `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C hoomd_tf_amd/csrc pybind`" % path)
    spec = importlib.util.spec_from_file_location("_htf_abi", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_ctypes_lib = _load()  # (also what resolves libhtf_amd.so for the pybind11 module: same library, loaded once)
# default: the pybind11 module when it has been built (build() / make pybind), else ctypes -- both thin, both on the one
# C ABI; the library itself is never optional (the _load() above has already failed loudly without it)
BINDING = os.environ.get("HTF_BINDING") or ("pybind11" if os.path.exists(os.path.join(_HERE, "_htf_abi.so")) and not os.environ.get("HTF_AMD_LIB")
                                              else "ctypes")
if BINDING == "pybind11":
    lib = _PybindLib(_load_pybind())
elif BINDING == "ctypes":
    lib = _ctypes_lib
else:
    raise ImportError("hoomd_tf_amd: HTF_BINDING must be 'ctypes' or 'pybind11', not %r" % BINDING)


def last_error():
    return lib.htf_last_error().decode("utf-8", "replace")


def check(rc):
    """Map C status codes to the exception types the reference raises."""
    if rc == HTF_OK:
        return
    msg = last_error()
    if rc == HTF_ERR_INVALID:
        raise ValueError(msg)
    if rc == HTF_ERR_NLIST_OVERFLOW:
        raise NlistOverflowError(msg)
    if rc == HTF_ERR_SKEWED_BOX:
        raise SkewedBoxError(msg)
    if rc == HTF_ERR_NOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg)


def make_box(box3x3, periodic=(1, 1, 1)):
    """3x3 [[lo],[hi],[xy,xz,yz]] (TensorflowCompute.cc:271-282) -> htf_box."""
    b = Box()
    for d in range(3):
        b.lo[d] = float(box3x3[0][d])
        b.hi[d] = float(box3x3[1][d])
        b.tilt[d] = float(box3x3[2][d])
        b.periodic[d] = int(periodic[d])
    return b
