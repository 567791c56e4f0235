"""Torch-tensor front of the C ABI: every function takes HOOMD-layout device tensors,
passes their raw pointers (zero-copy) and the current HIP stream to libhtf_amd.so and
returns torch tensors.  torch is plumbing here (device memory + streams); all
arithmetic happens in the HIP kernels.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import lib, check


def _dt(t):
    if t.dtype == torch.float32:
        return _lib.HTF_F32
    if t.dtype == torch.float64:
        return _lib.HTF_F64
    raise ValueError("expected a float32/float64 tensor, got %s" % t.dtype)


# torch.cuda.current_stream() builds a Stream object (and, without an explicit device index, asks the runtime for the device
# count: ~8 us) every call; the step loop asks four times a step.  The raw handle is one C call.
_raw_current_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def raw_stream(device_index=None):
    """hipStream_t of torch's current stream on ``device_index`` (default: the current device), as an integer."""
    if device_index is None:
        device_index = torch.cuda.current_device()
    if _raw_current_stream is not None:
        return _raw_current_stream(device_index)
    return torch.cuda.current_stream(device_index).cuda_stream


def _stream(t):
    return C.c_void_p(raw_stream(t.device.index))


def _dev(t, name, dtype=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError("%s must be a CUDA/HIP device tensor (the evaluator has no CPU path)" % name)
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    if dtype is not None and t.dtype != dtype:
        raise ValueError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t


def _u32(t, name):
    _dev(t, name)
    if t.dtype not in (torch.int32, torch.uint32):
        raise ValueError("%s must be int32/uint32 (HOOMD unsigned int), got %s" % (name, t.dtype))
    return t


class Potential:
    """Owns an ``htf_potential`` handle (device copies of any weights)."""

    def __init__(self, kind, sigma=0.0, coefs=(), powers=(), mlp=None, rbf=(0.0, 0.0),
                 activation="linear", mlp_precision="fp32", gauss=(0.0, 1.0, 0.0), lj_param=(1.0, 1.0),
                 theta=None, poly_cut=0.0):
        d = _lib.PotentialDesc()
        d.kind = kind
        d.sigma = float(sigma)
        d.gauss_r0, d.gauss_gap, d.gauss_coef = (float(x) for x in gauss)
        d.lj_w0, d.lj_w1 = float(lj_param[0]), float(lj_param[1])
        self.theta = theta  # device parameter vector of a trainable potential (kept alive here)
        if theta is not None:
            _dev(theta, "theta", torch.float32)
            d.d_theta = theta.data_ptr()
        d.n_terms = len(coefs)
        d.poly_cut = float(poly_cut)
        if len(coefs) != len(powers):
            raise ValueError("coefs and powers differ in length")
        if len(coefs) > _lib.MAX_POLY_TERMS:
            raise ValueError("at most %d polynomial terms" % _lib.MAX_POLY_TERMS)
        for k, (c, p) in enumerate(zip(coefs, powers)):
            d.coef[k] = float(c)
            d.power[k] = int(p)
        self._keep = []
        if mlp is not None:
            ws = [np.ascontiguousarray(mlp[k], dtype=np.float32) for k in ("W1", "b1", "W2", "b2", "W3", "b3")]
            self._keep = ws
            d.K, d.H1 = ws[0].shape
            d.H2 = ws[2].shape[1]
            if ws[2].shape[0] != d.H1 or ws[4].shape != (d.H2, 1):
                raise ValueError("inconsistent MLP weight shapes")
            d.W1, d.b1, d.W2, d.b2, d.W3, d.b3 = (w.ctypes.data for w in ws)
            d.rbf_low, d.rbf_high = float(rbf[0]), float(rbf[1])
            d.activation = {"linear": _lib.ACT_LINEAR, None: _lib.ACT_LINEAR, "tanh": _lib.ACT_TANH}[activation]
            if mlp_precision not in ("fp32", "bf16", "split", "split16"):
                raise ValueError("pair-MLP precision must be 'fp32', 'bf16', 'split' or 'split16', not %r" % (mlp_precision,))
            d.mlp_precision = {"fp32": _lib.MLP_FP32, "bf16": _lib.MLP_BF16, "split": _lib.MLP_SPLIT,
                               "split16": _lib.MLP_SPLIT16}[mlp_precision]
        self.kind = kind
        self._h = C.c_void_p()
        check(lib.htf_potential_create(C.byref(d), C.byref(self._h)))

    @property
    def handle(self):
        return self._h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and lib is not None:  # lib is None once the interpreter is tearing modules down
            lib.htf_potential_destroy(h)
            self._h = None

    # the reference's declarative models
    @classmethod
    def lj(cls):
        return cls(_lib.POT_LJ)

    @classmethod
    def wca(cls, sigma, theta=None):
        return cls(_lib.POT_WCA, sigma=sigma, theta=theta)

    @classmethod
    def simple(cls):
        return cls(_lib.POT_SIMPLE)

    @classmethod
    def rinv_poly(cls, coefs, powers, theta=None, cut=0.0):
        """sum_k c_k rinv^p_k; ``cut`` > 0: times ``cast(norm(nlist[:, :, :3]) < cut)`` (example 01's truncated r^-12)."""
        return cls(_lib.POT_RINV_POLY, coefs=coefs, powers=powers, theta=theta, poly_cut=cut)

    @classmethod
    def lj_param(cls, w0, w1, theta=None):
        """Trainable LJ of example 06: e = w0 * 4 (q^2 - q) / 2, q = w1^6 / safe_norm(x)^6."""
        return cls(_lib.POT_LJ_PARAM, lj_param=(w0, w1), theta=theta)

    @classmethod
    def jit(cls, body, reads_own_type=False, theta=None):
        """A traced elementwise pair energy as generated kernels (hoomd_tf_amd/codegen.py): ``body`` is the text
        codegen.unit_text emitted; compiled once per expression (hipRTC / ``hipcc --genco``, cached), loaded as HTF_POT_JIT.
        ``reads_own_type``: the body reads ``ti`` (positions[i, 3]) -- the streaming evaluator then needs the positions.
        ``theta``: the device float32 vector of the energy's weights (``p.theta[k]`` in the body): kernels read it at launch, the
        training sweep (htf_train_pair_grad) differentiates with respect to it."""
        from . import codegen
        image, key = codegen.compile_body(body)
        self = cls.__new__(cls)
        d = _lib.PotentialDesc()
        if theta is not None:
            _dev(theta, "theta", torch.float32)
            d.d_theta = theta.data_ptr()
        d.kind = _lib.POT_JIT
        self._image = C.create_string_buffer(image, len(image))   # (kept alive with the potential)
        d.jit_image = C.cast(self._image, C.c_void_p)
        d.jit_image_bytes = len(image)
        d.jit_flags = _lib.JIT_READS_OWN_TYPE if reads_own_type else 0
        self.reads_own_type = bool(reads_own_type)
        self.theta, self._keep, self.kind, self.jit_key, self.body = theta, [], _lib.POT_JIT, key, body
        self._h = C.c_void_p()
        check(lib.htf_potential_create(C.byref(d), C.byref(self._h)))
        return self

    @property
    def num_params(self):
        return int(lib.htf_potential_num_params(self._h))

    @classmethod
    def gauss(cls, r0, gap, coef=1.0):
        """c * exp(-(r - r0)^2 / gap): one RBFExpansion channel as a pair energy (C4's soft RDF bin)."""
        return cls(_lib.POT_GAUSS, gauss=(r0, gap, coef))

    @classmethod
    def pair_mlp(cls, params, low, high, activation="tanh", precision="fp32", theta=None):
        """``theta``: flat device weights (Keras get_weights() order) the potential keeps reading;
        call ``refresh()`` after changing them."""
        return cls(_lib.POT_PAIR_MLP, mlp=params, rbf=(low, high), activation=activation,
                   mlp_precision=precision, theta=theta)

    @classmethod
    def topk_mlp(cls, params, activation=None):
        """Example 08's per-particle network on the ``K = W1.shape[0]`` largest 1/r of each row (descending):
        Dense(H1) -> Dense(H2) -> Dense(1); ``params``: W1 [K, H1], b1, W2 [H1, H2], b2, W3 [H2, 1], b3."""
        return cls(_lib.POT_TOPK_MLP, mlp=params, activation=activation or "linear")

    def refresh(self):
        """Rebuild derived device data (pair-MLP operand images) from the parameter vector."""
        check(lib.htf_potential_refresh(self._h, _stream(self.theta) if self.theta is not None else None))
        self.version = getattr(self, "version", 0) + 1


def build_pair_vectors(pos, n_neigh, head_list, nlist, box, r_cut, NN, offset=0, batch_size=None,
                       n_local=None, out=None, out_dtype=torch.float32, max_count=None, periodic=(1, 1, 1)):
    """prepareNeighbors (TensorflowCompute.cc:303-374): -> [B, NN, 4]."""
    _dev(pos, "pos")
    N = int(n_neigh.shape[0]) if n_local is None else int(n_local)
    B = N - offset if batch_size is None else int(batch_size)
    if out is None:
        out = torch.empty((B, NN, 4), dtype=out_dtype, device=pos.device)
    _dev(out, "out")
    if B == 0:
        return out
    b = box if isinstance(box, _lib.Box) else _lib.make_box(box, periodic)
    check(lib.htf_build_pair_vectors(
        out.data_ptr(), _dt(out), pos.data_ptr(), _dt(pos), N, NN, offset, B, pos.shape[0] - N, C.byref(b),
        _u32(n_neigh, "n_neigh").data_ptr(), _u32(nlist, "nlist").data_ptr(),
        _u32(head_list, "head_list").data_ptr(), float(r_cut),
        max_count.data_ptr() if max_count is not None else None, _stream(pos)))
    return out


def eval_forces(potential, nlist, virial=False, out=None, out_dtype=None, virial_out=None, positions=None):
    """SimModel.compute for a declarative potential: nlist [B,NN,4] -> forces [B,4]
    (fx, fy, fz, energy) and, if ``virial``, the [B,3,3] virial.  ``positions`` ([B,4], the tensor compute() receives beside
    nlist): needed by a traced energy that reads the row particle's own type (htf_eval_forces_typed), ignored otherwise."""
    _dev(nlist, "nlist")
    if nlist.dim() != 3 or nlist.shape[2] != 4:
        raise ValueError("nlist must be [B, NN, 4]")
    B, NN = int(nlist.shape[0]), int(nlist.shape[1])
    od = out_dtype or (out.dtype if out is not None else nlist.dtype)
    if out is None:
        out = torch.empty((B, 4), dtype=od, device=nlist.device)
    _dev(out, "out")
    v = None
    if virial:
        v = virial_out if virial_out is not None else torch.empty((B, 3, 3), dtype=out.dtype, device=nlist.device)
        _dev(v, "virial_out", out.dtype)
    if positions is not None:
        _dev(positions, "positions")
        if positions.dim() != 2 or positions.shape[0] < B or positions.shape[1] != 4 or not positions.is_contiguous():
            raise ValueError("positions must be a contiguous [B, 4] tensor")
        check(lib.htf_eval_forces_typed(potential.handle, nlist.data_ptr(), _dt(nlist), B, NN, positions.data_ptr(), _dt(positions),
                                        out.data_ptr(), _dt(out), v.data_ptr() if v is not None else None, _stream(nlist)))
    else:
        check(lib.htf_eval_forces(potential.handle, nlist.data_ptr(), _dt(nlist), B, NN, out.data_ptr(), _dt(out),
                                  v.data_ptr() if v is not None else None, _stream(nlist)))
    return (out, v) if virial else out


def fused_forces(potential, pos, n_neigh, head_list, nlist, box, r_cut, NN, offset=0, batch_size=None,
                 n_local=None, virial=False, out_dtype=None, check_count=None, periodic=(1, 1, 1), pair_vectors=None):
    """build_pair_vectors + eval_forces in one kernel.  ``pair_vectors`` (fp32 [B, NN, 4]) also
    receives the tensor, bit-identical to build_pair_vectors' (htf_build_eval_forces); without it
    the tensor is never materialised (htf_fused_forces)."""
    _dev(pos, "pos")
    N = int(n_neigh.shape[0]) if n_local is None else int(n_local)
    B = N - offset if batch_size is None else int(batch_size)
    od = out_dtype or pos.dtype
    out = torch.empty((B, 4), dtype=od, device=pos.device)
    v = torch.empty((B, 3, 3), dtype=od, device=pos.device) if virial else None
    if B == 0:
        return (out, v) if virial else out
    b = box if isinstance(box, _lib.Box) else _lib.make_box(box, periodic)
    tail = (pos.data_ptr(), _dt(pos), N, NN, offset, B, C.byref(b),
            _u32(n_neigh, "n_neigh").data_ptr(), _u32(nlist, "nlist").data_ptr(),
            _u32(head_list, "head_list").data_ptr(), float(r_cut), out.data_ptr(), _dt(out),
            v.data_ptr() if v is not None else None,
            check_count.data_ptr() if check_count is not None else None, _stream(pos))
    if pair_vectors is None:
        check(lib.htf_fused_forces(potential.handle, *tail))
    else:
        _dev(pair_vectors, "pair_vectors", torch.float32)
        if tuple(pair_vectors.shape) != (B, NN, 4):
            raise ValueError("pair_vectors must be [%d, %d, 4]" % (B, NN))
        check(lib.htf_build_eval_forces(potential.handle, pair_vectors.data_ptr(), *tail))
    return (out, v) if virial else out


def eval_forces2(pot_a, pot_b, nlist, out_a=None, out_b=None, partials=None, out_dtype=None, rdf=None):
    """Two potentials in one pass: -> (forces_a [B,4], forces_b [B,4]); ``partials`` receives
    the per-block sums of forces_b[:, 3] (see reduce_partials).  ``rdf = (r0, r1, hist)`` with a
    zeroed int32 ``hist`` of nbins + 2 entries fuses the compute_rdf histogram into the sweep."""
    _dev(nlist, "nlist")
    B, NN = int(nlist.shape[0]), int(nlist.shape[1])
    od = out_dtype or (out_a.dtype if out_a is not None else nlist.dtype)
    if out_a is None:
        out_a = torch.empty((B, 4), dtype=od, device=nlist.device)
    if out_b is None:
        out_b = torch.empty((B, 4), dtype=od, device=nlist.device)
    _dev(out_a, "out_a")
    _dev(out_b, "out_b", out_a.dtype)
    r0, r1, nbt, hist = (0.0, 1.0, 0, None) if rdf is None else (rdf[0], rdf[1], int(rdf[2].numel()), rdf[2])
    check(lib.htf_eval_forces2(pot_a.handle, pot_b.handle, nlist.data_ptr(), _dt(nlist), B, NN, out_a.data_ptr(),
                               out_b.data_ptr(), _dt(out_a), partials.data_ptr() if partials is not None else None,
                               float(r0), float(r1), nbt, hist.data_ptr() if hist is not None else None,
                               _stream(nlist)))
    return out_a, out_b


def build_eval_forces2(pot_a, pot_b, pos, n_neigh, head_list, nlist, box, r_cut, NN, offset=0, batch_size=None,
                       n_local=None, out_dtype=None, partials=None, rdf=None, pair_vectors=None, periodic=(1, 1, 1),
                       out_a=None, out_b=None):
    """build_pair_vectors + eval_forces2 in ONE kernel (config C4's sweep): -> (forces_a, forces_b).
    ``partials``: num_partials_fused(B) floats; ``rdf = (r0, r1, hist)``; ``pair_vectors`` (fp32
    [B, NN, 4], optional) also receives the tensor."""
    _dev(pos, "pos")
    N = int(n_neigh.shape[0]) if n_local is None else int(n_local)
    B = N - offset if batch_size is None else int(batch_size)
    od = out_dtype or (out_a.dtype if out_a is not None else pos.dtype)
    if out_a is None:
        out_a = torch.empty((B, 4), dtype=od, device=pos.device)
    if out_b is None:
        out_b = torch.empty((B, 4), dtype=od, device=pos.device)
    _dev(out_a, "out_a")
    _dev(out_b, "out_b", out_a.dtype)
    if out_a.shape[0] < B or out_b.shape[0] < B:
        raise ValueError("out_a / out_b must hold %d rows" % B)
    if B == 0:
        return out_a, out_b
    b = box if isinstance(box, _lib.Box) else _lib.make_box(box, periodic)
    r0, r1, nbt, hist = (0.0, 1.0, 0, None) if rdf is None else (rdf[0], rdf[1], int(rdf[2].numel()), rdf[2])
    if pair_vectors is not None:
        _dev(pair_vectors, "pair_vectors", torch.float32)
        if tuple(pair_vectors.shape) != (B, NN, 4):
            raise ValueError("pair_vectors must be [%d, %d, 4]" % (B, NN))
    check(lib.htf_build_eval_forces2(pot_a.handle, pot_b.handle, pair_vectors.data_ptr() if pair_vectors is not None else None,
                                     pos.data_ptr(), _dt(pos), N, NN, offset, B, C.byref(b),
                                     _u32(n_neigh, "n_neigh").data_ptr(), _u32(nlist, "nlist").data_ptr(),
                                     _u32(head_list, "head_list").data_ptr(), float(r_cut), out_a.data_ptr(),
                                     out_b.data_ptr(), _dt(out_a), partials.data_ptr() if partials is not None else None,
                                     float(r0), float(r1), nbt, hist.data_ptr() if hist is not None else None,
                                     _stream(pos)))
    return out_a, out_b


def num_partials_fused(B):
    return int(lib.htf_build_eval2_num_partials(int(B)))


def num_partials(B, NN):
    return int(lib.htf_eval2_num_partials(int(B), int(NN)))


def reduce_partials(partials, n, scale, out):
    check(lib.htf_reduce_partials(partials.data_ptr(), int(n), float(scale), out.data_ptr(), _stream(partials)))
    return out


def bias_combine(force, bias, alpha, cv):
    """force += alpha * (bias.xyz, cv) with alpha, cv device scalars (fp32)."""
    _dev(force, "force")
    _dev(bias, "bias", force.dtype)
    check(lib.htf_bias_combine(force.data_ptr(), bias.data_ptr(), alpha.data_ptr(), cv.data_ptr(), _dt(force),
                               int(force.shape[0]), _stream(force)))
    return force


def train_pair_grad(potential, nlist, labels, pred=None, accum=None):
    """One training sweep (train_on_batch's forward + gradient for MSE over [B, 4]): returns
    accum [1 + P] = {sum of squared residuals, d/dtheta_k} on the device (and fills ``pred``)."""
    _dev(nlist, "nlist")
    _dev(labels, "labels")
    B, NN = int(nlist.shape[0]), int(nlist.shape[1])
    P = potential.num_params
    if P == 0:
        raise ValueError("this potential has no trainable parameters")
    if accum is None:
        accum = torch.empty(1 + P, dtype=torch.float32, device=nlist.device)
    scratch = torch.empty(int(lib.htf_train_scratch_floats(potential.handle, B, NN)), dtype=torch.float32,
                          device=nlist.device)
    if pred is not None:
        _dev(pred, "pred", torch.float32)
    check(lib.htf_train_pair_grad(potential.handle, nlist.data_ptr(), _dt(nlist), B, NN, labels.data_ptr(), _dt(labels),
                                  pred.data_ptr() if pred is not None else None, accum.data_ptr(), scratch.data_ptr(),
                                  _stream(nlist)))
    return accum


def train_pair_grad_list(potential, pos, n_neigh, head_list, nlist, box, r_cut, NN, labels, n_local=None, pred=None, accum=None,
                         periodic=(1, 1, 1)):
    """train_pair_grad from HOOMD's index neighbor list (htf_train_pair_grad_list): no [B, NN, 4] tensor is written or read.
    Rows 0 .. n_local-1 are trained on.  Closed forms and traced energies with weights."""
    _dev(pos, "pos")
    _dev(labels, "labels")
    B = int(n_neigh.shape[0]) if n_local is None else int(n_local)
    P = potential.num_params
    if P == 0:
        raise ValueError("this potential has no trainable parameters")
    if accum is None:
        accum = torch.empty(1 + P, dtype=torch.float32, device=pos.device)
    scratch = torch.empty(int(lib.htf_train_scratch_floats(potential.handle, B, NN)), dtype=torch.float32, device=pos.device)
    if pred is not None:
        _dev(pred, "pred", torch.float32)
    b = box if isinstance(box, _lib.Box) else _lib.make_box(box, periodic)
    check(lib.htf_train_pair_grad_list(potential.handle, pos.data_ptr(), _dt(pos), B, int(NN), C.byref(b),
                                       _u32(n_neigh, "n_neigh").data_ptr(), _u32(nlist, "nlist").data_ptr(),
                                       _u32(head_list, "head_list").data_ptr(), float(r_cut), labels.data_ptr(), _dt(labels),
                                       pred.data_ptr() if pred is not None else None, accum.data_ptr(), scratch.data_ptr(), _stream(pos)))
    return accum


def optimizer_step(theta, accum, scale, state, desc):
    """Keras SGD / Adam / Nadam step on the device parameter vector (no host round trip).
    ``state``: optimizer_state_floats(P) zero-initialised floats."""
    P = int(theta.numel())
    if int(state.numel()) < optimizer_state_floats(P):
        raise ValueError("optimizer state too small for %d parameters" % P)
    fn = lib.htf_optimizer_step if P <= 8 else lib.htf_optimizer_step_n
    check(fn(theta.data_ptr(), P, accum.data_ptr(), float(scale), state.data_ptr(), C.byref(desc), _stream(theta)))
    return theta


def optimizer_state_floats(P):
    return _lib.OPT_STATE_FLOATS + (2 * int(P) if P > 8 else 0)


def add_virial(dest, src9, N, pitch):
    """receiveVirial (TensorflowCompute.cc:284-301)."""
    _dev(dest, "dest")
    _dev(src9, "src9", dest.dtype)
    check(lib.htf_add_virial(dest.data_ptr(), src9.data_ptr(), _dt(dest), int(N), int(pitch), _stream(dest)))
    return dest


def add_scalar4(dest, src):
    """sumReferenceForces (TensorflowCompute.cc:250-269)."""
    _dev(dest, "dest")
    _dev(src, "src", dest.dtype)
    check(lib.htf_add_scalar4(dest.data_ptr(), src.data_ptr(), _dt(dest), int(dest.shape[0]), _stream(dest)))
    return dest


def copy_positions(src, offset=0, N=None, unstuff4=True, out_dtype=None):
    """TFArrayComm::receiveArray (TFArrayComm.h:86-130)."""
    _dev(src, "src")
    N = int(src.shape[0]) - offset if N is None else int(N)
    out = torch.empty((N, 4), dtype=out_dtype or src.dtype, device=src.device)
    check(lib.htf_copy_positions(out.data_ptr(), _dt(out), src.data_ptr(), _dt(src), int(offset), N,
                                 1 if unstuff4 else 0, _stream(src)))
    return out


def positions_forces_radial(positions, coef=1.0, power=-1, ncomp=4, out=None, out_dtype=None):
    """compute_positions_forces(positions, coef * |positions[:, :ncomp]|^power) in one kernel -> [N, 4]."""
    _dev(positions, "positions")
    if positions.dim() != 2 or positions.shape[1] != 4:
        raise ValueError("positions must be [N, 4]")
    N = int(positions.shape[0])
    if out is None:
        out = torch.empty((N, 4), dtype=out_dtype or positions.dtype, device=positions.device)
    _dev(out, "out")
    check(lib.htf_positions_forces_radial(positions.data_ptr(), _dt(positions), N, int(ncomp), int(power), float(coef),
                                          out.data_ptr(), _dt(out), _stream(positions)))
    return out


def topk_desc(x, k):
    """tf.math.top_k(x, k) along the last axis of a [B, n] fp32 tensor: (values, indices), largest first,
    equal values in index order."""
    _dev(x, "x", torch.float32)
    B, n = int(x.shape[0]), int(x.shape[1])
    vals = torch.empty((B, int(k)), dtype=torch.float32, device=x.device)
    idx = torch.empty((B, int(k)), dtype=torch.int32, device=x.device)
    check(lib.htf_top_k(x.data_ptr(), B, n, int(k), vals.data_ptr(), idx.data_ptr(), _stream(x)))
    return vals, idx


def energy_sum(force, out=None):
    """calcEnergySum: sum of the energy column, in double, on the device."""
    _dev(force, "force")
    if out is None:
        out = torch.empty(1, dtype=torch.float64, device=force.device)
    check(lib.htf_energy_sum(force.data_ptr(), _dt(force), int(force.shape[0]), out.data_ptr(), _stream(force)))
    return out


def copy3(dest, src, N=None):
    """htf_gpu_copy3 (TFArrayComm.cu:31-56): dest[:N, :3] = src[:N, :3], dest's stuffed type kept."""
    _dev(dest, "dest")
    _dev(src, "src")
    n = int(src.shape[0]) if N is None else int(N)
    check(lib.htf_copy3(dest.data_ptr(), _dt(dest), src.data_ptr(), _dt(src), n, _stream(dest)))
    return dest


def check_nlist(nlist):
    """simmodel.py:214-219: max_i sum_j [nlist[i,j,0] > 0] (synchronises)."""
    _dev(nlist, "nlist")
    flag = torch.zeros(1, dtype=torch.int32, device=nlist.device)
    check(lib.htf_check_nlist(nlist.data_ptr(), _dt(nlist), int(nlist.shape[0]), int(nlist.shape[1]),
                              flag.data_ptr(), _stream(nlist)))
    return int(flag.item())


def nlist_rinv(nlist):
    """simmodel.py:618-635 -> [B, NN] fp32."""
    _dev(nlist, "nlist")
    out = torch.empty(nlist.shape[:2], dtype=torch.float32, device=nlist.device)
    check(lib.htf_nlist_rinv(nlist.data_ptr(), _dt(nlist), int(nlist.shape[0]), int(nlist.shape[1]),
                             out.data_ptr(), _stream(nlist)))
    return out


def stuff_types(pos_xyz, types, dtype=torch.float32):
    """Build a HOOMD position array: w carries the int type id's BITS
    (ParticleData: __int_as_scalar)."""
    N = pos_xyz.shape[0]
    out = torch.zeros((N, 4), dtype=dtype, device=pos_xyz.device)
    out[:, :3] = pos_xyz.to(dtype)
    t = types.to(torch.int32).contiguous()
    if dtype == torch.float32:
        out[:, 3] = t.view(torch.float32)
    else:
        out[:, 3] = t.to(torch.int64).view(torch.float64)
    return out


class Context:
    """htf_ctx: the TensorflowCompute object (TensorflowCompute.h:75-250).  ``fused``: 0 two
    kernels (build, then evaluate), 2 one kernel that writes the tensor and evaluates it in
    registers, 1 / True the same without writing the tensor."""

    def __init__(self, r_cut, nneighs, period=1, batch_size=0, scalar_dtype=torch.float32,
                 check_nlist=False, virial=False, max_n=0, force_mode=_lib.HTF_TF2HOOMD, fused=False):
        cfg = _lib.Config(float(r_cut), int(nneighs), int(force_mode), int(period), int(batch_size),
                          _lib.HTF_F64 if scalar_dtype == torch.float64 else _lib.HTF_F32,
                          int(bool(check_nlist)), int(bool(virial)), int(max_n), int(fused))
        self.cfg = cfg
        self.scalar_dtype = scalar_dtype
        self._h = C.c_void_p()
        self._pot = None
        self._device_index = torch.cuda.current_device()  # htf_create binds the context to the current device
        check(lib.htf_create(C.byref(cfg), C.byref(self._h)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and lib is not None:
            lib.htf_destroy(h)
            self._h = None

    def set_potential(self, pot):
        self._pot = pot  # keep alive: the C side only borrows it
        check(lib.htf_set_potential(self._h, pot.handle if pot is not None else None))

    def make_arrays(self, pos, n_local, n_neigh, head_list, nlist, box, force, virial=None, virial_pitch=0,
                    periodic=(1, 1, 1)):
        a = _lib.HoomdArrays()
        a.pos = _dev(pos, "pos", self.scalar_dtype).data_ptr()
        a.N = int(n_local)
        a.n_ghost = int(pos.shape[0]) - int(n_local)
        a.n_neigh = _u32(n_neigh, "n_neigh").data_ptr() if n_neigh is not None else None
        a.nlist = _u32(nlist, "nlist").data_ptr() if nlist is not None else None
        a.head_list = _u32(head_list, "head_list").data_ptr() if head_list is not None else None
        a.box = box if isinstance(box, _lib.Box) else _lib.make_box(box, periodic)
        a.force = _dev(force, "force", self.scalar_dtype).data_ptr()
        a.virial = _dev(virial, "virial", self.scalar_dtype).data_ptr() if virial is not None else None
        a.virial_pitch = int(virial_pitch)
        return a

    def compute_forces(self, timestep, arrays, stream=None, rows=None):
        """``rows=(begin, count)``: only those particle rows (htf_compute_forces_rows)."""
        s = stream if stream is not None else raw_stream(self._device_index)
        if rows is None:
            check(lib.htf_compute_forces(self._h, int(timestep), C.byref(arrays), C.c_void_p(s)))
        elif rows[1] > 0:
            check(lib.htf_compute_forces_rows(self._h, int(timestep), C.byref(arrays), int(rows[0]), int(rows[1]),
                                              C.c_void_p(s)))

    def compute_forces_overlapped(self, timestep, arrays, domain, stream=None):
        """One step under domain decomposition: interior rows (no ghost neighbors) while the
        ghost-position halo is in flight, boundary rows once it has landed."""
        if domain is None or not domain.pending:
            self.compute_forces(timestep, arrays, stream)
            return
        n_int = domain.n_interior
        self.compute_forces(timestep, arrays, stream, rows=(0, n_int))
        domain.exchange_end()
        self.compute_forces(timestep, arrays, stream, rows=(n_int, arrays.N - n_int))

    def set_step_epilogue(self, slot, vel, pos_next, dt, box, brick=None, row_slots=None, halo_send=None, ghost_direct=None):
        """Register a step epilogue (include/htf_standin.h htfs_step_epilogue: the stand-in integrator, and a brick's halo pack, done
        by the force kernel's own lanes) in descriptor ``slot`` (0 / 1: the two directions of a position ping-pong) -> whether this
        context's launches will honour it.  A blocking upload: set-up time."""
        e = _lib.StepEpilogue()
        e.d_vel, e.d_pos_next = _dev(vel, "vel", self.scalar_dtype).data_ptr(), _dev(pos_next, "pos_next", self.scalar_dtype).data_ptr()
        e.dtype, e.dt = self.cfg.scalar_dtype, float(dt)
        e.box = box
        self._epilogue_keep = getattr(self, "_epilogue_keep", {})
        self._epilogue_keep[slot] = (vel, pos_next, brick, row_slots, halo_send, ghost_direct)
        if brick is not None:
            e.brick = C.addressof(brick)
            e.d_row_slots = row_slots.data_ptr()
            e.d_halo_send = halo_send.data_ptr() if halo_send is not None else None
            e.d_ghost_direct = ghost_direct.data_ptr() if ghost_direct is not None else None
        ok = C.c_int(0)
        check(lib.htfs_set_step_epilogue(self._h, int(slot), C.byref(e), C.byref(ok)))
        return bool(ok.value)

    def use_step_epilogue(self, slot):
        """The descriptor every later compute_forces of this context carries (-1 / None: none)."""
        check(lib.htfs_use_step_epilogue(self._h, -1 if slot is None else int(slot)))

    def profile_enable(self, on=True):
        """Event-bracket the build and eval scopes (HOOMD Profiler analogue); ``on=k`` (int > 1)
        brackets every k-th batch only."""
        check(lib.htf_profile_enable(self._h, int(on)))

    def profile_read(self):
        """-> (build_ms_total, eval_ms_total, n_calls) since the last read."""
        b, e, n = C.c_double(), C.c_double(), C.c_uint()
        check(lib.htf_profile_read(self._h, C.byref(b), C.byref(e), C.byref(n)))
        return b.value, e.value, n.value

    def _view(self, ptr, shape, dtype, device):
        n = int(np.prod(shape))
        if not ptr or n == 0:
            return torch.empty(shape, dtype=dtype, device=device)
        class _Holder:
            pass
        h = _Holder()
        typestr = {torch.float32: "<f4", torch.float64: "<f8"}[dtype]
        h.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                      "version": 2, "strides": None}
        return torch.as_tensor(h, device=device)

    def nlist_buffer(self, B, device="cuda"):
        """getNlistBuffer (TensorflowCompute.cc:406): zero-copy [B, NN, 4] fp32 view.  Read-only between steps
        (see reset_nlist_buffer)."""
        return self._view(lib.htf_get_nlist_buffer(self._h), (B, self.cfg.nneighs, 4), torch.float32, device)

    def reset_nlist_buffer(self):
        """Call after WRITING into ``nlist_buffer()``: the next step rewrites every row's zero tail in full
        (the context otherwise re-zeroes only the slots a row lost since the previous step)."""
        check(lib.htf_reset_nlist_buffer(self._h, C.c_void_p(raw_stream(self._device_index))))

    def positions_buffer(self, B, device="cuda"):
        return self._view(lib.htf_get_positions_buffer(self._h), (B, 4), torch.float32, device)

    def virial_buffer(self, B, device="cuda"):
        return self._view(lib.htf_get_virial_buffer(self._h), (B, 9), self.scalar_dtype, device)
