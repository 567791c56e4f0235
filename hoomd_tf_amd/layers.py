"""Counterpart of ``hoomd/htf/layers.py``: RBFExpansion, WCARepulsion, EDSLayer, plus the
declarative pair-MLP (RBF -> Dense -> Dense -> Dense) that config C3 evaluates on MFMA."""
import ctypes as C

import numpy as np
import torch

from . import _lib, ops, simmodel
from ._lib import lib, check
from .initializers import mlp_params


class RBFExpansion:
    """layers.py:7-49: ``exp(-(d - mu)^2 / gap)`` on an evenly spaced grid of ``count``
    centres from ``low`` to ``high`` (inclusive); gap = centre spacing (not squared)."""

    def __init__(self, low, high, count):
        self.low, self.high, self.count = float(low), float(high), int(count)
        self.centers = np.linspace(self.low, self.high, self.count).astype(np.float32)
        self.gap = np.float32(self.centers[1] - self.centers[0])

    def get_config(self):
        return {'low': self.low, 'high': self.high, 'count': self.count}

    def __call__(self, inputs):
        if isinstance(inputs, simmodel.SafeNorm) or (isinstance(inputs, torch.Tensor) and inputs.requires_grad):
            # the distances of compute()'s neighbor tensor (or anything else on an autograd graph): the expansion in torch ops, so
            # that whatever the model builds from it -- a descriptor summed over the neighbors and fed to Dense layers, a per-pair
            # network written by hand -- is differentiable by compute_nlist_forces (the reference's layer is a TF op like any
            # other: layers.py:36-49).  [.., count] floats: the generic route, not a fused kernel (htf.PairMLP is that).
            if isinstance(inputs, simmodel.SafeNorm):
                t = inputs.nlist.ad[:, :, :3] + inputs.delta
                inputs = torch.sqrt((t * t).sum(dim=2))
            simmodel._trace_log().append({"op": "rbf"})
            c = torch.as_tensor(self.centers, dtype=inputs.dtype, device=inputs.device)
            d = inputs[..., None] - c
            return torch.exp(-(d * d) / float(self.gap))
        x = inputs.to(torch.float32).contiguous()
        ops._dev(x, "inputs")
        simmodel._trace_log().append({"op": "rbf"})
        out = torch.empty(tuple(x.shape) + (self.count,), dtype=torch.float32, device=x.device)
        check(lib.htf_rbf_expansion(x.data_ptr(), x.numel(), self.low, self.high, self.count, out.data_ptr(),
                                    ops._stream(x)))
        return out


class Dense:
    """tf.keras.layers.Dense for the per-particle networks of the force path (example 08): ``x W + b``,
    glorot-uniform kernel, zero bias, ``activation=None`` unless given ('tanh').  Built on first call, like
    Keras.  Applied to the symbolic top-k features it stays symbolic (three of them lower to one kernel);
    applied to a tensor it is a torch matmul."""

    def __init__(self, units, activation=None, seed=0):
        self.units, self.activation, self.seed = int(units), activation, int(seed)
        self.kernel = self.bias = None
        self.version = 0

    def build(self, fan_in):
        if self.kernel is None:
            rng = np.random.default_rng(self.seed)
            lim = np.sqrt(6.0 / (fan_in + self.units))
            self.kernel = rng.uniform(-lim, lim, size=(fan_in, self.units)).astype(np.float32)
            self.bias = np.zeros(self.units, dtype=np.float32)

    def get_weights(self):
        return [] if self.kernel is None else [self.kernel.copy(), self.bias.copy()]

    def set_weights(self, ws):
        k, b = np.asarray(ws[0], dtype=np.float32), np.asarray(ws[1], dtype=np.float32)
        if k.shape[1] != self.units or b.shape != (self.units,):
            raise ValueError("Dense(%d): weight shapes %r, %r do not fit" % (self.units, k.shape, b.shape))
        self.kernel, self.bias = k.copy(), b.copy()
        self.version += 1

    def __call__(self, x):
        if isinstance(x, simmodel.TopRinv):
            self.build(x.k)
            return simmodel.DenseOut(x, [self])
        if isinstance(x, simmodel.DenseOut):
            self.build(x.layers[-1].units)
            return simmodel.DenseOut(x.top, x.layers + [self])
        t = x.tensor() if hasattr(x, "tensor") and callable(x.tensor) else x
        self.build(int(t.shape[-1]))
        y = t @ torch.as_tensor(self.kernel, dtype=t.dtype, device=t.device) + torch.as_tensor(self.bias, dtype=t.dtype, device=t.device)
        return torch.tanh(y) if self.activation == "tanh" else y


class WCARepulsion:
    """layers.py:52-98: trainable WCA repulsion ``(sigma/r)^6`` inside ``2^(1/3) sigma``,
    clipped to [0, 10].  Called on the neighbor list; returns the pair energy.  ``sigma`` is
    a trainable scalar weight with regulariser ``-regularization_strength * sigma``; it lives
    on the device once the layer takes part in training."""
    name = 'wca-repulsion'

    def __init__(self, sigma, regularization_strength=1e-3):
        self._sigma0 = float(np.float32(sigma))
        self.regularization_strength = regularization_strength
        self.w = None  # device weight, created on first training use

    @property
    def sigma(self):
        return float(self.w[0]) if self.w is not None else self._sigma0

    def get_config(self):
        return {'sigma': float(self.sigma)}

    def get_weights(self):
        return [np.array([self.sigma], dtype=np.float32)]

    def set_weights(self, ws):
        v = float(np.asarray(ws[0]).reshape(-1)[0])
        if self.w is not None:
            self.w[0] = v
        self._sigma0 = float(np.float32(v))

    # trainable-layer protocol used by tfcompute's training step
    nonneg_mask = 0

    @property
    def l1_reg(self):
        return (-float(self.regularization_strength),)

    def make_trainable(self, device):
        if self.w is None:
            self.w = torch.tensor([self._sigma0], dtype=torch.float32, device=device)
        return self.w

    @property
    def trainable_weights(self):
        return [self.w] if self.w is not None else []

    def potential(self):
        """The layer's own potential: kept while the layer lives, rebuilt when sigma is reset on the host
        or the device weight appears."""
        key = ("theta", id(self.w)) if self.w is not None else ("sigma", self._sigma0)
        if getattr(self, "_pot", None) is None or self._pot[0] != key:
            self._pot = (key, ops.Potential.wca(self.sigma, theta=self.w))
        return self._pot[1]

    def __call__(self, nlist):
        return simmodel.WCAPair(simmodel._as_nlist(nlist), self.sigma, layer=self)


class LJLayer:
    """The trainable Lennard-Jones layer of example 06 and build_examples.py:336-359: weights
    ``w = [sig, eps]`` (NonNeg constraint); called on ``r = safe_norm(nlist[:, :, :3], axis=2)``
    it returns the pair energy ``w[0] * 4 * (r6**2 - r6) / 2`` with ``r6 = w[1]**6 / r**6``."""
    name = 'lj'
    nonneg_mask = 0b11
    l1_reg = (0.0, 0.0)

    def __init__(self, sig, eps, device="cuda"):
        self.start = [sig, eps]
        self.w = torch.tensor([sig, eps], dtype=torch.float32, device=device)

    def get_config(self):
        return {'sig': self.start[0], 'eps': self.start[1]}

    def get_weights(self):
        return [self.w.detach().cpu().numpy().copy()]

    def set_weights(self, ws):
        self.w.copy_(torch.as_tensor(np.asarray(ws[0], dtype=np.float32).reshape(2)))

    def make_trainable(self, device):
        return self.w

    @property
    def trainable_weights(self):
        return [self.w]

    def potential(self):
        """The layer's own potential (reads ``self.w`` on the device at every launch)."""
        if getattr(self, "_pot", None) is None:
            w = self.w.cpu().numpy()
            self._pot = ops.Potential.lj_param(float(w[0]), float(w[1]), theta=self.w)
        return self._pot

    def __call__(self, r):
        if not isinstance(r, simmodel.SafeNorm):
            raise ValueError("LJLayer expects r = safe_norm(nlist[:, :, :3], axis=2)")
        return simmodel.LJParamEnergy(r.nlist, self)


class SoftRDFCV:
    """Differentiable stand-in for one RDF bin: cv = (1/N) sum_i sum_j
    exp(-(r_ij - r0)^2 / gap), r = safe_norm, padded slots masked -- a single RBFExpansion
    channel summed over the neighbor list.  Call it on the nlist; feed the result to EDSLayer."""

    def __init__(self, r0, gap):
        self.r0, self.gap = float(r0), float(gap)

    def get_config(self):
        return {'r0': self.r0, 'gap': self.gap}

    def __call__(self, nlist):
        return simmodel.PairCV(simmodel._as_nlist(nlist), self.r0, self.gap)


class PairMLP:
    """safe_norm -> RBFExpansion(low, high, K) -> Dense(H1) -> Dense(H2) -> Dense(1), masked
    with the nlist_rinv criterion and halved per pair (SURVEY 8(a) closed forms).  Keras
    Dense defaults: glorot-uniform kernels, zero biases, ``activation=None``; pass
    ``activation='tanh'`` for the C3 benchmark model.
    ``precision``: how the dense layers meet the matrix cores.  ``"split16"`` (default): every fp32 operand as hi + lo in
    fp16 (2^-22), three partial products on the fp16 MFMA -- the fp32 evaluator's accuracy (same tolerances against the
    fp64 oracle, tests/test_gpu_parity.py::test_pair_mlp_split_operands) at 2.5x its speed; weights and activations must
    stay inside fp16's range (|x| < 6e4: always true of a tanh network with sane weights).  ``"fp32"``: exact fp32 products
    on the fp32 MFMA.  ``"split"``: three bf16 parts, six partial products (no range limit).  ``"bf16"``: plain bf16
    operands, reduced precision."""

    name = 'pair-mlp'
    nonneg_mask = 0
    l1_reg = (0.0,)
    _KEYS = ("W1", "b1", "W2", "b2", "W3", "b3")

    def __init__(self, K=32, H1=64, H2=64, low=0.0, high=3.0, activation=None, seed=3, precision="split16"):
        self.low, self.high = float(low), float(high)
        self.activation = activation or "linear"
        self.precision = precision
        self.params = mlp_params(seed=seed, K=K, H1=H1, H2=H2)
        self.w = None      # flat device weights (Keras get_weights() order) once the potential exists
        self._pot = None

    def _flat(self):
        return np.concatenate([np.asarray(self.params[k], dtype=np.float32).ravel() for k in self._KEYS])

    def make_trainable(self, device="cuda"):
        if self.w is None:
            self.w = torch.tensor(self._flat(), dtype=torch.float32, device=device)
        return self.w

    @property
    def trainable_weights(self):
        return [self.w] if self.w is not None else []

    def potential(self):
        """One persistent potential reading ``self.w``: an optimizer step or set_weights only
        rebuilds its operand images on the device (htf_potential_refresh)."""
        if self._pot is None:
            self._pot = ops.Potential.pair_mlp(self.params, self.low, self.high, activation=self.activation,
                                               precision=self.precision, theta=self.make_trainable())
        return self._pot

    def after_update(self):
        if self._pot is not None:
            self._pot.refresh()

    def _sync_host(self):
        if self.w is None:
            return
        flat, o = self.w.detach().cpu().numpy(), 0
        for k in self._KEYS:
            n = self.params[k].size
            self.params[k] = flat[o:o + n].reshape(self.params[k].shape).copy()
            o += n

    def get_weights(self):
        self._sync_host()
        return [self.params[k].copy() for k in self._KEYS]

    def set_weights(self, ws):
        for k, w in zip(self._KEYS, ws):
            if np.shape(w) != self.params[k].shape:
                raise ValueError("shape mismatch for %s" % k)
        for k, w in zip(self._KEYS, ws):
            self.params[k] = np.asarray(w, dtype=np.float32).copy()
        if self.w is not None:
            self.w.copy_(torch.from_numpy(self._flat()))
            self.after_update()

    def save_weights(self, path):
        self._sync_host()
        np.savez(path, **self.params)

    def load_weights(self, path):
        with np.load(path) as z:
            self.set_weights([z[k] for k in self._KEYS])

    def __call__(self, nlist):
        return simmodel.MLPEnergy(simmodel._as_nlist(nlist), self)


class EDSLayer:
    """layers.py:101-195.  Call it on the collective variable every step; returns alpha,
    the EDS coupling constant.  The running mean / ssd / TF1-Adam state lives on the
    device and advances in a one-thread kernel, so a device-resident CV never visits
    the host."""

    def __init__(self, set_point, period, learning_rate=1e-2, cv_scale=1.0, name='eds-layer', device="cuda"):
        if isinstance(set_point, (int, np.integer)) and not isinstance(set_point, bool):
            raise ValueError('EDS only works with floats, not dtype' + str(type(set_point)))
        self.set_point = float(set_point)
        self.period = int(period)
        self.cv_scale = float(cv_scale)
        self.learning_rate = float(learning_rate)
        self.name = name
        self.state = torch.zeros(8, dtype=torch.float32, device=device)

    def get_config(self):
        return {'set_point': self.set_point, 'period': self.period, 'cv_scale': self.cv_scale,
                'learning_rate': self.learning_rate}

    @property
    def alpha(self):
        return self.state[2]

    @property
    def mean(self):
        return self.state[0]

    @property
    def n(self):
        return int(self.state[5].item())

    def __call__(self, cv):
        if isinstance(cv, simmodel.PairCV):
            return simmodel.DeferredAlpha(self, cv)  # advances when the biased energy is lowered
        if not isinstance(cv, torch.Tensor):
            cv = torch.tensor(float(cv), dtype=torch.float32, device=self.state.device)
        cv = cv.detach().to(torch.float32).reshape(1).contiguous()
        check(lib.htf_eds_update(self.state.data_ptr(), cv.data_ptr(), self.set_point, self.period,
                                 self.learning_rate, self.cv_scale, ops._stream(self.state)))
        simmodel._trace_log().append({"stateful": self.name})
        return self.state[2]
