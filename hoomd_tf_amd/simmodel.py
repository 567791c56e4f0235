"""Model API of the force path: the counterpart of ``hoomd/htf/simmodel.py``.

The reference lets users write ``compute(nlist, positions, box)`` in TensorFlow and
differentiates the energy with ``tf.gradients``.  Here the same method bodies are
written against a small *declarative* expression layer: ``nlist_rinv``, ``safe_norm``,
``RBFExpansion``, ``WCARepulsion``, ``PairMLP`` return symbolic pair-energy
expressions, and ``compute_nlist_forces`` lowers the expression to ONE fused HIP kernel
(energy, analytic gradient, x2, neighbor sum, energy column, optional virial) -- the
analogue of ``tf.function`` tracing ``compute`` into a graph.  Names, argument meaning
and error behaviour follow the reference (cited per function).
"""
import ctypes as C
import os
import threading

import numpy as np
import torch

from . import _lib, ops
from ._lib import lib, check, NlistOverflowError, SkewedBoxError  # noqa: F401

_trace = threading.local()


def _trace_log():
    if not hasattr(_trace, "calls"):
        _trace.calls = []
    return _trace.calls


# --------------------------------------------------------------------------- tensors
def _unwrap(x):
    if isinstance(x, _TorchOperand):
        return x.ad
    if isinstance(x, (list, tuple)):
        return type(x)(_unwrap(v) for v in x)
    return x


class _TorchOperand:
    """Lets the neighbor tensor be used in ordinary torch code (``torch.norm(nlist[:, :, :3],
    dim=2)``, arithmetic, slicing): the generic-model route of SURVEY 8(f)-3.  Any torch
    function sees ``.ad`` -- the pair-vector buffer as an autograd leaf -- so
    ``compute_nlist_forces`` can differentiate whatever the model builds from it."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        return func(*_unwrap(args), **{k: _unwrap(v) for k, v in (kwargs or {}).items()})

    def __add__(self, o): return self.ad + _unwrap(o)
    def __radd__(self, o): return _unwrap(o) + self.ad
    def __sub__(self, o): return self.ad - _unwrap(o)
    def __rsub__(self, o): return _unwrap(o) - self.ad
    def __mul__(self, o): return self.ad * _unwrap(o)
    def __rmul__(self, o): return _unwrap(o) * self.ad
    def __truediv__(self, o): return self.ad / _unwrap(o)
    def __rtruediv__(self, o): return _unwrap(o) / self.ad
    def __pow__(self, o): return self.ad ** _unwrap(o)

    def __neg__(self):
        return -self.ad


class Nlist(_TorchOperand):
    """The ``N x NN x 4`` neighbor tensor handed to ``compute`` (simmodel.py:99-105):
    a zero-copy view of the pair-vector buffer plus the identity the expression layer
    needs.  ``nlist[:, :, :3]`` stays symbolic for the declarative layers; torch code gets
    the autograd leaf ``ad``."""

    def __init__(self, tensor):
        self.tensor = tensor
        self._ad = None
        self._weights = []      # the scalar weights traced expressions of this list read: [(leaf tensor, flat index), ...] -> p.theta[k]

    shape = property(lambda self: self.tensor.shape)
    dtype = property(lambda self: self.tensor.dtype)
    device = property(lambda self: self.tensor.device)

    @property
    def ad(self):
        if self._ad is None:
            self._ad = self.tensor.detach().requires_grad_(True)
        return self._ad

    def weight_index(self, leaf, flat):
        """k of (leaf, flat) in this trace's weight vector (appended on first sight)."""
        for k, (t, i) in enumerate(self._weights):
            if t is leaf and i == flat:
                return k
        self._weights.append((leaf, int(flat)))
        return len(self._weights) - 1

    def __getitem__(self, idx):
        full = slice(None)
        if isinstance(idx, tuple) and len(idx) == 3 and idx[0] == full and idx[1] == full and idx[2] == slice(None, 3):
            return NlistXYZ(self)
        if (isinstance(idx, tuple) and len(idx) == 3 and idx[0] == full and idx[1] == full and isinstance(idx[2], int)
                and not isinstance(idx[2], bool) and idx[2] == 3):
            from . import codegen as cg
            return TypeExpr(cg.TJ, nlist=self)   # the neighbors' types: symbolic for traced energies, a tensor to torch code
        return self.ad[idx]

    def numpy(self):
        return self.tensor.cpu().numpy()


class NlistXYZ(_TorchOperand):
    """``nlist[:, :, :3]``"""

    def __init__(self, parent):
        self.parent = parent

    @property
    def tensor(self):
        return self.parent.tensor[:, :, :3]

    @property
    def ad(self):
        return self.parent.ad[:, :, :3]

    def numpy(self):
        return self.tensor.cpu().numpy()


class Positions(_TorchOperand):
    """The ``N x 4`` positions tensor (x, y, z, type) handed to ``compute`` of a positions-only model
    (nneighbor_cutoff = 0): behaves like the torch tensor it wraps -- any torch function, arithmetic,
    indexing or attribute works on the autograd leaf ``ad`` -- while ``htf.norm(positions, axis=1)`` stays
    symbolic so that ``compute_positions_forces`` of a radial energy lowers to one kernel."""

    def __init__(self, tensor):
        self.tensor = tensor
        self._ad = None

    shape = property(lambda self: self.tensor.shape)
    dtype = property(lambda self: self.tensor.dtype)
    device = property(lambda self: self.tensor.device)

    @property
    def ad(self):
        if self._ad is None:
            self._ad = self.tensor.detach().requires_grad_(True)
        return self._ad

    def __getitem__(self, idx):
        return self.ad[idx]

    def __len__(self):
        return int(self.tensor.shape[0])

    def __getattr__(self, name):  # .detach(), .cpu(), .sum(...): whatever a tensor offers
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.ad, name)

    def numpy(self):
        return self.tensor.cpu().numpy()


class PositionsInput(torch.Tensor):
    """The ``N x 4`` positions tensor handed to ``compute`` of a neighbor-list model: a torch tensor in every respect but one --
    ``positions[:, 3]``, the particles' own types, keeps its identity (:class:`TypeExpr`) so that a traced pair energy can look
    parameters up by species pair (``htf.gather(table, ti[:, None] * ntypes + tj)``) inside a generated kernel.  Everything
    computed from it is a plain tensor."""

    @staticmethod
    def wrap(t):
        out = t.as_subclass(PositionsInput)
        out._htf_input = True
        return out

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **(kwargs or {}))

    def plain(self):
        return self.as_subclass(torch.Tensor)

    def __getitem__(self, idx):
        if (getattr(self, "_htf_input", False) and isinstance(idx, tuple) and len(idx) == 2 and idx[0] == slice(None)
                and isinstance(idx[1], int) and not isinstance(idx[1], bool) and idx[1] == 3):
            from . import codegen as cg
            return TypeExpr(cg.TI, positions=self.plain(), expanded=False)
        return self.plain()[idx]


class TypeExpr(_TorchOperand):
    """An expression of particle TYPES only: ``nlist[:, :, 3]`` (the neighbors', [N, NN]), ``positions[:, 3]`` (the row particles'
    own, [N]; ``[:, None]`` makes it [N, 1]) and integer arithmetic on them -- the index of a parameter table.  Symbolic until it
    meets a traced pair expression, ``htf.gather`` or a comparison; torch code sees the tensor it stands for."""

    def __init__(self, node, nlist=None, positions=None, expanded=True):
        self.node, self.nlist, self.positions, self.expanded = node, nlist, positions, expanded

    __hash__ = object.__hash__

    @property
    def pair_shaped(self):
        return self.nlist is not None

    @property
    def ad(self):
        from . import codegen as cg
        tj = self.nlist.tensor[:, :, 3] if self.nlist is not None else None
        ti = None
        if self.positions is not None:
            ti = self.positions[:, 3]
            ti = ti[:, None] if (self.expanded or self.nlist is not None) else ti
        ref = tj if tj is not None else ti
        return cg.evaluate(self.node, ref, ref, ref, tj=tj, ti=ti)

    def tensor(self):
        return self.ad

    def __getitem__(self, idx):
        if self.nlist is None and not self.expanded and idx in ((slice(None), None), (Ellipsis, None)):
            return TypeExpr(self.node, None, self.positions, expanded=True)
        return self.ad[idx]

    def _join(self, o):
        if self.nlist is not None and o.nlist is not None and o.nlist is not self.nlist and o.nlist.tensor is not self.nlist.tensor:
            raise ValueError("expressions come from different neighbor lists")
        for t in (self, o):
            if t.nlist is None and not t.expanded and (self.nlist is not None or o.nlist is not None):
                raise ValueError("positions[:, 3] is [N]: write positions[:, 3][:, None] to combine it with per-neighbor values")
        return (self.nlist or o.nlist, self.positions if self.positions is not None else o.positions,
                self.expanded and o.expanded)

    def _bin(self, op, other, swap=False):
        from . import codegen as cg
        if isinstance(other, (PairExpr, RinvPoly, SafeNorm, PairNorm, PairMask)):
            e = PairExpr.of(other)
            me = PairExpr.of(self, e.nlist)
            return e._with(op, me) if swap else me._with(op, e)
        if isinstance(other, TypeExpr):
            nl, pos, ex = self._join(other)
            args = (other.node, self.node) if swap else (self.node, other.node)
            return TypeExpr(cg.Node(op, args), nl, pos, ex)
        if isinstance(other, (int, float, np.integer, np.floating)) and not isinstance(other, bool):
            args = (cg.const(other), self.node) if swap else (self.node, cg.const(other))
            return TypeExpr(cg.Node(op, args), self.nlist, self.positions, self.expanded)
        a, b = (_unwrap(other), self.ad) if swap else (self.ad, _unwrap(other))   # a tensor: torch from here on
        return {"add": torch.add, "sub": torch.sub, "mul": torch.mul, "div": torch.div}[op](torch.as_tensor(a), torch.as_tensor(b))

    def __add__(self, o): return self._bin("add", o)
    def __radd__(self, o): return self._bin("add", o, swap=True)
    def __sub__(self, o): return self._bin("sub", o)
    def __rsub__(self, o): return self._bin("sub", o, swap=True)
    def __mul__(self, o): return self._bin("mul", o)
    def __rmul__(self, o): return self._bin("mul", o, swap=True)
    def __truediv__(self, o): return self._bin("div", o)
    def __rtruediv__(self, o): return self._bin("div", o, swap=True)

    def _cmp(self, op, o):
        from . import codegen as cg
        if isinstance(o, TypeExpr) and (self.nlist is not None or o.nlist is not None):
            nl, pos, _ = self._join(o)
            return PairCond(nl, cg.Node(op, (self.node, o.node)), pos)
        if isinstance(o, (int, float, np.integer, np.floating)) and not isinstance(o, bool) and self.nlist is not None:
            return PairCond(self.nlist, cg.Node(op, (self.node, cg.const(o))), self.positions)
        # (the particles' own types alone -- ``positions[:, 3] == 0`` -- are not a per-pair condition: the tensor's comparison)
        return {"lt": torch.lt, "le": torch.le, "gt": torch.gt, "ge": torch.ge, "eq": torch.eq, "ne": torch.ne}[op](self.ad, _unwrap(o))

    def __lt__(self, o): return self._cmp("lt", o)
    def __le__(self, o): return self._cmp("le", o)
    def __gt__(self, o): return self._cmp("gt", o)
    def __ge__(self, o): return self._cmp("ge", o)
    def __eq__(self, o): return self._cmp("eq", o)
    def __ne__(self, o): return self._cmp("ne", o)

    # what a tensor offers (``.long()``, ``.shape`` ...): torch code gets the tensor
    def __getattr__(self, name):
        if name.startswith("_") or name in ("node", "nlist", "positions", "expanded"):
            raise AttributeError(name)
        return getattr(self.ad, name)

    def numpy(self):
        return self.ad.cpu().numpy()


class PosNorm(_TorchOperand):
    """tf.norm(positions, axis=1) of the whole row (all four columns, as the reference's models take it)
    or of ``positions[:, :3]``."""

    def __init__(self, positions, ncomp):
        self.positions, self.ncomp = positions, ncomp

    @property
    def ad(self):
        return torch.sqrt((self.positions.ad[:, :self.ncomp] ** 2).sum(dim=1))


class PosRadial(_TorchOperand):
    """coef * |positions|^power, per particle: the energy of BenchmarkNonlistModel (build_examples.py:59-64)."""

    def __init__(self, norm, coef, power):
        self.norm, self.coef, self.power = norm, float(coef), int(power)

    @property
    def ad(self):
        n = self.norm.ad
        if self.power < 0:  # divide_no_nan
            safe = torch.where(n > 0, n, torch.ones_like(n))
            return torch.where(n > 0, self.coef * safe ** self.power, torch.zeros_like(n))
        return self.coef * n ** self.power

    def __mul__(self, o):
        if isinstance(o, (int, float)):
            return PosRadial(self.norm, self.coef * o, self.power)
        return self.ad * _unwrap(o)

    __rmul__ = __mul__


class PairNorm(_TorchOperand):
    """``tf.norm(nlist[:, :, :3], axis=2)``: the plain pair distance.  Symbolic so that ``r < cut`` can become the hard
    mask of a closed-form energy (examples/01. Quickstart.ipynb cell 3); torch code gets the autograd value ``ad``."""

    def __init__(self, xyz):
        self.xyz = xyz

    @property
    def ad(self):
        t = self.xyz.ad
        return torch.sqrt((t * t).sum(dim=2))

    @property
    def tensor(self):
        t = self.xyz.tensor
        return torch.sqrt((t * t).sum(dim=2))

    def __lt__(self, cut):
        if isinstance(cut, (int, float)):
            return PairMask(self.xyz.parent, float(cut))
        return self.ad < _unwrap(cut)

    def __gt__(self, o): return self.ad > _unwrap(o)
    def __le__(self, o): return self.ad <= _unwrap(o)
    def __ge__(self, o): return self.ad >= _unwrap(o)


class PairMask(_TorchOperand):
    """``r < cut`` (and its ``tf.cast(..., tf.float32)``): multiplies a per-pair rinv polynomial into its truncated form --
    one kernel (HTF_POT_RINV_POLY with ``poly_cut``).  Everything else sees the bool tensor: it is a ``_TorchOperand``
    (``torch.where(r < cut, ...)``, ``(r < cut).to(...)``, returning it as an output), with the boolean operators a
    ``a < r < b`` shell needs."""

    def __init__(self, nlist, cut, as_dtype=None):
        # as_dtype: what ``cast(mask, dtype)`` asked for.  The mask stays symbolic (so that mask * rinv-polynomial still lowers
        # to one kernel) but everything that falls through to torch -- ``1.0 - cast(r < cut, float32)``, ``-mask``,
        # ``mask ** 2``, ``mask * mask`` -- sees a tensor of that dtype, as TensorFlow's cast result would be (bool
        # arithmetic raised, or stayed bool)
        self.nlist, self.cut, self._as = nlist, cut, as_dtype

    def tensor(self):
        t = self.nlist.tensor[:, :, :3]
        return (torch.sqrt((t * t).sum(dim=2)) < self.cut)

    @property
    def ad(self):
        m = self.tensor()
        return m if self._as is None else m.to(self._as)

    shape = property(lambda self: self.nlist.tensor.shape[:2])
    dtype = property(lambda self: torch.bool if self._as is None else self._as)
    device = property(lambda self: self.nlist.tensor.device)

    def __mul__(self, o):
        if isinstance(o, RinvPoly) and o.nlist is self.nlist and not o.reduced and o.cut in (None, self.cut):
            return RinvPoly(o.nlist, o.terms, cut=self.cut)
        if isinstance(o, PairMask):
            both = self.tensor() & o.tensor()
            return both if self._as is None and o._as is None else both.to(self._as or o._as)
        return self.tensor().to(self._as or torch.float32) * _unwrap(o)

    __rmul__ = __mul__

    def __and__(self, o): return self.tensor() & _unwrap(o)
    def __or__(self, o): return self.tensor() | _unwrap(o)
    def __xor__(self, o): return self.tensor() ^ _unwrap(o)
    __rand__, __ror__, __rxor__ = __and__, __or__, __xor__
    def __invert__(self): return ~self.tensor()
    def __getitem__(self, idx): return self.tensor()[idx]
    def to(self, *a, **k): return self.tensor().to(*a, **k)
    def float(self): return self.tensor().float()
    def numpy(self): return self.tensor().cpu().numpy()


def cast(x, dtype=None):
    """tf.cast for model code: a symbolic mask stays symbolic, tensors are converted."""
    if isinstance(x, PairCond):
        from . import codegen as cg
        return PairExpr(x.nlist, cg.Node("mask", (x.node,)), positions=x.positions, folded=x.folded)
    if isinstance(x, TypeExpr):
        return x   # (types are small integers held exactly as floats: an int cast changes nothing the kernel sees)
    if isinstance(x, PairMask):
        return x if dtype is None or dtype == torch.bool else PairMask(x.nlist, x.cut, as_dtype=dtype)
    t = _unwrap(x)
    return t.to(dtype) if dtype is not None and isinstance(t, torch.Tensor) else t


def norm(tensor, axis=None):
    """tf.norm for model code: symbolic on the positions tensor (axis=1) and on ``nlist[:, :, :3]`` (axis=2), torch
    otherwise."""
    if isinstance(tensor, Positions) and axis in (1, -1):
        return PosNorm(tensor, 4)
    if isinstance(tensor, NlistXYZ) and axis in (2, -1):
        return PairNorm(tensor)
    t = _unwrap(tensor)
    return torch.sqrt((t * t).sum()) if axis is None else torch.sqrt((t * t).sum(dim=axis))


def divide_no_nan(x, y):
    """tf.math.divide_no_nan: x / y, 0 where y == 0."""
    if isinstance(y, PosNorm) and isinstance(x, (int, float)):
        return PosRadial(y, x, -1)
    x, y = _unwrap(x), _unwrap(y)
    y = torch.as_tensor(y)
    safe = torch.where(y != 0, y, torch.ones_like(y))
    return torch.where(y != 0, x / safe, torch.zeros_like(safe))


def _as_nlist(x):
    if isinstance(x, Nlist):
        return x
    if isinstance(x, torch.Tensor):
        return Nlist(x)
    raise ValueError("expected the nlist tensor")


# --------------------------------------------------------------------------- expressions
class PairEnergy:
    """Base of the symbolic per-pair energies.  ``reduced`` marks a per-particle sum."""
    reduced = False

    def potential(self):
        raise NotImplementedError

    def key(self):
        raise NotImplementedError


class RinvPoly(PairEnergy):
    """sum_k c_k * rinv^p_k with rinv = nlist_rinv(nlist)."""

    def __init__(self, nlist, terms, reduced=False, cut=None):
        self.nlist = nlist
        self.terms = {p: c for p, c in terms.items() if c != 0}
        self.reduced = reduced
        self.cut = cut  # hard mask norm(nlist[:, :, :3]) < cut (no gradient), None: unmasked

    def _bin(self, other, sign):
        if isinstance(other, RowExpr) or (self.reduced and not (isinstance(other, RinvPoly) and other.reduced and other.cut == self.cut)):
            # a per-particle sum meeting anything but another sum of the same kind: a row expression (RowExpr)
            return RowExpr.of(self)._with("add" if sign > 0 else "sub", other)
        if isinstance(other, RinvPoly):
            if other.nlist is not self.nlist:
                raise ValueError("expressions come from different neighbor lists")
            if other.cut != self.cut:
                raise ValueError("cannot add rinv polynomials with different masks in one kernel")
            t = dict(self.terms)
            for p, c in other.terms.items():
                t[p] = t.get(p, 0.0) + sign * c
            return RinvPoly(self.nlist, t, self.reduced and other.reduced, cut=self.cut)
        if isinstance(other, (PairExpr, SafeNorm, torch.Tensor)) or (isinstance(other, (int, float)) and not self.reduced):
            # not a polynomial in rinv any more: a traced expression (generated kernel); with a tensor -- a live weight or per-pair
            # values -- its torch value (the autograd route)
            e = PairExpr.of(self)
            return e._with("add" if sign > 0 else "sub", other)
        raise TypeError("cannot combine a rinv polynomial with %r (constants carry no force)" % (other,))

    def __add__(self, o):
        if isinstance(o, BiasTerm):
            return BiasedEnergy(self, o)
        return self._bin(o, 1.0)

    __radd__ = __add__

    def __sub__(self, o):
        return self._bin(o, -1.0)

    def __neg__(self):
        return RinvPoly(self.nlist, {p: -c for p, c in self.terms.items()}, self.reduced, cut=self.cut)

    def __mul__(self, o):
        if isinstance(o, (int, float)):
            return RinvPoly(self.nlist, {p: c * o for p, c in self.terms.items()}, self.reduced, cut=self.cut)
        if self.reduced or isinstance(o, RowExpr) or (isinstance(o, (RinvPoly, PairExpr)) and o.reduced):
            return RowExpr.of(self)._with("mul", o)    # (per-particle sums: a row expression)
        if isinstance(o, PairMask):
            return o.__mul__(self)
        if isinstance(o, RinvPoly):
            if o.nlist is not self.nlist or self.reduced or o.reduced:
                raise ValueError("can only multiply per-pair expressions of one neighbor list")
            if self.cut is not None and o.cut is not None and self.cut != o.cut:
                raise ValueError("cannot multiply rinv polynomials with different masks in one kernel")
            t = {}
            for p1, c1 in self.terms.items():
                for p2, c2 in o.terms.items():
                    t[p1 + p2] = t.get(p1 + p2, 0.0) + c1 * c2
            return RinvPoly(self.nlist, t, cut=self.cut if self.cut is not None else o.cut)
        if isinstance(o, (PairExpr, SafeNorm, torch.Tensor)):   # (a tensor: a live weight or per-pair values -> the torch route)
            return PairExpr.of(self) * o
        return NotImplemented

    __rmul__ = __mul__

    def __truediv__(self, o):
        if isinstance(o, RowExpr) or (self.reduced and not isinstance(o, (int, float))):
            return RowExpr.of(self)._with("div", o)
        if isinstance(o, (PairExpr, SafeNorm, RinvPoly, torch.Tensor)):
            return PairExpr.of(self) / o
        return self * (1.0 / o)

    def __rtruediv__(self, o):
        if self.reduced:
            return RowExpr.of(self)._with("div", o, swap=True)
        return PairExpr.of(self).__rtruediv__(o)

    def __rsub__(self, o):
        if self.reduced:
            return RowExpr.of(self)._with("sub", o, swap=True)
        return PairExpr.of(self).__rsub__(o)

    def __lt__(self, o): return PairExpr.of(self) < o
    def __le__(self, o): return PairExpr.of(self) <= o
    def __gt__(self, o): return PairExpr.of(self) > o
    def __ge__(self, o): return PairExpr.of(self) >= o

    def __pow__(self, n):
        if self.reduced:
            return self if n == 1 else RowExpr.of(self) ** n
        if int(n) != n or n < 1:
            return PairExpr.of(self) ** n
        out = self
        for _ in range(int(n) - 1):
            out = out * self
        return out

    def key(self):
        return ("poly", tuple(sorted(self.terms.items())), self.cut)

    def potential(self):
        t = dict(self.terms)
        if self.cut is None and len(t) == 2 and t.get(12) == 2.0 and t.get(6) == -2.0:
            return ops.Potential.lj()  # LJModel: 4/2 (rinv^12 - rinv^6), build_examples.py:67-77
        if any(p < 1 for p in t):
            raise ValueError("rinv powers must be >= 1")
        powers = sorted(t)
        return ops.Potential.rinv_poly([t[p] for p in powers], powers, cut=self.cut or 0.0)

    def tensor(self):
        """Eager value [N, NN] (or [N] once reduced) for model outputs other than forces."""
        _trace_log().append({"op": "eager_value"})
        s = ops.nlist_rinv(self.nlist.tensor)
        out = torch.zeros_like(s)
        for p, c in self.terms.items():
            out += c * s ** p
        if self.cut is not None:
            out = out * PairMask(self.nlist, self.cut).tensor().to(out.dtype)
        return out.sum(dim=1) if self.reduced else out


class PairExpr(PairEnergy):
    """Any ELEMENTWISE expression of nlist_rinv (s), safe_norm (r) and masks on the plain norm, kept symbolic (round 5:
    hoomd_tf_amd/codegen.py).  What the zoo above cannot express -- Morse, Yukawa, a switched or tabulated-by-formula potential,
    whatever a notebook writes -- becomes the body of a generated kernel instead of twenty eager torch ops per step:
    ``compute_nlist_forces`` lowers it to an ``HTF_POT_JIT`` potential (compiled once per expression, cached on disk), and
    tfcompute replays it as the one-kernel step like any built-in closed form.  An expression whose energy does not vanish on a
    padded slot keeps the torch route (codegen.vanishes_on_padding)."""
    owns_potential = False
    trains_on_kernels = True   # (a generated unit with weights carries its training sweep; a row-function unit does not)

    def __init__(self, nlist, node, reduced=False, positions=None, folded=()):
        # positions: the [N, 4] tensor of compute(), kept when the expression reads the row particles' own types (TypeExpr)
        # folded: ((weight tensor, its _version when its value became a constant of the expression), ...) -- see _with
        self.nlist, self.node, self.reduced, self.positions, self.folded = nlist, node, reduced, positions, tuple(folded)

    # ---- construction from the other symbolic types
    @staticmethod
    def of(x, nlist=None):
        from . import codegen as cg
        if isinstance(x, PairExpr):
            return x
        if isinstance(x, RinvPoly):
            node = None
            for p_, c_ in sorted(x.terms.items()):
                term = cg.Node("mul", (cg.const(c_), cg.Node("pow", (cg.S,), value=float(p_)))) if p_ != 1 else cg.Node("mul", (cg.const(c_), cg.S))
                node = term if node is None else cg.Node("add", (node, term))
            node = node or cg.const(0.0)
            if x.cut is not None:
                node = cg.Node("mul", (cg.Node("mask", (cg.Node("lt", (cg.RN, cg.const(x.cut))),)), node))
            return PairExpr(x.nlist, node, x.reduced)
        if isinstance(x, SafeNorm):
            if x.delta != 1e-7:
                raise TypeError("only safe_norm's default delta can be traced")
            return PairExpr(x.nlist, cg.R)
        if isinstance(x, PairNorm):
            return PairExpr(x.xyz.parent, cg.RN)
        if isinstance(x, PairMask):
            return PairExpr(x.nlist, cg.Node("mask", (cg.Node("lt", (cg.RN, cg.const(x.cut))),)))
        if isinstance(x, TypeExpr):
            nl = x.nlist if x.nlist is not None else nlist
            if nl is None:
                raise TypeError("an expression of positions[:, 3] alone is not a pair expression")
            if x.nlist is None and not x.expanded:
                raise ValueError("positions[:, 3] is [N]: write positions[:, 3][:, None] to combine it with per-neighbor values")
            return PairExpr(nl, x.node, positions=x.positions)
        if isinstance(x, PairCond):
            raise TypeError("a comparison is not a value: htf.cast it, or use it in htf.where")
        wn = _weight_node(x, nlist) if nlist is not None else None
        if wn is not None:
            return PairExpr(nlist, wn)
        if nlist is not None and isinstance(x, torch.Tensor) and (x.requires_grad or x.numel() != 1):
            fw = _fold_weight(x)
            if fw is not None:
                return PairExpr(nlist, cg.const(fw[0]), folded=fw[1])   # (an inference-time weight: see _with)
            raise TypeError("a trainable weight (while training) or a tensor of per-pair values cannot be an operand of a traced "
                            "htf.where / minimum / maximum: combine it with arithmetic, or write this part in torch ops")
        if nlist is not None:
            return PairExpr(nlist, cg.wrap(x))
        raise TypeError("cannot trace %r" % (type(x),))

    def _with(self, op, other=None, swap=False, value=None):
        from . import codegen as cg
        if self.reduced:
            # a per-particle sum fed into further arithmetic: a ROW expression (an embedding term F(sum_j g(r_ij)), a
            # coordination-number restraint) -- symbolic still, see RowExpr
            return RowExpr.of(self)._with(op, other, swap, value)
        if other is None:
            return PairExpr(self.nlist, cg.Node(op, (self.node,), value=value), positions=self.positions, folded=self.folded)
        new_folded = ()
        wn = _weight_node(other, self.nlist)
        if wn is not None:
            # an element of a trainable leaf, or scalar arithmetic on such: weights of the kernel (p.theta[k]), in inference and in
            # training alike
            other = PairExpr(self.nlist, wn)
        fw = _fold_weight(other)
        if fw is not None:
            # a one-element WEIGHT (or a scalar computed from weights) while nothing is being trained -- inference MD with a model
            # that owns Parameters, as every Keras layer upstream does: its current value becomes a constant of the kernel, and
            # (leaf tensor, _version) pairs are kept so that tfcompute re-traces the model as soon as a weight is written again
            # (load_weights, an optimizer step)
            other, new_folded = fw
        if isinstance(other, torch.Tensor) and (other.requires_grad or other.numel() != 1):
            # a trainable weight, or a tensor of per-particle / per-pair values: not a constant of a generated kernel -- the
            # expression becomes its torch value here and the model takes the autograd route (folding a weight into the kernel
            # would freeze it at its traced value)
            fn = {"add": torch.add, "sub": torch.sub, "mul": torch.mul, "div": torch.div, "min": torch.minimum, "max": torch.maximum,
                  "lt": torch.lt, "le": torch.le, "gt": torch.gt, "ge": torch.ge, "eq": torch.eq, "ne": torch.ne}[op]
            return fn(other, self.ad) if swap else fn(self.ad, other)
        o = PairExpr.of(other, self.nlist)
        if o.nlist is not self.nlist and o.nlist.tensor is not self.nlist.tensor:
            raise ValueError("expressions come from different neighbor lists")
        args = (o.node, self.node) if swap else (self.node, o.node)
        return PairExpr(self.nlist, cg.Node(op, args), positions=self.positions if self.positions is not None else o.positions,
                        folded=self.folded + o.folded + new_folded)

    def __add__(self, o):
        if isinstance(o, BiasTerm):
            return NotImplemented
        return self._with("add", o)
    def __radd__(self, o): return self._with("add", o, swap=True)
    def __sub__(self, o): return self._with("sub", o)
    def __rsub__(self, o): return self._with("sub", o, swap=True)
    def __mul__(self, o): return self._with("mul", o)
    def __rmul__(self, o): return self._with("mul", o, swap=True)
    def __truediv__(self, o): return self._with("div", o)
    def __rtruediv__(self, o): return self._with("div", o, swap=True)
    def __neg__(self): return self._with("neg")
    def __abs__(self): return self._with("abs")

    def __pow__(self, n):
        if not isinstance(n, (int, float)):
            raise TypeError("only constant exponents can be traced")
        return self._with("pow", value=float(n))

    def _cmp(self, op, o):
        e = PairExpr.of(self, self.nlist)._with(op, o)
        if isinstance(e, torch.Tensor):   # (compared with a tensor: torch from here on)
            return e
        return PairCond(self.nlist, e.node, e.positions, e.folded)
    def __lt__(self, o): return self._cmp("lt", o)
    def __le__(self, o): return self._cmp("le", o)
    def __gt__(self, o): return self._cmp("gt", o)
    def __ge__(self, o): return self._cmp("ge", o)

    # ---- lowering
    def body(self):
        from . import codegen as cg
        if getattr(self, "_body", None) is None:
            self._body = self._unit()["text"]     # (forward body; + the training jets when the expression reads weights)
        return self._body

    def _unit(self):
        from . import codegen as cg
        return cg.unit_of(self.node)

    def _weight_indices(self):
        from . import codegen as cg
        return cg.params_of(self.node)

    @property
    def weight_elements(self):
        """[(leaf, flat index)] the expression's p.theta[0 .. n) stand for (the trace's list, up to the highest index it reads)."""
        ks = self._weight_indices()
        return list(self.nlist._weights[:ks[-1] + 1]) if ks else []

    @property
    def layer(self):
        """What the training step takes for a trainable layer (TracedWeights), or None for a weight-free energy."""
        el = self.weight_elements
        if not el:
            return None
        if getattr(self, "_tw", None) is None:
            self._tw = TracedWeights.of(self.body(), el, self.reads_own_type)
        return self._tw

    def lowers(self):
        """Can this expression run as a generated kernel?  It must vanish on a padded slot (codegen.vanishes_on_padding), the ROCm
        compiler must be there to build the unit (a cached code object needs none), and HTF_NO_JIT must not say otherwise;
        else the torch route (forces by autograd)."""
        from . import codegen as cg
        if getattr(self, "_lowers", None) is None:
            unit = self._unit()
            ok = os.environ.get("HTF_NO_JIT") != "1" and unit["vanishes"]
            if ok and getattr(_trace, "training_graph", False) and self._weight_indices() and (self.reads_own_type or not self.trains_on_kernels):
                ok = False   # (the training sweep has no positions tensor beside the pair vectors: the torch route trains it)
            if ok and unit["built"]:
                self._lowers = True   # (this process has built or loaded it before: a model traced at every step asks every step)
                return True
            if ok and not cg.available(self.body()):
                import warnings
                warnings.warn("hoomd_tf_amd: hipcc not found (set HIPCC): the traced pair energy runs as torch ops + autograd "
                              "instead of a generated kernel")
                ok = False
            if ok:
                try:
                    cg.compile_body(self.body())   # (cached: potential() finds the code object)
                except RuntimeError as e:
                    # the generated unit does not compile (an emitter bug, a broken toolchain): the model still runs, slower
                    import warnings
                    warnings.warn("hoomd_tf_amd: the generated kernel of a traced pair energy did not compile -- torch ops + autograd "
                                  "instead (%s)" % str(e).splitlines()[0])
                    ok = False
            unit["built"] = bool(ok)
            self._lowers = ok
        return self._lowers

    def key(self):
        return ("jit", self.body())

    @property
    def reads_own_type(self):
        from . import codegen as cg
        return cg.reads(self.node, "ti")

    def potential(self):
        tw = self.layer
        if tw is not None:
            return tw.potential(self.nlist.tensor.device)
        return ops.Potential.jit(self.body(), reads_own_type=self.reads_own_type)

    def _param_tensors(self):
        return [t.reshape(-1)[i] for t, i in self.weight_elements]

    def torch_value(self, nl_tensor):
        """[N, NN] value from a pair-vector tensor (an autograd leaf for the generic route), in its dtype."""
        from . import codegen as cg
        x = nl_tensor[:, :, :3]
        t = x + 1e-7
        r = torch.sqrt((t * t).sum(dim=2))
        ok = r > 3e-6
        s = torch.where(ok, 1.0 / (torch.where(ok, r, torch.ones_like(r)) + 3e-6), torch.zeros_like(r))
        rn = torch.sqrt((x * x).sum(dim=2)).detach()
        ti = self.positions[:nl_tensor.shape[0], 3][:, None].to(nl_tensor.dtype) if self.positions is not None else None
        return cg.evaluate(self.node, s, r, rn, tj=nl_tensor[:, :, 3].detach(), ti=ti, params=self._param_tensors() or None)

    def tensor(self):
        _trace_log().append({"op": "eager_value"})
        v = self.torch_value(self.nlist.tensor)
        return v.sum(dim=1) if self.reduced else v

    @property
    def ad(self):
        v = self.torch_value(self.nlist.ad)
        return v.sum(dim=1) if self.reduced else v


class RowFnEnergy(PairExpr):
    """energy_i = F(sum_j g(r_ij)): a traced pair expression g and a ROW FUNCTION F of its per-particle sum (``row``: an expression of
    codegen.rowsum(0)).  One generated unit: the kernels accumulate row i's (2 dg/dx, g) as for a pair energy and finish the row
    with forces x F'(rho_i), energy = F(rho_i) (csrc/pair_math.h row_function) -- the one-kernel step, its virial form and the
    streaming evaluator alike.  Trains on the torch route (the unit carries no training sweep)."""
    trains_on_kernels = False

    def __init__(self, nlist, node, row, positions=None, folded=()):
        super().__init__(nlist, node, reduced=True, positions=positions, folded=folded)
        self.row = row

    def _unit(self):
        from . import codegen as cg
        return cg.unit_of(self.node, self.row)

    def _weight_indices(self):
        from . import codegen as cg
        return sorted(set(cg.params_of(self.node)) | set(cg.params_of(self.row)))


class RowExpr(_TorchOperand):
    """A per-particle expression of ROW SUMS of traced pair expressions (round 6): what ``htf.reduce_sum(pair_expr, axis=1)`` becomes
    when model code keeps computing with it -- ``-A * htf.sqrt(rho)`` of an embedded-atom term, ``k * (n - n0) ** 2`` of a
    coordination-number restraint -- the reference's "any graph of the neighbor tensor" (simmodel.py:87-121) one step past
    elementwise.  ``node`` is an expression over codegen.rowsum(k) leaves, ``sums[k]`` the (unreduced) pair expression of sum k.
    ``compute_nlist_forces`` lowers an energy that is a SUM of functions of one row sum each, sum_k F_k(rho_k) (+ plain pair sums),
    to one generated unit per term (RowFnEnergy); anything else -- a product of two different sums, a tensor operand -- is its torch
    value and takes the autograd route, as does training."""

    def __init__(self, nlist, node, sums, positions=None, folded=()):
        self.nlist, self.node, self.sums, self.positions, self.folded = nlist, node, tuple(sums), positions, tuple(folded)

    @staticmethod
    def of(x, nlist=None):
        from . import codegen as cg
        if isinstance(x, RowExpr):
            return x
        scale = 1.0
        if isinstance(x, RinvPoly) and x.reduced and not getattr(x, "total", False):
            # c * sum and sum are the SAME row sum (u + 0.02 * u * u must be a function of one sum): the polynomial travels
            # normalised to its lowest power's coefficient, the factor stays outside
            if x.terms:
                scale = float(x.terms[min(x.terms)])
                x = RinvPoly(x.nlist, {p_: c_ / scale for p_, c_ in x.terms.items()}, reduced=True, cut=x.cut)
            x = PairExpr.of(x)
        if isinstance(x, PairExpr) and x.reduced and not getattr(x, "total", False) and not isinstance(x, RowFnEnergy):
            node = x.node
            while node.op == "mul" and any(a_.op == "const" for a_ in node.args):    # (constant factors of a traced sum likewise)
                c_, rest = (node.args[0], node.args[1]) if node.args[0].op == "const" else (node.args[1], node.args[0])
                scale *= float(c_.value)
                node = rest
            out = RowExpr(x.nlist, cg.rowsum(0), [PairExpr(x.nlist, node, positions=x.positions, folded=x.folded)], x.positions, x.folded)
            return out if scale == 1.0 else out._with("mul", scale, swap=True)
        if nlist is not None and isinstance(x, (int, float, np.floating, np.integer)) and not isinstance(x, bool):
            return RowExpr(nlist, cg.const(x), [])
        if nlist is not None and isinstance(x, torch.Tensor) and x.numel() == 1:
            wn = _weight_node(x, nlist)
            if wn is not None:
                return RowExpr(nlist, wn, [])
            fw = _fold_weight(x)
            if fw is not None:
                return RowExpr(nlist, cg.const(fw[0]), [], folded=fw[1])
            if not x.requires_grad:
                return RowExpr(nlist, cg.const(float(x)), [])
        raise TypeError("cannot trace %r as a per-particle expression" % (type(x),))

    _TORCH = {"add": torch.add, "sub": torch.sub, "mul": torch.mul, "div": torch.div}
    _BINARY = {"mul": "mul", "__mul__": "mul", "multiply": "mul", "add": "add", "__add__": "add", "sub": "sub", "__sub__": "sub",
               "subtract": "sub", "div": "div", "__truediv__": "div", "true_divide": "div", "divide": "div"}

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        # ``weight * row_expr`` reaches torch first (Tensor.__mul__): arithmetic with a one-element tensor -- a weight, a scalar
        # computed from weights -- stays symbolic; everything else sees the expression's torch value, like the other operands
        op = cls._BINARY.get(getattr(func, "__name__", ""))
        if op is not None and len(args) == 2 and not kwargs:
            a, b = args
            if isinstance(a, RowExpr):
                return a._with(op, b)
            if isinstance(b, RowExpr):
                return b._with(op, a, swap=True)
        return func(*_unwrap(args), **{k: _unwrap(v) for k, v in (kwargs or {}).items()})

    def _with(self, op, other=None, swap=False, value=None):
        from . import codegen as cg
        if other is None:
            return RowExpr(self.nlist, cg.Node(op, (self.node,), value=value), self.sums, self.positions, self.folded)
        try:
            o = RowExpr.of(other, self.nlist)
        except TypeError:
            o = None
        if o is None or (o.sums and o.nlist is not self.nlist and o.nlist.tensor is not self.nlist.tensor) or op not in self._TORCH:
            a, b = self.ad, _unwrap(other)   # (a tensor of per-particle values, a comparison, another list: torch from here on)
            fn = self._TORCH.get(op) or {"min": torch.minimum, "max": torch.maximum, "lt": torch.lt, "le": torch.le, "gt": torch.gt,
                                         "ge": torch.ge, "eq": torch.eq, "ne": torch.ne}[op]
            b = torch.as_tensor(b, dtype=a.dtype, device=a.device) if not isinstance(b, torch.Tensor) else b
            return fn(b, a) if swap else fn(a, b)
        sums, mapping = list(self.sums), {}
        for k, e in enumerate(o.sums):
            for j, mine in enumerate(sums):
                if mine.node.key() == e.node.key():
                    mapping[k] = j
                    break
            else:
                mapping[k] = len(sums)
                sums.append(e)
        onode = cg.remap_rowsums(o.node, mapping) if mapping else o.node
        args = (onode, self.node) if swap else (self.node, onode)
        return RowExpr(self.nlist, cg.Node(op, args), sums, self.positions if self.positions is not None else o.positions,
                       self.folded + o.folded)

    def __add__(self, o):
        if isinstance(o, BiasTerm):
            return NotImplemented
        return self._with("add", o)
    def __radd__(self, o): return self._with("add", o, swap=True)
    def __sub__(self, o): return self._with("sub", o)
    def __rsub__(self, o): return self._with("sub", o, swap=True)
    def __mul__(self, o): return self._with("mul", o)
    def __rmul__(self, o): return self._with("mul", o, swap=True)
    def __truediv__(self, o): return self._with("div", o)
    def __rtruediv__(self, o): return self._with("div", o, swap=True)
    def __neg__(self): return self._with("neg")
    def __abs__(self): return self._with("abs")
    def __lt__(self, o): return self._with("lt", o)
    def __le__(self, o): return self._with("le", o)
    def __gt__(self, o): return self._with("gt", o)
    def __ge__(self, o): return self._with("ge", o)

    def __pow__(self, n):
        if not isinstance(n, (int, float)):
            return self.ad ** _unwrap(n)
        return self._with("pow", value=float(n))

    # ---- values
    def torch_value(self, nl_tensor):
        """[N] value from a pair-vector tensor (an autograd leaf for the generic route), in its dtype."""
        from . import codegen as cg
        rows = [e.torch_value(nl_tensor).sum(dim=1) for e in self.sums]
        like = rows[0] if rows else torch.zeros(nl_tensor.shape[0], dtype=nl_tensor.dtype, device=nl_tensor.device)
        ks = sorted(set(cg.params_of(self.node)))
        params = [t.reshape(-1)[i] for t, i in self.nlist._weights[:ks[-1] + 1]] if ks else None
        out = cg.evaluate(self.node, like, None, None, params=params, rows=rows)
        return out if out.dim() else out.expand(nl_tensor.shape[0])

    def tensor(self):
        _trace_log().append({"op": "eager_value"})
        return self.torch_value(self.nlist.tensor)

    @property
    def ad(self):
        return self.torch_value(self.nlist.ad)

    # ---- lowering
    def groups(self):
        """The energy as [RowFnEnergy, ...] -- one per row sum, each a function of that sum alone -- or None when some term mixes
        two different sums (or none carries a sum at all)."""
        from . import codegen as cg
        terms = []

        def flatten(n, sign):
            if n.op == "add":
                flatten(n.args[0], sign)
                flatten(n.args[1], sign)
            elif n.op == "sub":
                flatten(n.args[0], sign)
                flatten(n.args[1], -sign)
            elif n.op == "neg":
                flatten(n.args[0], -sign)
            else:
                terms.append((sign, n))
        flatten(self.node, 1)
        by_sum, consts = {}, []
        for sign, n in terms:
            ks = cg.rowsums_of(n)
            if len(ks) > 1:
                return None
            (by_sum.setdefault(ks[0], []) if ks else consts).append((sign, n))
        if not by_sum:
            return None
        first = min(by_sum)
        by_sum[first] += consts          # (a constant per-particle energy: no force; it rides on the first term)
        out = []
        for k in sorted(by_sum):
            f = None
            for sign, n in by_sum[k]:
                n = n if sign > 0 else cg.Node("neg", (n,))
                f = n if f is None else cg.Node("add", (f, n))
            e = self.sums[k]
            out.append(RowFnEnergy(self.nlist, e.node, cg.remap_rowsums(f, {k: 0}), positions=e.positions if e.positions is not None else self.positions,
                                   folded=self.folded + e.folded))
        return out


class PairCond(_TorchOperand):
    """A comparison of traced expressions: the condition of ``htf.where`` / the argument of ``htf.cast`` (no gradient).  Symbolic
    for those two; to torch code it is the bool tensor it stands for, as :class:`PairMask` is -- the reference's own per-type
    masking is ``tf.equal(nlist[:, :, 3], type_j)`` fed to ordinary ops (simmodel.py:661-693 masked_nlist), so
    ``torch.where(nlist[:, :, 3] == k, a, b)``, ``t[nlist[:, :, 3] == k]``, ``(nlist[:, :, 3] == k) & mask`` and ``.float()``
    must keep working on the eager route."""

    def __init__(self, nlist, node, positions=None, folded=()):
        self.nlist, self.node, self.positions, self.folded = nlist, node, positions, tuple(folded)
        if nlist is None:
            raise TypeError("a comparison of positions[:, 3] alone is not a per-pair condition: bring the neighbor types in, or "
                            "use torch on the tensor")

    __hash__ = object.__hash__

    def tensor(self):
        """The [N, NN] bool tensor: the node evaluated on the pair-vector buffer (codegen.evaluate, the arithmetic a generated
        kernel would do per slot)."""
        v = PairExpr(self.nlist, self.node, positions=self.positions).torch_value(self.nlist.tensor)
        return v if v.dtype == torch.bool else v != 0

    @property
    def ad(self):
        return self.tensor()

    shape = property(lambda self: self.nlist.tensor.shape[:2])
    dtype = property(lambda self: torch.bool)
    device = property(lambda self: self.nlist.tensor.device)

    def _other(self, o):
        o = _unwrap(o)
        return o if isinstance(o, torch.Tensor) else torch.as_tensor(o, device=self.nlist.tensor.device)

    def __and__(self, o): return self.tensor() & self._other(o)
    def __or__(self, o): return self.tensor() | self._other(o)
    def __xor__(self, o): return self.tensor() ^ self._other(o)
    __rand__, __ror__, __rxor__ = __and__, __or__, __xor__
    def __invert__(self): return ~self.tensor()
    def __getitem__(self, idx): return self.tensor()[idx]
    def __mul__(self, o): return self.tensor().to(torch.float32) * _unwrap(o)
    __rmul__ = __mul__
    def to(self, *a, **k): return self.tensor().to(*a, **k)
    def float(self): return self.tensor().float()
    def double(self): return self.tensor().double()
    def long(self): return self.tensor().long()
    def int(self): return self.tensor().int()
    def bool(self): return self.tensor()
    def any(self, *a, **k): return self.tensor().any(*a, **k)
    def all(self, *a, **k): return self.tensor().all(*a, **k)
    def sum(self, *a, **k): return self.tensor().sum(*a, **k)
    def nonzero(self, *a, **k): return self.tensor().nonzero(*a, **k)
    def numpy(self): return self.tensor().cpu().numpy()


def _sym(x):
    return isinstance(x, (PairExpr, RinvPoly, SafeNorm))


def _weight_leaves(t):
    """The leaf tensors (weights) a one-element tensor that requires grad was computed from: ``t`` itself when it is a leaf,
    otherwise the variables of the AccumulateGrad nodes of its autograd graph.  None when the graph is too large to be a scalar
    combination of weights (then the value is not folded)."""
    if t.is_leaf:
        return (t,)
    out, seen, alive, todo = [], set(), [], [t.grad_fn]
    while todo:
        fn = todo.pop()
        if fn is None or id(fn) in seen:
            continue
        seen.add(id(fn))
        alive.append(fn)   # (the Python wrappers of graph nodes are made on demand: kept alive so that their ids stay theirs)
        if len(seen) > 256:
            return None
        v = getattr(fn, "variable", None)
        if v is not None:
            out.append(v)
        todo.extend(f for f, _ in fn.next_functions)
    return tuple(out)


MAX_TRACED_WEIGHTS = 8     # (htf_optimizer_step's small-vector form; a traced energy with more weights takes the torch route)


def _weight_element(x):
    """(leaf, flat index) when ``x`` is ONE ELEMENT of a trainable leaf tensor -- a one-element Parameter itself, or ``w[k]`` of a
    weight vector as upstream's layers write it (LJLayer: ``self.w[1] ** 6``, build_examples.py:336-360) -- else None.  Such an
    operand of a traced pair energy becomes weight k of the generated kernel (``p.theta[k]``: an ARGUMENT, not a constant of the
    text), so the energy keeps its kernel while the weight changes and can be TRAINED on the fast path (round 6)."""
    if not (isinstance(x, torch.Tensor) and x.numel() == 1 and x.requires_grad and x.dtype.is_floating_point):
        return None
    if x.is_leaf:
        return (x, 0)
    fn = x.grad_fn
    if fn is not None and fn.name() == "SelectBackward0" and len(fn.next_functions) == 1:
        src = fn.next_functions[0][0]
        leaf = getattr(src, "variable", None)
        if leaf is not None and leaf.dim() == 1:
            i = int(fn._saved_index)
            return (leaf, i if i >= 0 else i + leaf.numel())
    return None


def _weight_node(x, nlist):
    """A traced Node for a ONE-ELEMENT tensor computed from trainable leaves by scalar arithmetic -- ``w[0]``, ``2.0 * w[0]``,
    ``w[1] ** 6``, ``-a / b`` -- or None (then the value is folded / the torch route taken, as before).  The autograd graph is
    walked, not the tensor: MulBackward / DivBackward save exactly the constant operand the other side's gradient needs,
    PowBackward its exponent; an Add / Sub with a hidden constant operand cannot be recovered and declines."""
    from . import codegen as cg
    if not (isinstance(x, torch.Tensor) and x.numel() == 1 and x.requires_grad and x.dtype.is_floating_point):
        return None
    budget = [32]

    def leaf_param(leaf, flat):
        if not leaf.dtype.is_floating_point:
            return None
        known = any(t is leaf and i == flat for t, i in nlist._weights)
        if not known and len(nlist._weights) >= MAX_TRACED_WEIGHTS:
            return None
        return cg.param(nlist.weight_index(leaf, flat))

    def const_of(t):
        return None if t is None or t.numel() != 1 else cg.const(float(t))

    def walk(fn):
        budget[0] -= 1
        if fn is None or budget[0] < 0:
            return None
        name = fn.name()
        if name == "torch::autograd::AccumulateGrad":
            v = fn.variable
            return leaf_param(v, 0) if v.numel() == 1 else None
        if name == "SelectBackward0":
            src = fn.next_functions[0][0]
            v = getattr(src, "variable", None)
            if v is None or v.dim() != 1:
                return None
            i = int(fn._saved_index)
            return leaf_param(v, i if i >= 0 else i + v.numel())
        kids = [k for k, _ in fn.next_functions]
        if name in ("MulBackward0", "DivBackward0") and len(kids) == 2:
            a = walk(kids[0]) if kids[0] is not None else const_of(fn._saved_self)
            b = walk(kids[1]) if kids[1] is not None else const_of(fn._saved_other)
            return None if a is None or b is None else cg.Node("mul" if name == "MulBackward0" else "div", (a, b))
        if name in ("AddBackward0", "SubBackward0") and len(kids) == 2 and kids[0] is not None and kids[1] is not None and float(fn._saved_alpha) == 1.0:
            a, b = walk(kids[0]), walk(kids[1])
            return None if a is None or b is None else cg.Node("add" if name == "AddBackward0" else "sub", (a, b))
        if name == "NegBackward0":
            a = walk(kids[0])
            return None if a is None else cg.Node("neg", (a,))
        if name == "PowBackward0" and len(kids) == 1:
            a = walk(kids[0])
            e = fn._saved_exponent
            return None if a is None or isinstance(e, torch.Tensor) else cg.Node("pow", (a,), value=float(e))
        if name in ("ExpBackward0", "LogBackward0", "SqrtBackward0", "TanhBackward0") and len(kids) == 1:
            a = walk(kids[0])
            return None if a is None else cg.Node({"E": "exp", "L": "log", "S": "sqrt", "T": "tanh"}[name[0]], (a,))
        return None

    n0 = len(nlist._weights)
    node = walk(x.grad_fn) if not x.is_leaf else leaf_param(x, 0)
    if node is None:
        del nlist._weights[n0:]      # (a partial walk registers nothing)
    return node


class TracedWeights:
    """The weights of a traced pair energy as the training step sees a trainable layer (tensorflowcompute._train_on_batch:
    make_trainable / potential / after_update / nonneg_mask / l1_reg): a float32 device vector theta mirroring the (leaf, index)
    elements in trace order.  The generated kernels read it at launch; the device optimizer updates it and ``after_update`` writes
    the elements back into the leaves (what ``model.w`` shows the user); a leaf written by anybody else (load_weights, an
    optimizer of the user's own) is copied in again before the next use (``refresh_if_stale``: no recompile)."""
    nonneg_mask = 0
    name = "traced-weights"
    _cache = {}

    def __init__(self, body, elements, reads_own_type):
        self.body, self.elements, self.reads_own_type = body, list(elements), bool(reads_own_type)
        self.l1_reg = (0.0,) * max(len(self.elements), 1)
        self.theta = None
        self._pot = None
        self._seen = None

    @classmethod
    def of(cls, body, elements, reads_own_type):
        key = (body, tuple((id(t), i) for t, i in elements))
        tw = cls._cache.get(key)
        if tw is None or any(a is not b for (a, _), (b, _) in zip(tw.elements, elements)):
            if len(cls._cache) > 32:
                cls._cache.clear()
            tw = cls._cache[key] = cls(body, elements, reads_own_type)
        return tw

    def _values(self, device):
        return torch.stack([t.detach().reshape(-1)[i].to(device=device, dtype=torch.float32) for t, i in self.elements])

    def make_trainable(self, device="cuda"):
        if self.theta is None:
            self.theta = self._values(device).contiguous()
            self._seen = tuple(t._version for t, _ in self.elements)
        return self.theta

    def refresh_if_stale(self):
        if self.theta is not None and self._seen != tuple(t._version for t, _ in self.elements):
            self.theta.copy_(self._values(self.theta.device))
            self._seen = tuple(t._version for t, _ in self.elements)

    def potential(self, device="cuda"):
        self.make_trainable(device)
        self.refresh_if_stale()
        if self._pot is None:
            self._pot = ops.Potential.jit(self.body, reads_own_type=self.reads_own_type, theta=self.theta)
        return self._pot

    def after_update(self):
        """theta (just stepped by the device optimizer) -> the leaves the user holds."""
        if getattr(self, "_scatter", None) is None:
            # per leaf: (leaf, its flat indices, the theta slots that go there) -- one index_copy_ per leaf and step
            by_leaf = {}
            for k, (t, i) in enumerate(self.elements):
                by_leaf.setdefault(id(t), (t, [], []))
                by_leaf[id(t)][1].append(i)
                by_leaf[id(t)][2].append(k)
            self._scatter = [(t, torch.tensor(ii, device=t.device), torch.tensor(kk, device=self.theta.device))
                             for t, ii, kk in by_leaf.values()]
        with torch.no_grad():
            for t, ii, kk in self._scatter:
                t.view(-1).index_copy_(0, ii, self.theta.index_select(0, kk).to(device=t.device, dtype=t.dtype))
        self._seen = tuple(t._version for t, _ in self.elements)

    @property
    def w(self):
        return self.theta


class _TracedWeightsOfTerms:
    """The weight vectors of an energy's several row terms, as one thing tfcompute can refresh before a replayed step."""

    def __init__(self, tws):
        self.tws = list(tws)

    def refresh_if_stale(self):
        for t in self.tws:
            t.refresh_if_stale()


def _fold_weight(x):
    """(value, ((leaf, version), ...)) of a one-element tensor that requires grad, while nothing is being trained; else None."""
    if (isinstance(x, torch.Tensor) and x.numel() == 1 and x.requires_grad and not getattr(_trace, "training_graph", False)):
        leaves = _weight_leaves(x)
        if leaves:
            return float(x.detach()), tuple((l, l._version) for l in leaves)
    return None


def _first_positions(*xs):
    for x in xs:
        if getattr(x, "positions", None) is not None:
            return x.positions
    return None


def gather(params, indices, axis=0):
    """tf.gather for model code.  A CONSTANT 1-D ``params`` looked up by a traced type expression
    (``ti[:, None] * ntypes + tj``) becomes a table of the generated kernel: per-species-pair parameters of a traced energy."""
    from . import codegen as cg
    if isinstance(indices, (TypeExpr, PairExpr)):
        vals = params.detach().cpu().numpy() if isinstance(params, torch.Tensor) else np.asarray(params, dtype=np.float64)
        if vals.ndim == 1 and axis == 0 and 1 <= vals.size <= cg.MAX_TABLE:
            if isinstance(indices, TypeExpr) and indices.nlist is None:
                raise TypeError("gather by positions[:, 3] alone is not a per-pair value: index with the neighbor types too, or use torch")
            e = PairExpr.of(indices)
            return PairExpr(e.nlist, cg.table(vals, e.node), positions=e.positions, folded=e.folded)
    p = torch.as_tensor(_unwrap(params))
    i = torch.as_tensor(_unwrap(indices)).to(torch.int64)
    return p[i] if axis == 0 else torch.index_select(p, axis, i.reshape(-1)).reshape(p.shape[:axis] + i.shape + p.shape[axis + 1:])


def equal(a, b):
    """tf.equal: traced on type / pair expressions."""
    if isinstance(a, (TypeExpr, PairExpr)):
        return a._cmp("eq", b)
    if isinstance(b, (TypeExpr, PairExpr)):
        return b._cmp("eq", a)
    return torch.eq(torch.as_tensor(_unwrap(a)), torch.as_tensor(_unwrap(b)))


def not_equal(a, b):
    """tf.not_equal: traced on type / pair expressions."""
    if isinstance(a, (TypeExpr, PairExpr)):
        return a._cmp("ne", b)
    if isinstance(b, (TypeExpr, PairExpr)):
        return b._cmp("ne", a)
    return torch.ne(torch.as_tensor(_unwrap(a)), torch.as_tensor(_unwrap(b)))


def _unary(op, x, torch_fn):
    if isinstance(x, RowExpr):
        return x._with(op)
    if _sym(x):
        return PairExpr.of(x)._with(op)
    return torch_fn(_unwrap(x))


def exp(x):
    """tf.exp for model code: traced on pair expressions, torch otherwise."""
    return _unary("exp", x, torch.exp)


def log(x):
    return _unary("log", x, torch.log)


def tanh(x):
    return _unary("tanh", x, torch.tanh)


def sqrt(x):
    return _unary("sqrt", x, torch.sqrt)


def square(x):
    return _unary("square", x, torch.square)


def abs(x):  # noqa: A001 (tf.abs)
    return _unary("abs", x, torch.abs)


def erf(x):
    """tf.math.erf (traced on pair expressions)."""
    return _unary("erf", x, torch.erf)


def erfc(x):
    """tf.math.erfc: the real-space part of Ewald / damped-shifted-force electrostatics, erfc(alpha r) / r."""
    return _unary("erfc", x, torch.erfc)


def sigmoid(x):
    return _unary("sigmoid", x, torch.sigmoid)


def softplus(x):
    return _unary("softplus", x, lambda t: torch.nn.functional.softplus(t, threshold=1e30))


def sin(x):
    return _unary("sin", x, torch.sin)


def cos(x):
    return _unary("cos", x, torch.cos)


def pow(x, n):  # noqa: A001 (tf.pow)
    if isinstance(x, RowExpr):
        return x ** n
    if _sym(x):
        return PairExpr.of(x) ** n
    return _unwrap(x) ** n


def minimum(a, b):
    if _sym(a) or _sym(b):
        e = PairExpr.of(a if _sym(a) else b)
        return PairExpr.of(a, e.nlist)._with("min", PairExpr.of(b, e.nlist))
    return torch.minimum(torch.as_tensor(_unwrap(a)), torch.as_tensor(_unwrap(b)))


def maximum(a, b):
    if _sym(a) or _sym(b):
        e = PairExpr.of(a if _sym(a) else b)
        return PairExpr.of(a, e.nlist)._with("max", PairExpr.of(b, e.nlist))
    return torch.maximum(torch.as_tensor(_unwrap(a)), torch.as_tensor(_unwrap(b)))


def where(cond, a, b):
    """tf.where(cond, a, b) for model code: traced when the condition compares traced expressions."""
    from . import codegen as cg
    if isinstance(cond, PairCond):
        try:
            ea, eb = PairExpr.of(a, cond.nlist), PairExpr.of(b, cond.nlist)
        except TypeError:
            if not (isinstance(_unwrap(a), torch.Tensor) or isinstance(_unwrap(b), torch.Tensor)):
                raise
            ea = None      # a branch is a tensor of per-pair values: the condition becomes its bool tensor, torch from here on
        if ea is not None:
            return PairExpr(cond.nlist, cg.Node("where", (cond.node, ea.node, eb.node)), positions=_first_positions(cond, ea, eb),
                            folded=cond.folded + ea.folded + eb.folded)
    return torch.where(_unwrap(cond), torch.as_tensor(_unwrap(a)), torch.as_tensor(_unwrap(b)))


class WCAPair(PairEnergy):
    """WCARepulsion(sigma)(nlist): layers.py:91-98."""

    def __init__(self, nlist, sigma, layer=None):
        self.nlist, self.sigma, self.layer = nlist, float(sigma), layer

    def key(self):
        if self.layer is not None and self.layer.w is not None:
            return ("wca-trainable",)
        return ("wca", self.sigma)

    def potential(self):
        if self.layer is not None and self.layer.w is not None:
            return self.layer.potential()
        return ops.Potential.wca(self.sigma)


class LJParamEnergy(PairEnergy):
    """LJLayer(r): the trainable LJ pair energy of example 06 (per pair; reduce_sum(axis=1)
    gives the per-particle energy)."""

    def __init__(self, nlist, layer, reduced=False):
        self.nlist, self.layer, self.reduced = nlist, layer, reduced

    def key(self):
        return ("ljparam",)

    def potential(self):
        return self.layer.potential()

    def tensor(self):
        _trace_log().append({"op": "eager_value"})
        f = ops.eval_forces(self.layer.potential(), self.nlist.tensor)
        return f[:, 3].clone()


class MLPEnergy(PairEnergy):
    """1/2 sum_j [r > 3e-6] MLP(RBF(safe_norm(x_ij))): the C3 pair potential."""
    reduced = True

    def __init__(self, nlist, layer):
        self.nlist, self.layer = nlist, layer

    def key(self):
        return ("mlp",)

    def potential(self):
        return self.layer.potential()


class SortedRinv:
    """tf.sort(nlist_rinv(nlist), axis=1, direction='DESCENDING'): stays symbolic until sliced."""

    def __init__(self, nlist):
        self.nlist = nlist

    def __getitem__(self, idx):
        full = slice(None)
        if (isinstance(idx, tuple) and len(idx) == 2 and idx[0] == full and isinstance(idx[1], slice)
                and idx[1].start in (None, 0) and idx[1].step in (None, 1) and idx[1].stop is not None):
            return TopRinv(self.nlist, int(idx[1].stop))
        return self.tensor()[idx]

    def tensor(self):
        _trace_log().append({"op": "eager_value"})
        s = ops.nlist_rinv(self.nlist.tensor)
        return ops.topk_desc(s, s.shape[1])[0]


class TopRinv:
    """The K largest 1/r of every row, descending: the input features of example 08's network."""

    def __init__(self, nlist, k):
        self.nlist, self.k = nlist, int(k)

    def tensor(self):
        _trace_log().append({"op": "eager_value"})
        return ops.topk_desc(ops.nlist_rinv(self.nlist.tensor), self.k)[0]


class TopkMLPEnergy(PairEnergy):
    """Dense(1)(Dense(H2)(Dense(H1)(top_n))): a per-particle energy [N, 1] (build_examples.py:199-218)."""
    reduced = True

    def __init__(self, top, layers):
        self.nlist, self.top, self.layers = top.nlist, top, list(layers)
        self.owns_potential = True  # kept on its first Dense layer (see _potential_of)

    def key(self):
        return ("topk-mlp",)

    def potential(self):
        d1, d2, d3 = self.layers
        if d3.units != 1:
            raise ValueError("the last Dense of a per-particle energy has one unit")
        if d1.activation != d2.activation or d3.activation not in (None, "linear"):
            raise ValueError("the fused top-k network takes one activation for both hidden layers and a linear output")
        sig = tuple(id(d) for d in self.layers) + tuple(d.version for d in self.layers)
        holder = self.layers[0]
        if getattr(holder, "_topk_pot", None) is None or holder._topk_pot[0] != sig:
            params = {"W1": d1.kernel, "b1": d1.bias, "W2": d2.kernel, "b2": d2.bias, "W3": d3.kernel, "b3": d3.bias}
            holder._topk_pot = (sig, ops.Potential.topk_mlp(params, activation=d1.activation))
        return holder._topk_pot[1]


class DenseOut:
    """The value of a Dense stack applied to the top-k features, still symbolic."""

    def __init__(self, top, layers):
        self.top, self.layers = top, layers

    def energy(self):
        if len(self.layers) != 3:
            raise ValueError("the fused top-k network is Dense-Dense-Dense(1); got %d layers" % len(self.layers))
        return TopkMLPEnergy(self.top, self.layers)


def sort(values, axis=-1, direction='ASCENDING'):
    """tf.sort for model code: descending sort of nlist_rinv(nlist) along the neighbor axis stays symbolic
    (example 08 keeps only its first K columns); anything else is torch.sort."""
    if (isinstance(values, RinvPoly) and not values.reduced and values.terms == {1: 1.0} and axis in (1, -1)
            and direction == 'DESCENDING'):
        return SortedRinv(values.nlist)
    t = values.tensor() if hasattr(values, "tensor") and callable(values.tensor) else _unwrap(values)
    return torch.sort(t, dim=axis, descending=(direction == 'DESCENDING'))[0]


class PairCV:
    """Scalar collective variable (1/N) sum_i sum_j phi(r_ij) with phi one masked
    RBFExpansion channel (layers.SoftRDFCV): the soft RDF bin of config C4.  Stays symbolic
    so that an EDS-biased energy lowers to one pass over the pair vectors."""

    def __init__(self, nlist, r0, gap):
        self.nlist, self.r0, self.gap = nlist, float(r0), float(gap)
        self.value = None  # device scalar once evaluated

    def potential(self):
        return ops.Potential.gauss(self.r0, self.gap, 1.0)

    def tensor(self):
        if self.value is None:
            _trace_log().append({"op": "eager_value"})
            f = ops.eval_forces(self.potential(), self.nlist.tensor)
            self.value = (f[:, 3].sum() / f.shape[0]).reshape(1)
        return self.value.reshape(())

    def __rmul__(self, o):
        if isinstance(o, DeferredAlpha):
            return BiasTerm(o, self)
        return NotImplemented

    def __mul__(self, o):
        if isinstance(o, DeferredAlpha):
            return BiasTerm(o, self)
        return NotImplemented


class DeferredAlpha:
    """alpha = EDSLayer(cv) for a symbolic cv: the state machine advances when the biased
    energy is lowered (exactly once per model call), on the device."""

    def __init__(self, eds, cv):
        self.eds, self.cv = eds, cv
        self.done = False

    def tensor(self):
        if not self.done:  # alpha requested as an output without a biased force: advance here
            self.eds(self.cv.tensor())
            self.done = True
        return self.eds.state[2].clone()

    def __mul__(self, o):
        if isinstance(o, PairCV):
            return BiasTerm(self, o)
        return NotImplemented

    __rmul__ = __mul__


class BiasTerm:
    """alpha * cv"""

    def __init__(self, alpha, cv):
        self.alpha, self.cv = alpha, cv

    def __radd__(self, base):
        return BiasedEnergy(base, self)

    def __add__(self, base):
        return BiasedEnergy(base, self)


class BiasedEnergy(PairEnergy):
    """E_base,i + alpha * cv  (a scalar bias added to every particle's energy)."""
    reduced = True

    def __init__(self, base, term):
        if not isinstance(base, RinvPoly) and not isinstance(base, WCAPair):
            raise TypeError("the EDS bias adds to a closed-form pair energy")
        if base.nlist is not term.cv.nlist:
            raise ValueError("expressions come from different neighbor lists")
        self.nlist, self.base, self.term = base.nlist, base, term

    def key(self):
        return ("biased",) + self.base.key()


class SafeNorm:
    """safe_norm(nlist[:, :, :3], axis=2) kept symbolic so RBF/MLP layers can fuse."""

    def __init__(self, nlist, delta):
        self.nlist, self.delta = nlist, delta

    def tensor(self):
        t = self.nlist.tensor[:, :, :3] + self.delta
        return torch.sqrt((t * t).sum(dim=2))

    # arithmetic on the distance: a traced expression (round 5; the RBF / MLP layers keep taking the SafeNorm itself)
    def __add__(self, o): return PairExpr.of(self) + o
    def __radd__(self, o): return o + PairExpr.of(self)
    def __sub__(self, o): return PairExpr.of(self) - o
    def __rsub__(self, o): return o - PairExpr.of(self)
    def __mul__(self, o): return PairExpr.of(self) * o
    def __rmul__(self, o): return o * PairExpr.of(self)
    def __truediv__(self, o): return PairExpr.of(self) / o
    def __rtruediv__(self, o): return o / PairExpr.of(self)
    def __neg__(self): return -PairExpr.of(self)
    def __pow__(self, n): return PairExpr.of(self) ** n
    def __lt__(self, o): return PairExpr.of(self) < o
    def __le__(self, o): return PairExpr.of(self) <= o
    def __gt__(self, o): return PairExpr.of(self) > o
    def __ge__(self, o): return PairExpr.of(self) >= o


def reduce_sum(x, axis=None):
    """tf.reduce_sum for expressions (axis=1: per-particle energy) and tensors."""
    if isinstance(x, RinvPoly):
        if axis is None:
            # a scalar total energy: compute_nlist_forces differentiates sum(energy) either way, and
            # _add_energy tiles a rank-0 energy into every particle's column (simmodel.py:558-578)
            out = RinvPoly(x.nlist, x.terms, reduced=True, cut=x.cut)
            out.total = True
            return out
        if axis not in (1, -1):
            raise ValueError("pair energies reduce over the neighbor axis (axis=1), or over everything (axis=None)")
        return RinvPoly(x.nlist, x.terms, reduced=True, cut=x.cut)
    if isinstance(x, PairExpr):
        if axis is None:
            out = PairExpr(x.nlist, x.node, reduced=True, positions=x.positions, folded=x.folded)
            out.total = True
            return out
        if axis not in (1, -1):
            raise ValueError("pair energies reduce over the neighbor axis (axis=1), or over everything (axis=None)")
        return PairExpr(x.nlist, x.node, reduced=True, positions=x.positions, folded=x.folded)
    if isinstance(x, WCAPair):
        return x
    if isinstance(x, LJParamEnergy):
        return LJParamEnergy(x.nlist, x.layer, reduced=True)
    if isinstance(x, RowExpr):
        x = x.ad     # (a total over particles: a torch scalar from here on)
    return x.sum() if axis is None else x.sum(dim=axis)


# --------------------------------------------------------------------------- free functions
def nlist_rinv(nlist):
    """simmodel.py:618-635: 1/r per neighbor, padded slots exactly 0, differentiable."""
    return RinvPoly(_as_nlist(nlist), {1: 1.0})


def safe_norm(tensor, delta=1e-7, axis=None, **kwargs):
    """simmodel.py:581-594: ``tf.norm(tensor + delta)``."""
    if isinstance(tensor, NlistXYZ):
        if axis not in (2, -1):
            raise ValueError("safe_norm of the pair vectors reduces axis=2")
        return SafeNorm(tensor.parent, delta)
    t = tensor + delta
    return torch.sqrt((t * t).sum()) if axis is None else torch.sqrt((t * t).sum(dim=axis))


def compute_nlist_forces(nlist, energy, virial=False):
    """simmodel.py:526-555.  Returns forces [N,4] (fx,fy,fz,energy), or (forces, virial
    [N,3,3]) when ``virial``.  Raises ValueError when energy does not depend on nlist
    ('Did you put them in wrong order?')."""
    if isinstance(energy, torch.Tensor):
        return _autograd_nlist_forces(_as_nlist(nlist), energy, virial)
    if isinstance(energy, DenseOut):
        energy = energy.energy()
    if isinstance(energy, RowExpr):
        return _row_forces(_as_nlist(nlist), energy, virial)
    if isinstance(energy, PairExpr) and not energy.lowers():
        # (its energy does not vanish on a padded slot, or HTF_NO_JIT=1: the generic route, forces by torch.autograd)
        nl_ = _as_nlist(nlist)
        return _autograd_nlist_forces(nl_, energy.torch_value(nl_.ad).sum(dim=1), virial)
    if not isinstance(energy, PairEnergy):
        raise ValueError('Could not find dependence between energy and nlist.'
                         ' Did you put them in wrong order?')
    nl = _as_nlist(nlist)
    if energy.nlist is not nl and energy.nlist.tensor is not nl.tensor:
        raise ValueError('Could not find dependence between energy and nlist.'
                         ' Did you put them in wrong order?')
    if isinstance(energy, BiasedEnergy):
        if virial:
            raise ValueError("virial of an EDS-biased energy is not implemented")
        return _biased_forces(nl, energy)
    pot = _potential_of(energy)
    own = None
    if isinstance(energy, PairExpr) and energy.reads_own_type:
        if energy.positions is None:
            raise ValueError("the traced energy reads the particles' own types but no positions tensor reached it")
        own = energy.positions[:nl.tensor.shape[0]].to(nl.tensor.dtype).contiguous()
    out = ops.eval_forces(pot, nl.tensor, virial=virial, positions=own)
    f = out[0] if virial else out
    if getattr(energy, "total", False):
        f[:, 3] = f[:, 3].sum()  # rank-0 energy: the total in every particle's column
        _trace_log().append({"op": "total_energy"})  # (keeps the step on the eager path)
    _trace_log().append({"potential": pot, "nlist": nl, "virial": virial, "forces": f,
                         "layer": getattr(energy, "layer", None), "folded": tuple(getattr(energy, "folded", ()))})
    return out


def _row_forces(nl, energy, virial):
    """compute_nlist_forces of a RowExpr: one generated unit per term F_k(rho_k) (RowFnEnergy), their outputs added; a single term
    is logged like any lowered energy (tfcompute replays it as the one-kernel step).  Terms that mix sums, training, HTF_NO_JIT or a
    pair expression that does not vanish on padding: the torch value and autograd."""
    if energy.sums and energy.nlist is not nl and energy.nlist.tensor is not nl.tensor:
        raise ValueError('Could not find dependence between energy and nlist.'
                         ' Did you put them in wrong order?')
    groups = energy.groups()
    # (a virial of several terms: simmodel.py:509-523 takes the norm of a pair's TOTAL force, which separate launches cannot form)
    if groups is None or (virial and len(groups) > 1) or not all(g.lowers() for g in groups):
        return _autograd_nlist_forces(nl, energy.torch_value(nl.ad), virial)
    outs, pots = [], []
    for g in groups:
        pot = _potential_of(g)
        own = None
        if g.reads_own_type:
            if g.positions is None:
                raise ValueError("the traced energy reads the particles' own types but no positions tensor reached it")
            own = g.positions[:nl.tensor.shape[0]].to(nl.tensor.dtype).contiguous()
        outs.append(ops.eval_forces(pot, nl.tensor, virial=virial, positions=own))
        pots.append(pot)
    if len(groups) == 1:
        f = outs[0][0] if virial else outs[0]
        _trace_log().append({"potential": pots[0], "nlist": nl, "virial": virial, "forces": f,
                             "layer": None if getattr(_trace, "training_graph", False) else groups[0].layer,
                             "folded": tuple(groups[0].folded)})
        return outs[0]
    # several terms: the first is logged like any lowered energy, the others ride along -- tfcompute's plan is then the one-kernel
    # step of the first term (which writes the tensor) followed by one streaming evaluation per further term, added in place
    f = sum(outs[1:], outs[0])
    typed = [bool(g.reads_own_type) for g in groups]
    # (weights of the terms are kernel arguments of their units: the plan re-uploads whichever were written since the last step)
    tws = [g.layer for g in groups if g.layer is not None]
    _trace_log().append({"potential": pots[0], "nlist": nl, "virial": False, "forces": f,
                         "layer": _TracedWeightsOfTerms(tws) if tws and not getattr(_trace, "training_graph", False) else None,
                         "folded": tuple(x for g in groups for x in g.folded),
                         "extra_potentials": [(p_, t_) for p_, t_ in zip(pots[1:], typed[1:])]})
    return f


def _potential_of(energy):
    """The lowered potential of a symbolic energy.  Energies that read a layer's weights get the
    potential that layer OWNS (the layer keeps it for as long as it lives, so another layer can never be
    handed it); only weight-free closed forms, identified by their VALUES (polynomial terms, a fixed
    sigma), share the process-wide table."""
    if getattr(energy, "owns_potential", False) or (getattr(energy, "layer", None) is not None and energy.key()[0] != "wca"):
        return energy.potential()
    cache = getattr(compute_nlist_forces, "_cache", None)
    if cache is None:
        cache = compute_nlist_forces._cache = {}
    k = energy.key()
    pot = cache.get(k)
    if pot is None:
        if len(cache) > 64:
            cache.clear()
        pot = cache[k] = energy.potential()
    return pot


def _add_energy(forces, energy):
    """simmodel.py:558-578: column 3 <- per-particle energy (rank > 1: summed over the trailing
    axes; rank 0: the scalar tiled to every particle); drops dE/dtype."""
    N = forces.shape[0]
    if energy.dim() > 1:
        e = energy.reshape(N, -1).sum(dim=1, keepdim=True)
    elif energy.dim() == 0:
        e = energy.reshape(1, 1).expand(N, 1)
    else:
        e = energy.reshape(N, 1)
    return torch.cat([forces[:, :3], e.to(forces.dtype)], dim=-1)


def _autograd_nlist_forces(nl, energy, virial):
    """The generic route: ``energy`` is any torch expression of the neighbor
    tensor; forces by torch.autograd exactly as simmodel.py:526-555 does with tf.gradients
    (x2, summed over the neighbor axis).  No fused kernel, no plan: the step stays eager."""
    g = None
    if nl._ad is not None and energy.requires_grad:
        # while a generic model is being trained the forces must stay differentiable w.r.t. its weights
        (g,) = torch.autograd.grad(energy.sum(), nl._ad, allow_unused=True, retain_graph=True,
                                   create_graph=bool(getattr(_trace, "training_graph", False)))
    if g is None:
        raise ValueError('Could not find dependence between energy and nlist.'
                         ' Did you put them in wrong order?')
    nlist_forces = 2.0 * g
    forces = _add_energy(nlist_forces.sum(dim=1), energy if getattr(_trace, "training_graph", False) else energy.detach())
    _trace_log().append({"op": "generic"})
    if not virial:
        return forces
    n3 = nl.tensor[:, :, :3]
    rmag = torch.sqrt((n3 * n3).sum(dim=2))
    fmag = torch.sqrt((nlist_forces * nlist_forces).sum(dim=2))
    den = 2.0 * rmag
    f_rs = torch.where(den == 0, torch.zeros_like(den), fmag / den)  # divide_no_nan
    return forces, -1.0 * torch.einsum("ij,ijk,ijl->ikl", f_rs, n3, n3)


def _biased_forces(nl, energy):
    """E_base + alpha * cv in one sweep: base forces and unit-bias forces from
    htf_eval_forces2, cv from a deterministic block reduction, alpha from the device-side EDS
    state machine, then force = base + alpha * bias -- nothing returns to the host."""
    cache = getattr(compute_nlist_forces, "_cache", None)
    if cache is None:
        cache = compute_nlist_forces._cache = {}
    base, cv, alpha = energy.base, energy.term.cv, energy.term.alpha
    kb, kg = base.key(), ("gauss", cv.r0, cv.gap)
    if kb not in cache or (getattr(base, "layer", None) is not None and kb[0] != "wca"):
        cache[kb] = _potential_of(base)
    if kg not in cache:
        cache[kg] = cv.potential()
    t = nl.tensor
    B, NN = int(t.shape[0]), int(t.shape[1])
    n = ops.num_partials(B, NN)
    partials = torch.empty(n, dtype=torch.float32, device=t.device)
    fa, fb = ops.eval_forces2(cache[kb], cache[kg], t, partials=partials)
    cv.value = torch.empty(1, dtype=torch.float32, device=t.device)
    ops.reduce_partials(partials, n, 1.0 / B, cv.value)
    if not alpha.done:
        mark = len(_trace_log())
        alpha.eds(cv.value)  # htf_eds_update on the device scalar
        del _trace_log()[mark:]  # the layer's own "stateful" entry: covered by the "biased" entry below
        alpha.done = True
    ops.bias_combine(fa, fb, alpha.eds.state[2:3], cv.value)
    # "biased": what tfcompute needs to replay this step as ONE kernel (htf_build_eval_forces2)
    _trace_log().append({"stateful": "eds-bias", "forces": fa, "nlist": nl,
                         "biased": {"pot_a": cache[kb], "pot_b": cache[kg], "cv": cv, "eds": alpha.eds}})
    return fa


def pairwise_unit_forces(nlist):
    """SimplePotential's body (build_examples.py:9-22) as one fused op:
    F_i = -sum_j x_ij / |x_ij| with non-finite terms dropped.  Returns [N, 4] (w = 0)."""
    nl = _as_nlist(nlist)
    cache = getattr(pairwise_unit_forces, "_pot", None)
    if cache is None:
        cache = pairwise_unit_forces._pot = ops.Potential.simple()
    f = ops.eval_forces(cache, nl.tensor)
    _trace_log().append({"potential": cache, "nlist": nl, "virial": False, "forces": f})
    return f


def compute_positions_forces(positions, energy):
    """simmodel.py:492-506: ``-dE/dpositions`` with the energy column appended.  CV-bias
    models are O(few particles); this generic path differentiates with torch.autograd
    (SURVEY 8(f)-3 fallback), not a fused kernel."""
    if isinstance(energy, PosRadial):
        # the declarative positions energy: forces and the energy column from ONE kernel
        # (htf_positions_forces_radial), no autograd graph
        if energy.norm.positions is not positions:
            raise ValueError('Could not find dependence between energy and positions.')
        _trace_log().append({"op": "positions_radial"})
        return ops.positions_forces_radial(positions.tensor, energy.coef, energy.power, energy.norm.ncomp)
    energy = _unwrap(energy)
    if isinstance(positions, Positions):
        positions = positions.ad
    if not (isinstance(energy, torch.Tensor) and energy.requires_grad):
        raise ValueError('Could not find dependence between energy and positions.')
    (g,) = torch.autograd.grad(energy.sum(), positions, allow_unused=False)
    forces = -g
    N = positions.shape[0]
    e = energy.detach()
    if e.dim() == 0:
        col = e.reshape(1).repeat(N)
    elif e.dim() > 1:
        col = e.reshape(N, -1).sum(dim=1)
    else:
        col = e
    return torch.cat([forces[:, :3], col.reshape(N, 1)], dim=-1)


def box_size(box):
    """simmodel.py:597-603."""
    return box[1, :] - box[0, :]


def wrap_vector(r, box):
    """simmodel.py:606-615: minimum image of ``r`` (orthorhombic)."""
    _trace_log().append({"op": "wrap_vector"})
    r2 = r.reshape(-1, 3).contiguous()
    if not r2.is_cuda or r2.requires_grad:
        bs = box_size(box).to(r.dtype)
        return r - torch.round(r / bs) * bs
    out = torch.empty_like(r2)
    b = _lib.make_box(box.detach().cpu().numpy())
    check(lib.htf_wrap_vector(r2.data_ptr(), ops._dt(r2), r2.shape[0], C.byref(b), out.data_ptr(), ops._stream(r2)))
    return out.reshape(r.shape)


def masked_nlist(nlist, type_tensor, type_i=None, type_j=None):
    """simmodel.py:676-693 (eager; pure data movement)."""
    t = nlist.tensor if isinstance(nlist, Nlist) else nlist
    if isinstance(type_tensor, TypeExpr):
        type_tensor = type_tensor.ad   # (``positions[:, 3]``: the view it stands for)
    _trace_log().append({"op": "masked_nlist"})
    if type_i is not None:
        t = t[type_tensor == type_i]
    if type_j is not None:
        t = t * (t[:, :, 3] == type_j).to(t.dtype)[:, :, None]
    return t


def compute_rdf(nlist, r_range, type_tensor=None, nbins=100, type_i=None, type_j=None):
    """simmodel.py:638-673 -> (rdf[nbins], bin midpoints), both fp32 device tensors.
    One fused histogram pass over the pair vectors (type masking included).

    Traced: the untyped form over the step's own neighbor tensor records HOW to redo itself into the same output tensors
    (``replay``), so that a model which feeds it to a device-side metric (``MeanTensor``) can be replayed from the step
    plan like the reference's tf.function does -- no Python in the step loop (tfcompute._maybe_install_plan)."""
    t = nlist.tensor if isinstance(nlist, Nlist) else nlist
    ops._dev(t, "nlist")
    if isinstance(type_tensor, TypeExpr):
        type_tensor = type_tensor.ad   # (``positions[:, 3]``: the view of the step's positions tensor it stands for)
    r0, r1 = float(r_range[0]), float(r_range[1])
    hist = torch.zeros(nbins + 2, dtype=torch.int32, device=t.device)
    tt, stride = None, 0
    if type_tensor is not None:
        tt = type_tensor.to(torch.float32)
        stride = tt.stride(0) if tt.dim() == 1 else 1
        if not tt.is_cuda:
            raise ValueError("type_tensor must live on the device")
    else:
        type_i = type_j = None
    ti, tj = -1 if type_i is None else int(type_i), -1 if type_j is None else int(type_j)
    rdf = torch.empty(nbins, dtype=torch.float32, device=t.device)
    rs = torch.empty(nbins, dtype=torch.float32, device=t.device)

    def fill(src, types, tstride):
        check(lib.htf_rdf_histogram(src.data_ptr(), ops._dt(src), int(src.shape[0]), int(src.shape[1]), r0, r1, nbins + 2,
                                    types.data_ptr() if types is not None else None, int(tstride), ti, tj,
                                    hist.data_ptr(), ops._stream(src)))
        check(lib.htf_rdf_finalize(hist.data_ptr(), nbins, r0, r1, rdf.data_ptr(), rs.data_ptr(), ops._stream(hist)))

    def replay(src, pos=None):
        """The same observable of another step's tensor, into the SAME output tensors.  The typed form reads the row types from
        column 3 of that step's positions side buffer (fp32 [N, 4]: what ``positions[:, 3]`` is a view of in compute())."""
        hist.zero_()
        if tt is None:
            fill(src, None, 0)
        else:
            fill(src, pos[:, 3], pos.stride(0))

    fill(t, tt, stride)
    # an observable: a model whose outputs are saved stays on the eager path; when nothing reads the
    # outputs (save_output_period None) tfcompute may replay the step without it -- or, when it can redo itself, with it.
    # The typed form (simmodel.py:656-658 masked_nlist) can redo itself when its types are ``positions[:, 3]`` of the step's
    # own positions tensor: tfcompute._finish_update checks that (``type_tensor``) and withdraws ``replay`` otherwise.
    _trace_log().append({"op": "compute_rdf", "observable": True, "nlist": nlist, "outputs": (rdf, rs),
                         "type_tensor": type_tensor, "replay": replay})
    return rdf, rs


def rdf_from_histogram(hist, r0, r1):
    """compute_rdf's tail (simmodel.py:663-668) for a histogram of nbins + 2 counts."""
    nbins = int(hist.numel()) - 2
    rdf = torch.empty(nbins, dtype=torch.float32, device=hist.device)
    rs = torch.empty(nbins, dtype=torch.float32, device=hist.device)
    check(lib.htf_rdf_finalize(hist.data_ptr(), nbins, float(r0), float(r1), rdf.data_ptr(), rs.data_ptr(),
                               ops._stream(hist)))
    return rdf, rs


class MeanTensor:
    """tf.keras.metrics.MeanTensor, which examples/01 (Quickstart, cell 3) averages its RDF with: the element-wise running mean
    of a tensor -- or of a tuple of tensors, stacked as ``compute_rdf``'s [rdf, r] is -- kept on the device.

    ``update_state`` inside ``SimModel.compute`` is recorded in the trace: when every value it takes came out of a replayable
    observable, the update is replayable too, and tfcompute runs it from the step plan (a few device launches, capturable in a
    hipGraph) instead of calling ``compute`` again every step."""

    def __init__(self):
        self.total, self._count = None, None

    def _add(self, vals):
        for acc, v in zip(self.total, vals):
            acc.add_(v)
        self._count.add_(1.0)

    def update_state(self, values):
        raw = list(values) if isinstance(values, (tuple, list)) else [values]
        vals = [v.detach() for v in raw]  # (views of the same storage: a replayed observable refills it in place)
        for v in vals:
            ops._dev(v, "MeanTensor value")
        if self.total is None:
            self.total = [torch.zeros_like(v, dtype=torch.float32) for v in vals]
            self._count = torch.zeros((), dtype=torch.float32, device=vals[0].device)
            self._tuple = isinstance(values, (tuple, list))
        if len(vals) != len(self.total) or any(v.shape != a.shape for v, a in zip(vals, self.total)):
            raise ValueError("MeanTensor: the shape of the values changed between updates")
        self._add(vals)
        produced = [o for e in _trace_log() if e.get("replay") is not None for o in e.get("outputs", ())]
        replayable = all(any(v is o for o in produced) for v in raw)
        _trace_log().append({"op": "metric_update", "observable": True, "inputs": raw,
                             "replay": (lambda _src, _pos=None, vals=vals: self._add(vals)) if replayable else None})

    @property
    def count(self):
        return 0 if self._count is None else int(round(float(self._count)))

    def reset_states(self):
        if self.total is not None:
            for acc in self.total:
                acc.zero_()
            self._count.zero_()

    reset_state = reset_states

    def result(self):
        if self.total is None:
            raise ValueError("MeanTensor.result() before the first update_state()")
        n = torch.clamp(self._count, min=1.0)
        return torch.stack(self.total) / n if self._tuple else self.total[0] / n


class _LossMetric:
    """model.metrics[0]: running mean of the training loss (kept on the device by the
    optimizer kernel; read back only when asked)."""
    name = 'loss'

    def __init__(self):
        self.state = None

    def result(self):
        if self.state is None or float(self.state[19]) == 0:
            return torch.tensor(0.0)
        return (self.state[18] / self.state[19]).cpu()


# --------------------------------------------------------------------------- SimModel
class SimModel:
    """simmodel.py:8-145.  Subclass and implement ``compute``; optionally ``setup``."""

    def __init__(self, nneighbor_cutoff, output_forces=True, virial=False, check_nlist=False,
                 dtype=torch.float32, name='htf-model', **kwargs):
        self.nneighbor_cutoff = nneighbor_cutoff
        self.output_forces = output_forces
        self.virial = virial
        self.check_nlist = check_nlist
        self.dtype = dtype
        self.name = name
        self._map_nlist = False
        self.loss = None
        if SimModel.compute == self.__class__.compute:
            raise AttributeError('You must implement compute method in subclass')
        try:
            code = self.compute.__code__
            self._arg_count = code.co_argcount - 1  # - 1 for self
            self._pass_training = 'training' == code.co_varnames[self._arg_count]
            if self._pass_training:
                self._arg_count -= 1
        except AttributeError:
            raise AttributeError('SimModel child class must implement compute method, and should not implement call')
        self.batch_steps = 0
        self.optimizer = None
        self.metrics = []
        self.setup(**kwargs)

    def compile(self, optimizer='rmsprop', loss=None, **kwargs):
        """tf.keras.Model.compile for the training path (tensorflowcompute.py:83-95 reads
        ``model.loss``).  Losses: 'MeanSquaredError' (or a list whose first entry is it and the
        rest None, as example 06 passes)."""
        from . import optimizers
        self.optimizer = optimizers.get(optimizer)
        first = loss[0] if isinstance(loss, (list, tuple)) else loss
        if first not in ('MeanSquaredError', 'mse', 'mean_squared_error'):
            raise ValueError("loss %r is not built; available: 'MeanSquaredError'" % (first,))
        self.loss = list(loss) if isinstance(loss, (list, tuple)) else [loss]
        self.metrics = [_LossMetric()]

    # ---- Keras weights API over whatever the model's layers hold (simmodel.py inherits it from tf.keras.Model)
    def _weight_holders(self):
        return [v for _, v in sorted(vars(self).items()) if hasattr(v, "get_weights") and hasattr(v, "set_weights")]

    def parameters(self):
        """torch tensors with requires_grad among the model's attributes (generic-route models)."""
        out = []
        for _, v in sorted(vars(self).items()):
            if isinstance(v, torch.Tensor) and v.requires_grad:
                out.append(v)
            elif isinstance(v, torch.nn.Module):
                out.extend(p for p in v.parameters() if p.requires_grad)
        return out

    def get_weights(self):
        ws = []
        for h in self._weight_holders():
            ws.extend(h.get_weights())
        ws.extend(p.detach().cpu().numpy().copy() for p in self.parameters())
        return ws

    def set_weights(self, ws):
        ws = list(ws)
        for h in self._weight_holders():
            n = len(h.get_weights())
            h.set_weights(ws[:n])
            ws = ws[n:]
        for p, w in zip(self.parameters(), ws):
            with torch.no_grad():
                p.copy_(torch.as_tensor(w, dtype=p.dtype))
        # an installed plan may have baked a weight into its potential (a non-trainable WCARepulsion's
        # sigma): the next step traces compute() again and lowers it from the new weights
        self.retrace_compute()

    def save_weights(self, path):
        np.savez(path, *self.get_weights())

    def load_weights(self, path):
        with np.load(path if str(path).endswith(".npz") else str(path) + ".npz") as z:
            self.set_weights([z["arr_%d" % i] for i in range(len(z.files))])

    def get_config(self):
        return {'nneighbor_cutoff': self.nneighbor_cutoff, 'output_forces': self.output_forces,
                'virial': self.virial, 'check_nlist': self.check_nlist, 'name': self.name, 'dtype': self.dtype}

    def compute(self, nlist, positions, box, training=True):
        raise AttributeError('You must implement compute in your subclass')

    def setup(self, **kwargs):
        pass

    def retrace_compute(self):
        """simmodel.py:147-163: forget the fused plan so the next step re-traces compute."""
        self._plan = None

    def __call__(self, inputs, training=False):
        args = list(inputs[:self._arg_count])
        out = self.compute(*args, training) if self._pass_training else self.compute(*args)
        if not isinstance(out, (tuple, list)):
            out = (out,)
        return tuple(out)

    def mapped_nlist(self, nlist):
        """simmodel.py:257-271: (all-atom nlist, mapped nlist) after tfcompute.enable_mapped_nlist."""
        if not self._map_nlist:
            raise ValueError('You must call tfcompute.enable_mapped_nlist before using mapped_nlist')
        t = _unwrap(nlist)
        return t[:self._map_i], t[self._map_i:]

    def mapped_positions(self, positions):
        """simmodel.py:273-287."""
        if not self._map_nlist:
            raise ValueError('You must call tfcompute.enable_mapped_nlist before using mapped_nlist')
        return positions[:self._map_i], positions[self._map_i:]

    def precompute(self, system):
        """simmodel.py:289-339: re-map the coarse-grained beads from the all-atom positions and
        write their xyz back into the HOOMD position array (types untouched, htf_copy3) before
        the neighbor search.  ``_map_fxn(positions [AAN,4], [Lx,Ly,Lz]) -> [M,4]`` in torch."""
        if not self._map_nlist:
            return
        n = self._map_i
        pos = ops.copy_positions(system.pos, offset=0, N=n, unstuff4=True).to(self.dtype)
        L = torch.as_tensor(system.box3x3[1] - system.box3x3[0], dtype=self.dtype, device=pos.device)
        cg = self._map_fxn(pos, L).to(system.dtype).contiguous()
        ops.copy3(system.pos[n:], cg)

    def compute_inputs(self, nlist, positions, box):
        """simmodel.py:165-238 minus the copies: box-skew assert, optional check_nlist,
        cast to the model dtype (a no-op view when the wire dtype already matches)."""
        if not float(box[2].sum()) < 0.0001:
            raise SkewedBoxError('box is skewed')
        if self.check_nlist and self.nneighbor_cutoff > 0:
            if not ops.check_nlist(nlist) < self.nneighbor_cutoff:
                raise NlistOverflowError('Neighbor list is full!')
        positions = positions.to(self.dtype)
        if self.nneighbor_cutoff == 0:
            positions = Positions(positions)  # CV-bias models differentiate w.r.t. positions
        else:
            positions = PositionsInput.wrap(positions)   # (a tensor whose [:, 3] keeps its identity for traced energies)
        return [Nlist(nlist.to(self.dtype)), positions, box.to(self.dtype)]

    @staticmethod
    def compute_outputs(forces, out_dtype):
        """simmodel.py:240-255: pad [N,3] -> [N,4] with a zero energy column, cast."""
        if forces.shape[1] == 3:
            forces = torch.cat([forces, torch.zeros_like(forces[:, :1])], dim=1)
        return forces.to(out_dtype)


def _make_reverse_indices(mol_indices):
    """simmodel.py:714-733: atom (0-based) -> [molecule, slot]; mol_indices are 1-based here."""
    num_atoms = 0
    for m in mol_indices:
        num_atoms = max(num_atoms, max(m))
    rmi = [[] for _ in range(num_atoms)]
    for i in range(len(mol_indices)):
        for j in range(len(mol_indices[i])):
            index = mol_indices[i][j]
            if index > 0:
                rmi[index - 1] = [i, j]
    warned = False
    for r in rmi:
        if len(r) != 2 and not warned:
            warned = True
            print('Not all of your atoms are in a molecule\n')
            r.extend([-1, -1])
    return rmi


class MolSimModel(SimModel):
    """simmodel.py:342-489: a molecule-batched SimModel.  ``mol_compute(nlist, positions,
    mol_nlist [M,MN,NN,4], mol_positions [M,MN,4], box)`` -- write it in torch; forces still
    come from ``compute_nlist_forces(nlist, energy)`` (autograd flows through the gather)."""

    def __init__(self, MN, mol_indices, nneighbor_cutoff, output_forces=True, virial=False, check_nlist=False,
                 dtype=torch.float32, name='htf-mol-model', **kwargs):
        super().__init__(nneighbor_cutoff, output_forces=output_forces, virial=virial, check_nlist=check_nlist,
                         dtype=dtype, name=name, **kwargs)
        self.MN = MN
        self.mol_indices = mol_indices
        for mi in self.mol_indices:
            for i in range(len(mi)):
                mi[i] += 1  # index 0 slices the dummy atom
            if len(mi) > MN:
                raise ValueError('One of your molecule indices has more than MN indices.'
                                 'Increase MN in your graph.')
            while len(mi) < MN:
                mi.append(0)
        self.rev_mol_indices = _make_reverse_indices(mol_indices)
        self._mol_flat = None
        if MolSimModel.mol_compute == self.__class__.mol_compute:
            raise AttributeError('You must implement mol_compute method in subclass of MolSimModel')
        self._mol_arg_count = self.mol_compute.__code__.co_argcount - 1
        if self._mol_arg_count < 3:
            raise AttributeError('You are creating a molecular batched model, but are only using per atom '
                                 'nlist/positions. Either use only SimModel or increase your argument count '
                                 'to mol_compute')

    def get_config(self):
        config = super().get_config()
        config.update({'MN': self.MN, 'mol_indices': self.mol_indices})
        return config

    def mol_compute(self, nlist, positions, mol_nlist, mol_positions, box, training):
        raise AttributeError('You must implement mol_compute method')

    def compute(self, nlist, positions, box, training):
        nl = _as_nlist(nlist)
        if self._mol_flat is None or self._mol_flat.device != nl.device:
            self._mol_flat = torch.as_tensor(self.mol_indices, dtype=torch.int64, device=nl.device).reshape(-1)
        # one dummy particle in front, so that the 0 fill of mol_indices slices zeros
        ap = torch.cat([torch.zeros((1, 4), dtype=positions.dtype, device=positions.device), positions], dim=0)
        an = torch.cat([torch.zeros((1, self.nneighbor_cutoff, 4), dtype=nl.dtype, device=nl.device), nl.ad], dim=0)
        mol_positions = ap.index_select(0, self._mol_flat).reshape(-1, self.MN, 4)
        mol_nlist = an.index_select(0, self._mol_flat).reshape(-1, self.MN, self.nneighbor_cutoff, 4)
        inputs = [nl, positions, mol_nlist, mol_positions, box, training]
        return self.mol_compute(*inputs[:self._mol_arg_count])


def find_molecules(system):
    """utils.py:236-285 for the stand-in System: connected components of ``system.bonds``
    (a list of index pairs; absent = no bonds, every particle its own molecule), each sorted,
    ordered by smallest index."""
    N = system.N
    adj = [[] for _ in range(N)]
    for a, b in getattr(system, "bonds", []):
        adj[a].append(b)
        adj[b].append(a)
    seen, mapping = [False] * N, []
    for i in range(N):
        if seen[i]:
            continue
        stack, mol = [i], []
        seen[i] = True
        while stack:
            k = stack.pop()
            mol.append(k)
            for j in adj[k]:
                if not seen[j]:
                    seen[j] = True
                    stack.append(j)
        mapping.append(sorted(mol))
    return mapping
