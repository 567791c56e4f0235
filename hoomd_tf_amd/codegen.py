"""Generated kernels for TRACED pair energies: the fast path for a model outside the lowered zoo.

The reference's defining capability is an arbitrary ``compute()`` (htf/simmodel.py:87-121): any TF graph of the neighbor tensor is
legal, and every notebook writes its own energy.  Until round 5 anything that was not LJ / WCA / a rinv polynomial / a Gaussian / an
MLP ran as ~20 eager torch ops + autograd per step (17-23 x slower than a lowered model at C2 / C3).  Here an ELEMENTWISE pair
energy -- any expression of ``s = nlist_rinv(nlist)`` and ``r = safe_norm(nlist[:, :, :3], axis=2)`` built from + - * /, integer
and real powers, exp, log, tanh, sqrt, abs, erf / erfc, sigmoid, softplus, sin / cos, minimum / maximum, comparisons, ``where`` and cast masks -- is kept symbolic
(:class:`Node`), differentiated in forward mode with respect to r, emitted as the body of pair_math.h's
``pair_eval_f<HTF_POT_JIT>`` and compiled for gfx950 with ``hipcc --genco`` around the library's own row loops
(csrc/jit_unit.hip): the one-kernel step, its virial form, the streaming evaluator -- same launch geometry, same reductions as
the built-in closed forms.  The code object is cached by the hash of everything that went into it.  No second backend, no
Triton: the generated text is twenty lines of C in the middle of the hand-written kernels.

Semantics are TensorFlow's for the same expression: ``s`` is exactly 0 on padded and masked slots with d s / d r = -s^2
elsewhere (simmodel.py:618-635), ``r`` is |x + 1e-7| with gradient t / r (simmodel.py:581-594), comparisons and cast masks carry
no gradient, nlist_forces = 2 dE/dx (simmodel.py:548).  One restriction: the energy and its derivative must VANISH on a padded
slot (s = 0, r = sqrt(3) 1e-7) -- the kernels skip the zero padding behind a row's live slots -- which every physical pair
energy written on ``nlist_rinv`` does; an expression that does not (an unmasked Morse on ``safe_norm`` alone) keeps the torch route.
"""
import hashlib
import math
import os
import subprocess
import tempfile

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
ARCH = "gfx950"
PAD_R = math.sqrt(3.0) * 1e-7


class Node:
    """One node of an elementwise expression over the pair slot's (s, r, rn): op + children (+ a constant)."""
    __slots__ = ("op", "args", "value", "_key")

    def __init__(self, op, args=(), value=None):
        self.op, self.args, self.value, self._key = op, tuple(args), value, None

    def key(self):
        """Structural identity (hashable; kept on the node: a shared subexpression is walked once)."""
        if self._key is None:
            self._key = (self.op, self.value) + tuple(a.key() for a in self.args)
        return self._key


def const(v):
    return Node("const", value=float(v))


def wrap(x):
    if isinstance(x, Node):
        return x
    if isinstance(x, (int, float, np.floating, np.integer)):
        return const(x)
    if isinstance(x, torch.Tensor) and x.numel() == 1:
        return const(float(x))
    raise TypeError("cannot use %r in a traced pair energy (scalars and expressions of nlist_rinv / safe_norm only)" % (type(x),))


def param(k):
    """Weight k of the traced energy: a kernel ARGUMENT (``p.theta[k]``, the potential's device parameter vector), not a constant
    of the generated text -- a weight write costs a 4-byte copy, not a recompile, and the energy can be trained (round 6)."""
    return Node("param", value=int(k))


def params_of(node, seen=None, out=None):
    """The weight indices an expression reads, ascending."""
    seen, out = (set(), set()) if seen is None else (seen, out)
    if id(node) not in seen:
        seen.add(id(node))
        if node.op == "param":
            out.add(int(node.value))
        for a in node.args:
            params_of(a, seen, out)
    return sorted(out)


def rowsum(k):
    """Row sum k of a per-particle expression (round 6): sum over the neighbor axis of a traced pair expression -- the argument of a
    ROW FUNCTION, energy_i = F(sum_j g(r_ij)) (an embedding term, a coordination-number restraint)."""
    return Node("rowsum", value=int(k))


def rowsums_of(node, seen=None, out=None):
    """The row-sum indices an expression reads, ascending."""
    seen, out = (set(), set()) if seen is None else (seen, out)
    if id(node) not in seen:
        seen.add(id(node))
        if node.op == "rowsum":
            out.add(int(node.value))
        for a in node.args:
            rowsums_of(a, seen, out)
    return sorted(out)


def remap_rowsums(node, mapping, memo=None):
    """The same expression with row sum k renamed mapping[k]."""
    memo = {} if memo is None else memo
    if id(node) in memo:
        return memo[id(node)]
    if node.op == "rowsum":
        out = node if mapping[node.value] == node.value else rowsum(mapping[node.value])
    elif not node.args:
        out = node
    else:
        args = tuple(remap_rowsums(a, mapping, memo) for a in node.args)
        out = node if all(x is y for x, y in zip(args, node.args)) else Node(node.op, args, node.value)
    memo[id(node)] = out
    return out


S, R, RN = Node("s"), Node("r"), Node("rn")   # nlist_rinv, safe_norm, the plain norm (masks only: no gradient)
TJ, TI = Node("tj"), Node("ti")               # the neighbor's type nlist[i, j, 3] and the row particle's own positions[i, 3], as floats
UNARY = ("neg", "exp", "log", "tanh", "sqrt", "abs", "square", "mask", "erf", "erfc", "sigmoid", "softplus", "sin", "cos")
BINARY = ("add", "sub", "mul", "div", "min", "max")
COMPARE = ("lt", "le", "gt", "ge", "eq", "ne")
MAX_TABLE = 1024                               # entries of a ``gather`` table (a species-pair parameter matrix)


def table(values, index):
    """``tf.gather(values, index)`` with a constant 1-D ``values`` and an index EXPRESSION (types: ti * ntypes + tj): no gradient
    flows into an index, and the table is a constant of the generated unit."""
    vals = tuple(float(v) for v in np.asarray(values, dtype=np.float64).reshape(-1))
    if not 1 <= len(vals) <= MAX_TABLE:
        raise ValueError("a traced gather table holds 1..%d constants (got %d)" % (MAX_TABLE, len(vals)))
    return Node("table", (wrap(index),), value=vals)


def reads(node, op, seen=None):
    """Does the expression contain a node of kind ``op``?"""
    seen = set() if seen is None else seen
    if id(node) in seen:
        return False
    seen.add(id(node))
    return node.op == op or any(reads(a, op, seen) for a in node.args)


# --------------------------------------------------------------------------- evaluation in torch (the reference, and eager values)
def evaluate(node, s, r, rn, memo=None, tj=None, ti=None, params=None, rows=None):
    """The expression on torch tensors (any dtype): what the generated kernel computes per slot.  ``tj`` / ``ti``: the type
    leaves (broadcastable to ``s``), needed only by expressions that read them; ``params``: the weights (one-element tensors,
    on the autograd graph when they are being trained)."""
    memo = {} if memo is None else memo
    k = id(node)
    if k in memo:
        return memo[k]
    op = node.op
    a = [evaluate(x, s, r, rn, memo, tj, ti, params, rows) for x in node.args]
    if op == "rowsum":
        if rows is None or node.value >= len(rows):
            raise ValueError("the expression reads row sum %d: evaluate() needs rows" % node.value)
        out = rows[node.value]
    elif op == "param":
        if params is None or node.value >= len(params):
            raise ValueError("the expression reads weight %d: evaluate() needs params" % node.value)
        out = params[node.value].reshape(()).to(dtype=s.dtype, device=s.device)
    elif op == "s":
        out = s
    elif op == "r":
        out = r
    elif op == "rn":
        out = rn
    elif op in ("tj", "ti"):
        out = tj if op == "tj" else ti
        if out is None:
            raise ValueError("the expression reads particle types: evaluate() needs tj / ti")
    elif op == "table":
        vals = torch.as_tensor(node.value, dtype=s.dtype, device=s.device)
        out = vals[a[0].detach().to(torch.int64).clamp(0, len(node.value) - 1)]
    elif op == "const":
        out = torch.as_tensor(node.value, dtype=s.dtype, device=s.device)
    elif op == "neg":
        out = -a[0]
    elif op == "exp":
        out = torch.exp(a[0])
    elif op == "log":
        out = torch.log(a[0])
    elif op == "tanh":
        out = torch.tanh(a[0])
    elif op == "sqrt":
        out = torch.sqrt(a[0])
    elif op == "abs":
        out = torch.abs(a[0])
    elif op == "square":
        out = a[0] * a[0]
    elif op == "mask":
        out = a[0].to(s.dtype)
    elif op == "erf":
        out = torch.erf(a[0])
    elif op == "erfc":
        out = torch.erfc(a[0])
    elif op == "sigmoid":
        out = torch.sigmoid(a[0])
    elif op == "softplus":
        out = torch.nn.functional.softplus(a[0], threshold=1e30)
    elif op == "sin":
        out = torch.sin(a[0])
    elif op == "cos":
        out = torch.cos(a[0])
    elif op == "add":
        out = a[0] + a[1]
    elif op == "sub":
        out = a[0] - a[1]
    elif op == "mul":
        out = a[0] * a[1]
    elif op == "div":
        out = a[0] / a[1]
    elif op == "min":
        out = torch.minimum(a[0], a[1])
    elif op == "max":
        out = torch.maximum(a[0], a[1])
    elif op == "pow":
        out = a[0] ** node.value
    elif op in COMPARE:
        out = {"lt": torch.lt, "le": torch.le, "gt": torch.gt, "ge": torch.ge, "eq": torch.eq, "ne": torch.ne}[op](a[0], a[1])
    elif op == "where":
        out = torch.where(a[0], a[1], a[2])
    else:
        raise ValueError("unknown op %r" % op)
    memo[k] = out
    return out


# --------------------------------------------------------------------------- forward-mode code generation
class _Emitter:
    def __init__(self):
        self.lines, self.memo, self.n = [], {}, 0

    def tmp(self, expr):
        name = "t%d" % self.n
        self.n += 1
        self.lines.append("const float %s = %s;" % (name, expr))
        return name

    # erfc(x) = t P(t) exp(-x^2), t = 1 / (1 + 0.4 x), P of degree 10 (Chebyshev fit of erfcx(x) / t on t in (0, 1]: 8.5e-9 in
    # fp64); exp(-x^2) with the rounding of x^2 and of its product with log2(e) carried as a first-order correction.  ~3 ulp to
    # x = 4 (4e-7 relative; 1.4e-6 at x = 9), 24 vector instructions and two transcendentals where the device libm's erfcf takes
    # ~3 x that with branches -- and exp(-x^2), which the derivative needs anyway, comes out of it (round 6: the ionic model
    # 1.35 x lowered LJ -> see DESIGN 3.6).
    _ERFC_P = (0.225675831, 0.225676317, 0.20760072, 0.171890572, 0.118095123, 0.0871206414, -0.0537931659, 0.146803245, -0.243041731,
               0.143841069, -0.0298686285)

    def erfc_gauss(self, a):
        """-> (erfc(a), 2 / sqrt(pi) exp(-a^2)) as temporaries."""
        ax = self.tmp("fabsf(%s)" % a)
        t = self.tmp("__builtin_amdgcn_rcpf(fmaf(0.4f, %s, 1.0f))" % ax)
        poly = "%.9gf" % self._ERFC_P[-1]
        for c in self._ERFC_P[-2::-1]:
            poly = "fmaf(%s, %s, %s)" % (poly, t, self.lit(c))
        pv = self.tmp(poly)
        sq = self.tmp("%s * %s" % (ax, ax))
        lo = self.tmp("fmaf(%s, %s, -%s)" % (ax, ax, sq))
        q = self.tmp("-%s * 1.4426950408889634f" % sq)
        r1 = self.tmp("fmaf(-%s, 1.4426950408889634f, -%s)" % (sq, q))
        corr = self.tmp("fmaf(-%s, 1.4426950408889634f, %s)" % (lo, r1))
        ex = self.tmp("__builtin_amdgcn_exp2f(%s)" % q)
        e1 = self.tmp("fmaf(%s * %s, 0.6931471805599453f, %s)" % (ex, corr, ex))
        vp = self.tmp("%s * %s * %s" % (t, pv, e1))
        v = self.tmp("%s < 0.0f ? 2.0f - %s : %s" % (a, vp, vp))
        return v, self.tmp("1.1283791670955126f * %s" % e1)

    def weight(self, k):
        """w<k> = p.theta[k]: read once per body (a uniform load from the potential's device parameter vector)."""
        name = "w%d" % int(k)
        seen = getattr(self, "_weights", None)
        if seen is None:
            seen = self._weights = set()
        if k not in seen:
            seen.add(k)
            self.lines.append("const float %s = p.theta[%d];" % (name, int(k)))
        return name

    @staticmethod
    def lit(v):
        if v != v or v in (float("inf"), float("-inf")):
            raise ValueError("non-finite constant in a traced pair energy")
        t = repr(float(np.float32(v))) + "f"      # ('2.0f', '1e-07f': repr of a float always carries a '.' or an exponent)
        # a negative literal travels in parentheses: "-%s" of it would otherwise read "--2.0f" -- the decrement of an rvalue
        # (neg, sub, abs and sigmoid of a negative constant or folded weight did not compile: ADVICE r5)
        return "(%s)" % t if t.startswith("-") else t

    def emit(self, node):
        """-> (value, derivative with respect to r): C expressions naming temporaries; derivative None = identically zero."""
        k = id(node)
        if k in self.memo:
            return self.memo[k]
        op = node.op
        if op == "s":
            out = ("s", "ds")
        elif op == "r":
            out = ("r", "1.0f")
        elif op == "rn":
            out = (self.tmp("__builtin_amdgcn_sqrtf(x * x + y * y + z * z)"), None)
        elif op == "rowsum":
            out = ("rho", "1.0f")          # (a row function of ONE sum: the derivative slot is d / d rho)
        elif op == "const":
            out = (self.lit(node.value), None)
        elif op == "param":
            out = (self.weight(node.value), None)
        elif op in ("tj", "ti"):
            out = (op, None)
        elif op == "table":
            # (int)index clamped into the table; a handful of constants become a select chain (v_cndmask, no memory), a
            # species-pair matrix of more than 16 entries a constant array of the unit
            n = len(node.value)
            idx = "i%d" % self.n
            self.n += 1
            self.lines.append("const int %s = min(max((int)(%s), 0), %d);" % (idx, self.emit(node.args[0])[0], n - 1))
            if n <= 16:
                expr = self.lit(node.value[n - 1])
                for q in range(n - 2, -1, -1):
                    expr = "%s == %d ? %s : (%s)" % (idx, q, self.lit(node.value[q]), expr)
                out = (self.tmp(expr), None)
            else:
                arr = "tb%d" % self.n
                self.n += 1
                self.lines.append("static constexpr float %s[%d] = {%s};" % (arr, n, ", ".join(self.lit(v) for v in node.value)))
                out = (self.tmp("%s[%s]" % (arr, idx)), None)
        elif op in COMPARE:
            a, b = self.emit(node.args[0])[0], self.emit(node.args[1])[0]
            c = {"lt": "<", "le": "<=", "gt": ">", "ge": ">=", "eq": "==", "ne": "!="}[op]
            name = "c%d" % self.n
            self.n += 1
            self.lines.append("const bool %s = %s %s %s;" % (name, a, c, b))
            out = (name, None)
        elif op == "mask":
            out = (self.tmp("%s ? 1.0f : 0.0f" % self.emit(node.args[0])[0]), None)
        elif op == "where":
            c = self.emit(node.args[0])[0]
            (a, da), (b, db) = self.emit(node.args[1]), self.emit(node.args[2])
            v = self.tmp("%s ? %s : %s" % (c, a, b))
            d = None if da is None and db is None else self.tmp("%s ? %s : %s" % (c, da or "0.0f", db or "0.0f"))
            out = (v, d)
        elif op in ("add", "sub"):
            (a, da), (b, db) = self.emit(node.args[0]), self.emit(node.args[1])
            sg = "+" if op == "add" else "-"
            v = self.tmp("%s %s %s" % (a, sg, b))
            if da is None and db is None:
                d = None
            elif db is None:
                d = da
            elif da is None:
                d = db if op == "add" else self.tmp("-%s" % db)
            else:
                d = self.tmp("%s %s %s" % (da, sg, db))
            out = (v, d)
        elif op == "mul":
            (a, da), (b, db) = self.emit(node.args[0]), self.emit(node.args[1])
            v = self.tmp("%s * %s" % (a, b))
            terms = [("%s * %s" % (da, b)) if da is not None else None, ("%s * %s" % (a, db)) if db is not None else None]
            terms = [t for t in terms if t]
            out = (v, self.tmp(" + ".join(terms)) if terms else None)
        elif op == "div":
            (a, da), (b, db) = self.emit(node.args[0]), self.emit(node.args[1])
            ib = self.tmp("__builtin_amdgcn_rcpf(%s)" % b)
            v = self.tmp("%s * %s" % (a, ib))
            if da is None and db is None:
                d = None
            elif db is None:
                d = self.tmp("%s * %s" % (da, ib))
            elif da is None:
                d = self.tmp("-(%s * %s) * %s" % (v, ib, db))
            else:
                d = self.tmp("(%s - %s * %s) * %s" % (da, v, db, ib))
            out = (v, d)
        elif op == "neg":
            a, da = self.emit(node.args[0])
            out = (self.tmp("-%s" % a), None if da is None else self.tmp("-%s" % da))
        elif op == "square":
            a, da = self.emit(node.args[0])
            out = (self.tmp("%s * %s" % (a, a)), None if da is None else self.tmp("2.0f * %s * %s" % (a, da)))
        elif op == "exp":
            a, da = self.emit(node.args[0])
            v = self.tmp("__builtin_amdgcn_exp2f(%s * 1.4426950408889634f)" % a)
            out = (v, None if da is None else self.tmp("%s * %s" % (v, da)))
        elif op == "log":
            a, da = self.emit(node.args[0])
            v = self.tmp("__builtin_amdgcn_logf(%s) * 0.6931471805599453f" % a)
            out = (v, None if da is None else self.tmp("%s * __builtin_amdgcn_rcpf(%s)" % (da, a)))
        elif op == "tanh":
            a, da = self.emit(node.args[0])
            # tanh(a) = 1 - 2 / (exp(2a) + 1): one v_exp and one v_rcp; saturates cleanly (exp -> inf gives 1, -> 0 gives -1)
            ex = self.tmp("__builtin_amdgcn_exp2f(%s * 2.8853900817779268f)" % a)
            v = self.tmp("1.0f - 2.0f * __builtin_amdgcn_rcpf(%s + 1.0f)" % ex)
            out = (v, None if da is None else self.tmp("(1.0f - %s * %s) * %s" % (v, v, da)))
        elif op == "sqrt":
            a, da = self.emit(node.args[0])
            v = self.tmp("__builtin_amdgcn_sqrtf(%s)" % a)
            out = (v, None if da is None else self.tmp("0.5f * %s * __builtin_amdgcn_rcpf(%s)" % (da, v)))
        elif op == "abs":
            a, da = self.emit(node.args[0])
            out = (self.tmp("fabsf(%s)" % a), None if da is None else self.tmp("%s < 0.0f ? -%s : (%s > 0.0f ? %s : 0.0f)" % (a, da, a, da)))
        elif op in ("erf", "erfc"):
            # the real-space part of Ewald / DSF electrostatics: erfc(alpha r) / r.  d erf(a) = 2 / sqrt(pi) exp(-a^2)
            a, da = self.emit(node.args[0])
            if op == "erfc":
                v, g = self.erfc_gauss(a)
                out = (v, None if da is None else self.tmp("-%s * %s" % (g, da)))
            else:
                v = self.tmp("erff(%s)" % a)
                if da is None:
                    out = (v, None)
                else:
                    g = self.tmp("1.1283791670955126f * __builtin_amdgcn_exp2f(-(%s * %s) * 1.4426950408889634f)" % (a, a))
                    out = (v, self.tmp("%s * %s" % (g, da)))
        elif op == "sigmoid":
            a, da = self.emit(node.args[0])
            v = self.tmp("__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-%s * 1.4426950408889634f))" % a)
            out = (v, None if da is None else self.tmp("%s * (1.0f - %s) * %s" % (v, v, da)))
        elif op == "softplus":
            # log(1 + exp(a)) = max(a, 0) + log(1 + exp(-|a|)): no overflow; its derivative is the sigmoid
            a, da = self.emit(node.args[0])
            ex = self.tmp("__builtin_amdgcn_exp2f(-fabsf(%s) * 1.4426950408889634f)" % a)
            v = self.tmp("fmaxf(%s, 0.0f) + __builtin_amdgcn_logf(1.0f + %s) * 0.6931471805599453f" % (a, ex))
            if da is None:
                out = (v, None)
            else:
                sg = self.tmp("%s >= 0.0f ? __builtin_amdgcn_rcpf(1.0f + %s) : %s * __builtin_amdgcn_rcpf(1.0f + %s)" % (a, ex, ex, ex))
                out = (v, self.tmp("%s * %s" % (sg, da)))
        elif op in ("sin", "cos"):
            # v_sin_f32 / v_cos_f32 take REVOLUTIONS: the argument is scaled by 1 / 2 pi and reduced to [0, 1) first (v_fract_f32) --
            # two instructions each, where libm's full-range sinf / cosf is a 500 KB unit; absolute error ~1e-6 + 4e-7 |a|
            a, da = self.emit(node.args[0])
            rev = self.tmp("__builtin_amdgcn_fractf(%s * 0.15915494309189535f)" % a)
            v = self.tmp("__builtin_amdgcn_%s(%s)" % ("sinf" if op == "sin" else "cosf", rev))
            if da is None:
                out = (v, None)
            else:
                other = self.tmp("__builtin_amdgcn_%s(%s)" % ("cosf" if op == "sin" else "sinf", rev))
                out = (v, self.tmp("%s%s * %s" % ("" if op == "sin" else "-", other, da)))
        elif op in ("min", "max"):
            (a, da), (b, db) = self.emit(node.args[0]), self.emit(node.args[1])
            cmpop = "<=" if op == "min" else ">="
            v = self.tmp("%s %s %s ? %s : %s" % (a, cmpop, b, a, b))
            d = None if da is None and db is None else self.tmp("%s %s %s ? %s : %s" % (a, cmpop, b, da or "0.0f", db or "0.0f"))
            out = (v, d)
        elif op == "pow":
            a, da = self.emit(node.args[0])
            p = node.value
            if float(p).is_integer() and 0 <= abs(p) <= 64:
                n = int(abs(p))
                if n == 0:
                    out = ("1.0f", None)
                    self.memo[k] = out
                    return out
                # a^n and a^(n-1) by repeated squaring
                def ipow(base, e):
                    acc, cur = None, base
                    while e:
                        if e & 1:
                            acc = cur if acc is None else self.tmp("%s * %s" % (acc, cur))
                        e >>= 1
                        if e:
                            cur = self.tmp("%s * %s" % (cur, cur))
                    return acc or "1.0f"
                pm1 = ipow(a, n - 1) if n > 1 else "1.0f"
                pn = self.tmp("%s * %s" % (pm1, a)) if n > 1 else a
                if p > 0:
                    out = (pn, None if da is None else self.tmp("%s * %s * %s" % (self.lit(float(n)), pm1, da)))
                else:
                    inv = self.tmp("__builtin_amdgcn_rcpf(%s)" % pn)
                    out = (inv, None if da is None else self.tmp("-%s * %s * __builtin_amdgcn_rcpf(%s) * %s" % (self.lit(float(n)), inv, a, da)))
            else:
                lg = self.tmp("__builtin_amdgcn_logf(%s)" % a)
                v = self.tmp("__builtin_amdgcn_exp2f(%s * %s)" % (self.lit(p), lg))
                # d a^p = p a^(p-1): its own exp2 (p v / a would be 0 * inf at a = 0, where TensorFlow's pow gradient is an exact 0 for p > 1)
                out = (v, None if da is None else self.tmp("%s * __builtin_amdgcn_exp2f(%s * %s) * %s" % (self.lit(p), self.lit(p - 1.0), lg, da)))
        else:
            raise ValueError("unknown op %r" % op)
        self.memo[k] = out
        return out


def generate_body(node):
    """The statements pair_math.h splices into pair_eval_f<HTF_POT_JIT>: assign ``e`` and ``dedr`` from s, ds, r, x, y, z, tj, ti."""
    em = _Emitter()
    v, d = em.emit(node)
    em.lines.append("e = %s;" % v)
    em.lines.append("dedr = %s;" % (d or "0.0f"))
    return "\n".join(em.lines)


def generate_row_fn(node):
    """The statements pair_math.h splices into row_function<HTF_POT_JIT>: ``Fv`` = F(rho) and ``dF`` = F'(rho) of a per-particle
    function of ONE row sum (rho = sum_j e_ij of the unit's pair body).  The kernels accumulate a row's (2 de/dr rhat, e) as for any
    pair energy and finish it with forces x dF, energy = Fv (virial x |dF|: simmodel.py:509-523 takes the pair forces' norms)."""
    if len(rowsums_of(node)) > 1 or any(reads(node, leaf) for leaf in ("s", "r", "rn", "tj", "ti")):
        raise ValueError("a row function is an expression of one row sum")
    em = _Emitter()
    v, d = em.emit(node)
    em.lines.append("Fv = %s;" % v)
    em.lines.append("dF = %s;" % (d or "0.0f"))
    return "\n".join(em.lines)


class _JetEmitter(_Emitter):
    """Forward-mode JETS over (r', w_k) for the training sweep: every node yields (v, v_r, {k: v_w}, {k: v_rw}) -- value, d / dr',
    d / dw_k and the MIXED second derivative d2 / (dr' dw_k) -- as C expressions (None = identically zero).  The loss is
    differentiated through a force (tensorflowcompute.py:347-370 via simmodel.py:526-555), i.e. through de/dr': the mixed term is
    what pair_eval_grad<HTF_POT_JIT> turns into d nlist_forces / d w_k.  One rule per op shape: a unary f with f', f''
    (v_rw = f'' a_r a_w + f' a_rw), products, selects."""

    def __init__(self, weights):
        super().__init__()
        self.W = list(weights)
        self.jm = {}

    # -- small algebra on optional expressions
    def m(self, *xs):
        if any(x is None for x in xs):
            return None
        return self.tmp(" * ".join(xs))

    def a(self, *xs):
        xs = [x for x in xs if x is not None]
        if not xs:
            return None
        return xs[0] if len(xs) == 1 else self.tmp(" + ".join(xs))

    def ng(self, x):
        return None if x is None else self.tmp("-%s" % x)

    def sel(self, c, x, y):
        return None if x is None and y is None else self.tmp("%s ? %s : %s" % (c, x or "0.0f", y or "0.0f"))

    def unary(self, A, v, f1, f2):
        av, ar, aw, arw = A
        return (v, self.m(f1, ar), {k: self.m(f1, aw.get(k)) for k in self.W},
                {k: self.a(self.m(f2, ar, aw.get(k)) if f2 is not None else None, self.m(f1, arw.get(k))) for k in self.W})

    def product(self, A, B):
        (av, ar, aw, arw), (bv, br, bw, brw) = A, B
        return (self.tmp("%s * %s" % (av, bv)), self.a(self.m(ar, bv), self.m(av, br)),
                {k: self.a(self.m(aw.get(k), bv), self.m(av, bw.get(k))) for k in self.W},
                {k: self.a(self.m(arw.get(k), bv), self.m(ar, bw.get(k)), self.m(aw.get(k), br), self.m(av, brw.get(k))) for k in self.W})

    def recip(self, B):
        inv = self.tmp("__builtin_amdgcn_rcpf(%s)" % B[0])
        inv2 = self.tmp("%s * %s" % (inv, inv))
        return self.unary(B, inv, self.tmp("-%s" % inv2), self.tmp("2.0f * %s * %s" % (inv2, inv)))

    def ipow(self, base, e):
        if e == 0:
            return "1.0f"
        acc, cur = None, base
        while e:
            if e & 1:
                acc = cur if acc is None else self.tmp("%s * %s" % (acc, cur))
            e >>= 1
            if e:
                cur = self.tmp("%s * %s" % (cur, cur))
        return acc

    def power(self, a, q):
        """a^q for a real constant q (small integers by repeated multiplication, negative ones through v_rcp)."""
        if float(q).is_integer() and abs(q) <= 64:
            n = int(abs(q))
            pn = self.ipow(a, n)
            return pn if q >= 0 else self.tmp("__builtin_amdgcn_rcpf(%s)" % pn)
        return self.tmp("__builtin_amdgcn_exp2f(%s * __builtin_amdgcn_logf(%s))" % (self.lit(q), a))

    def jet(self, node):
        k = id(node)
        if k in self.jm:
            return self.jm[k]
        op, Z = node.op, {}
        if op == "s":
            out = ("s", "ds", Z, Z)
        elif op == "r":
            out = ("r", "1.0f", Z, Z)
        elif op == "param":
            out = (self.weight(node.value), None, {node.value: "1.0f"}, Z)
        elif op in ("rn", "const", "tj", "ti", "table", "mask") or op in COMPARE:
            out = (self.emit(node)[0], None, Z, Z)      # (no derivative flows: the base emitter's value)
        elif op == "where":
            c = self.emit(node.args[0])[0]
            A, B = self.jet(node.args[1]), self.jet(node.args[2])
            out = (self.tmp("%s ? %s : %s" % (c, A[0], B[0])), self.sel(c, A[1], B[1]),
                   {q: self.sel(c, A[2].get(q), B[2].get(q)) for q in self.W}, {q: self.sel(c, A[3].get(q), B[3].get(q)) for q in self.W})
        elif op in ("min", "max"):
            A, B = self.jet(node.args[0]), self.jet(node.args[1])
            c = self.tmp_bool("%s %s %s" % (A[0], "<=" if op == "min" else ">=", B[0]))
            out = (self.tmp("%s ? %s : %s" % (c, A[0], B[0])), self.sel(c, A[1], B[1]),
                   {q: self.sel(c, A[2].get(q), B[2].get(q)) for q in self.W}, {q: self.sel(c, A[3].get(q), B[3].get(q)) for q in self.W})
        elif op in ("add", "sub"):
            A, B = self.jet(node.args[0]), self.jet(node.args[1])
            if op == "sub":
                B = (B[0], self.ng(B[1]), {q: self.ng(B[2].get(q)) for q in self.W}, {q: self.ng(B[3].get(q)) for q in self.W})
                v = self.tmp("%s - %s" % (A[0], B[0]))
            else:
                v = self.tmp("%s + %s" % (A[0], B[0]))
            out = (v, self.a(A[1], B[1]), {q: self.a(A[2].get(q), B[2].get(q)) for q in self.W},
                   {q: self.a(A[3].get(q), B[3].get(q)) for q in self.W})
        elif op == "mul":
            out = self.product(self.jet(node.args[0]), self.jet(node.args[1]))
        elif op == "div":
            out = self.product(self.jet(node.args[0]), self.recip(self.jet(node.args[1])))
        elif op == "neg":
            A = self.jet(node.args[0])
            out = (self.tmp("-%s" % A[0]), self.ng(A[1]), {q: self.ng(A[2].get(q)) for q in self.W}, {q: self.ng(A[3].get(q)) for q in self.W})
        elif op == "pow":
            A, pw = self.jet(node.args[0]), float(node.value)
            if pw == 0.0:
                out = ("1.0f", None, Z, Z)
            else:
                f1 = self.tmp("%s * %s" % (self.lit(pw), self.power(A[0], pw - 1.0)))
                f2 = None if pw == 1.0 else self.tmp("%s * %s" % (self.lit(pw * (pw - 1.0)), self.power(A[0], pw - 2.0)))
                out = self.unary(A, self.power(A[0], pw), f1, f2)
        else:
            A = self.jet(node.args[0])
            a = A[0]
            if op == "square":
                out = self.unary(A, self.tmp("%s * %s" % (a, a)), self.tmp("2.0f * %s" % a), "2.0f")
            elif op == "exp":
                v = self.tmp("__builtin_amdgcn_exp2f(%s * 1.4426950408889634f)" % a)
                out = self.unary(A, v, v, v)
            elif op == "log":
                ia = self.tmp("__builtin_amdgcn_rcpf(%s)" % a)
                out = self.unary(A, self.tmp("__builtin_amdgcn_logf(%s) * 0.6931471805599453f" % a), ia, self.tmp("-(%s * %s)" % (ia, ia)))
            elif op == "tanh":
                ex = self.tmp("__builtin_amdgcn_exp2f(%s * 2.8853900817779268f)" % a)
                t = self.tmp("1.0f - 2.0f * __builtin_amdgcn_rcpf(%s + 1.0f)" % ex)
                u = self.tmp("1.0f - %s * %s" % (t, t))
                out = self.unary(A, t, u, self.tmp("-2.0f * %s * %s" % (t, u)))
            elif op == "sqrt":
                q = self.tmp("__builtin_amdgcn_sqrtf(%s)" % a)
                iq = self.tmp("__builtin_amdgcn_rcpf(%s)" % q)
                out = self.unary(A, q, self.tmp("0.5f * %s" % iq), self.tmp("-0.25f * %s * %s * %s" % (iq, iq, iq)))
            elif op == "abs":
                out = self.unary(A, self.tmp("fabsf(%s)" % a), self.tmp("%s < 0.0f ? -1.0f : (%s > 0.0f ? 1.0f : 0.0f)" % (a, a)), None)
            elif op in ("erf", "erfc"):
                if op == "erfc":
                    v, g = self.erfc_gauss(a)
                    out = self.unary(A, v, self.tmp("-%s" % g), self.tmp("2.0f * %s * %s" % (a, g)))
                else:
                    g = self.tmp("1.1283791670955126f * __builtin_amdgcn_exp2f(-(%s * %s) * 1.4426950408889634f)" % (a, a))
                    out = self.unary(A, self.tmp("erff(%s)" % a), g, self.tmp("-2.0f * %s * %s" % (a, g)))
            elif op == "sigmoid":
                v = self.tmp("__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-(%s) * 1.4426950408889634f))" % a)
                f1 = self.tmp("%s * (1.0f - %s)" % (v, v))
                out = self.unary(A, v, f1, self.tmp("%s * (1.0f - 2.0f * %s)" % (f1, v)))
            elif op == "softplus":
                ex = self.tmp("__builtin_amdgcn_exp2f(-fabsf(%s) * 1.4426950408889634f)" % a)
                v = self.tmp("fmaxf(%s, 0.0f) + __builtin_amdgcn_logf(1.0f + %s) * 0.6931471805599453f" % (a, ex))
                sg = self.tmp("%s >= 0.0f ? __builtin_amdgcn_rcpf(1.0f + %s) : %s * __builtin_amdgcn_rcpf(1.0f + %s)" % (a, ex, ex, ex))
                out = self.unary(A, v, sg, self.tmp("%s * (1.0f - %s)" % (sg, sg)))
            elif op in ("sin", "cos"):
                rev = self.tmp("__builtin_amdgcn_fractf(%s * 0.15915494309189535f)" % a)
                sn, cs = self.tmp("__builtin_amdgcn_sinf(%s)" % rev), self.tmp("__builtin_amdgcn_cosf(%s)" % rev)
                out = self.unary(A, sn, cs, self.tmp("-%s" % sn)) if op == "sin" else self.unary(A, cs, self.tmp("-%s" % sn), self.tmp("-%s" % cs))
            else:
                raise ValueError("unknown op %r" % op)
        self.jm[k] = out
        return out

    def tmp_bool(self, expr):
        name = "c%d" % self.n
        self.n += 1
        self.lines.append("const bool %s = %s;" % (name, expr))
        return name


def generate_train_body(node, nparams):
    """The statements pair_math.h splices into pair_eval_grad<HTF_POT_JIT>: e, dedr and, per weight k < nparams, dedw[k] and
    d2edrdw[k], from s, ds, r, x, y, z, tj and w_k = p.theta[k]."""
    em = _JetEmitter(range(int(nparams)))
    v, vr, vw, vrw = em.jet(node)
    em.lines.append("e = %s;" % v)
    em.lines.append("dedr = %s;" % (vr or "0.0f"))
    for k in range(int(nparams)):
        em.lines.append("dedw[%d] = %s;" % (k, vw.get(k) or "0.0f"))
        em.lines.append("d2edrdw[%d] = %s;" % (k, vrw.get(k) or "0.0f"))
    return "\n".join(em.lines)


def unit_text(node, row=None):
    """What identifies (and is compiled into) the generated unit of an expression: its forward body; when it reads weights, a
    marker line with their number followed by the training body; when the per-particle energy is a FUNCTION of the row's sum
    (``row``: an expression of rowsum(0)), a marker line followed by the row function (no training body then: such a model trains on
    the torch route)."""
    body = generate_body(node)
    if row is not None:
        ks = sorted(set(params_of(node)) | set(params_of(row)))
        if ks:
            body += "\n//@weights %d" % (ks[-1] + 1)
        return body + "\n//@row\n" + generate_row_fn(row)
    ks = params_of(node)
    if not ks:
        return body
    n = ks[-1] + 1
    return body + "\n//@train %d\n" % n + generate_train_body(node, n)


_UNITS = {}


def unit_of(node, row=None):
    """What this process already knows about an expression, by structure: {"text": unit_text, "vanishes": vanishes_on_padding,
    "built": its code object is in the cache}.  A model that is traced at every step (training runs compute() per step) pays for
    the emitter, the padding probe and the cache look-up once."""
    k = node.key() if row is None else (node.key(), "row", row.key())
    u = _UNITS.get(k)
    if u is None:
        if len(_UNITS) >= 256:
            _UNITS.clear()
        u = _UNITS[k] = {"text": unit_text(node, row), "vanishes": None, "built": False}
    if u["vanishes"] is None:
        u["vanishes"] = vanishes_on_padding(node)
    return u


def vanishes_on_padding(node):
    """Energy and derivative of a padded slot (s = 0, ds = 0, r = sqrt(3) 1e-7, plain norm 0, neighbor type 0), in fp64: must be
    exact zeros -- whatever the row particle's own type is (tried for 0..15 when the expression reads it)."""
    for own in (range(16) if reads(node, "ti") else (0,)):
        r = torch.tensor(PAD_R, dtype=torch.float64, requires_grad=True)
        s = r * 0.0          # (s = 0 with d s / d r = 0, but ON the graph: sqrt(s) has derivative 0 * inf = NaN there, in TF too)
        rn = torch.zeros((), dtype=torch.float64)
        try:
            ks = params_of(node)
            # (weights: judged at a generic value -- an energy that vanishes on padding only for particular weights does not lower)
            pr = [torch.tensor(0.7319 + 0.211 * q, dtype=torch.float64) for q in range(ks[-1] + 1)] if ks else None
            e = evaluate(node, s, r, rn, tj=torch.zeros((), dtype=torch.float64), ti=torch.full((), float(own), dtype=torch.float64), params=pr)
            if not isinstance(e, torch.Tensor) or not bool(torch.isfinite(e)) or float(e.detach()) != 0.0:
                return False
            if e.requires_grad:
                (g,) = torch.autograd.grad(e, r, allow_unused=True)
                if g is not None and not float(g) == 0.0:
                    return False
        except Exception:  # noqa: BLE001
            return False
    return True


# --------------------------------------------------------------------------- compile + cache
def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hoomd_tf_amd.codegen: hipcc not found (set HIPCC); generated kernels need the ROCm compiler at run time")


_CACHE_DIRS = {}


def _cache_dir():
    want = os.environ.get("HTF_JIT_CACHE") or os.path.join(_HERE, "_jit_cache")
    d = _CACHE_DIRS.get(want)
    if d is None or not os.path.isdir(d):
        d = _CACHE_DIRS[want] = _probe_cache_dir(want)
    return d


def _probe_cache_dir(d):
    try:
        os.makedirs(d, exist_ok=True)
        probe = os.path.join(d, ".w%d" % os.getpid())
        open(probe, "w").close()
        os.remove(probe)
        return d
    except OSError:
        d = os.path.join(tempfile.gettempdir(), "hoomd_tf_amd_jit_%d" % os.getuid())
        os.makedirs(d, exist_ok=True)
        return d


_SOURCES = ("jit_unit.hip", "fused_eval.hip", "eval_pair.hip", "train_pair.hip", "pair_math.h", "htf_common.h", "htf_internal.h", "box_math.h")
FLAGS = ["-std=c++17", "-O3", "-ffp-contract=on", "-DHTF_BUILD"]   # (-ffp-contract=on: as fused_eval.o is built, csrc/Makefile)


_DIGEST = []


def _source_digest():
    """(once per process: the sources a running library was built from do not change under it)"""
    if not _DIGEST:
        _DIGEST.append(_read_source_digest())
    return _DIGEST[0]


def _read_source_digest():
    h = hashlib.sha256()
    for name in _SOURCES:
        with open(os.path.join(_CSRC, name), "rb") as f:
            h.update(f.read())
    with open(os.path.join(os.path.dirname(_HERE), "include", "htf_amd.h"), "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def compiler():
    """Which compiler builds a generated unit: "hiprtc" -- the library's own htf_jit_compile, libhiprtc in process, sources handed
    over as text: no compiler driver, no temporary files -- when libhiprtc can be loaded, else "hipcc" (`hipcc --genco` as a
    subprocess); HTF_JIT_COMPILER forces either.  None: neither is there."""
    want = os.environ.get("HTF_JIT_COMPILER")
    if want in ("hiprtc", "hipcc"):
        return want
    probe = (os.environ.get("HIPCC"),)
    if probe not in _COMPILER:
        _COMPILER[probe] = _find_compiler()
    return _COMPILER[probe]


_COMPILER = {}


def _find_compiler():
    try:
        from . import _lib
        if _lib._ctypes_lib.htf_jit_available():
            return "hiprtc"
    except Exception:  # noqa: BLE001
        pass
    try:
        _hipcc()
        return "hipcc"
    except RuntimeError:
        return None


def _key(body, how=None):
    how = how or compiler() or "none"
    return hashlib.sha256((body + "\0" + _source_digest() + "\0" + ARCH + " ".join(FLAGS) + "\0" + how).encode()).hexdigest()[:24]


def available(body):
    """A generated kernel for this body can be had: its code object is cached, or a compiler is there to build it."""
    if os.path.exists(os.path.join(_cache_dir(), _key(body) + ".hsaco")):
        return True
    return compiler() is not None


def _body_include(body):
    """The unit's body file: HTF_JIT_BODY_TEXT and, for an energy with weights (unit_text's ``//@train N`` marker), HTF_JIT_NPARAMS
    and HTF_JIT_TRAIN_BODY_TEXT."""
    def macro(name, text):
        return "#define %s \\\n" % name + " \\\n".join("    " + l for l in text.splitlines()) + "\n"
    body, rmark, row = body.partition("\n//@row\n")
    body, wmark, nw = body.partition("\n//@weights ")
    fwd, mark, train = body.partition("\n//@train ")
    out = macro("HTF_JIT_BODY_TEXT", fwd)
    if mark:
        n, _, tb = train.partition("\n")
        out += "#define HTF_JIT_NPARAMS %d\n" % int(n) + macro("HTF_JIT_TRAIN_BODY_TEXT", tb)
    if rmark:
        # a row function: energy_i = F(sum_j e_ij).  Its weights (if any) are read from p.theta like the body's; HTF_JIT_NWEIGHTS
        # tells the library how long the vector is (htf_jit_nparams of a unit WITHOUT a training sweep)
        if wmark:
            out += "#define HTF_JIT_NWEIGHTS %d\n" % int(nw)
        out += macro("HTF_JIT_ROW_TEXT", row)
    return out


def _compile_hiprtc(body):
    """csrc/jit_unit.hip through htf_jit_compile (csrc/jit.hip: hipRTC): every header by name and text."""
    import ctypes as C
    from . import _lib
    lib = _lib._ctypes_lib
    with open(os.path.join(_CSRC, "jit_unit.hip")) as f:
        unit = f.read().replace("#include HTF_JIT_BODY_FILE", '#include "htf_jit_body.inc"')
    headers = {"htf_jit_body.inc": _body_include(body)}
    for name in _SOURCES[1:]:
        with open(os.path.join(_CSRC, name)) as f:
            headers[name] = f.read()
    with open(os.path.join(os.path.dirname(_HERE), "include", "htf_amd.h")) as f:
        headers["htf_amd.h"] = f.read()
    names = (C.c_char_p * len(headers))(*[n.encode() for n in headers])
    texts = (C.c_char_p * len(headers))(*[t.encode() for t in headers.values()])
    opts = (C.c_char_p * len(FLAGS))(*[o.encode() for o in FLAGS])
    image, nbytes = C.c_void_p(), C.c_size_t()
    log = C.create_string_buffer(16384)
    rc = lib.htf_jit_compile(unit.encode(), ARCH.encode(), len(headers), C.cast(names, C.c_void_p), C.cast(texts, C.c_void_p), len(FLAGS),
                             C.cast(opts, C.c_void_p), C.cast(C.byref(image), C.c_void_p), C.cast(C.byref(nbytes), C.c_void_p),
                             C.cast(log, C.c_void_p), len(log))
    if rc != 0:
        raise RuntimeError("hoomd_tf_amd.codegen: hipRTC failed on the generated unit (%s):\n%s\n--- body ---\n%s"
                           % (lib.htf_last_error().decode(errors="replace"), log.value.decode(errors="replace")[-3000:], body))
    try:
        return C.string_at(image.value, nbytes.value)
    finally:
        lib.htf_jit_free(image)


def _compile_hipcc(body):
    with tempfile.TemporaryDirectory() as tmp:
        inc = os.path.join(tmp, "body.inc")
        with open(inc, "w") as f:
            f.write(_body_include(body))
        out = os.path.join(tmp, "unit.hsaco")
        cmd = [_hipcc(), "--genco", "--offload-arch=" + ARCH] + FLAGS + [
            "-I" + os.path.join(os.path.dirname(_HERE), "include"), "-I" + _CSRC, "-DHTF_JIT_BODY_FILE=\"%s\"" % inc,
            os.path.join(_CSRC, "jit_unit.hip"), "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(out):
            raise RuntimeError("hoomd_tf_amd.codegen: hipcc failed on the generated unit:\n%s\n--- body ---\n%s" % (r.stderr[-3000:], body))
        with open(out, "rb") as f:
            return f.read()


def compile_body(body):
    """-> (bytes of the gfx950 code object, its cache key).  Compiles on a miss (hipRTC in process, ~4 s; or `hipcc --genco`,
    ~6 s), loads from the cache otherwise."""
    how = compiler()
    key = _key(body, how)
    path = os.path.join(_cache_dir(), key + ".hsaco")
    if not os.path.exists(path):
        if how is None:
            raise RuntimeError("hoomd_tf_amd.codegen: neither libhiprtc nor hipcc is available: generated kernels need one of them at run time")
        image = _compile_hiprtc(body) if how == "hiprtc" else _compile_hipcc(body)
        tmp_path = path + ".%d.tmp" % os.getpid()
        with open(tmp_path, "wb") as dst:
            dst.write(image)
        os.replace(tmp_path, path)   # (atomic: several ranks may compile the same expression at once)
    with open(path, "rb") as f:
        return f.read(), key
