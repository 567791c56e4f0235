"""Traced pair energies -> generated kernel bodies (hoomd_tf_amd/codegen.py), the parts that need no GPU: tracing through the
htf.* expression layer, forward-mode differentiation of the emitted C against torch autograd (the body is executed by a tiny
host-side C harness compiled with gcc), the padding rule, and the cross-compile of the generated unit for gfx950."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

from helpers import ROOT, random_nlist


def _models(htf, x):
    r = htf.safe_norm(x[:, :, :3], axis=2)
    s = htf.nlist_rinv(x)
    return {
        "morse": htf.cast(s > 0.0, torch.float32) * (1.3 * (1.0 - htf.exp(-2.0 * (r - 1.2))) ** 2 - 1.3),
        "yukawa": 2.5 * htf.exp(-0.7 * r) * s,
        "switched_lj": htf.where(r < 2.0, 4.0 * (s ** 12 - s ** 6), 0.0 * s),
        "mix": 3.0 * s ** 9 - htf.tanh(r) * s ** 2 + htf.minimum(s ** 3, 2.0 * s) + htf.sqrt(r) * s * htf.log(1.0 + r) / (1.0 + htf.abs(r - 1.5)),
        "real_power": s ** 2.5 * htf.square(htf.maximum(r, 1.0)),
        # round 5, second half: damped electrostatics (the real-space Ewald term), a soft switch, an oscillatory tail
        "ewald_real": 1.7 * htf.erfc(0.8 * r) * s,
        "switches": htf.sigmoid(4.0 * (2.0 - r)) * s ** 6 + 0.3 * htf.softplus(1.5 - r) * s + 0.1 * htf.erf(r - 1.0) * s ** 2,
        "friedel": htf.cos(2.2 * r + 0.3) * s ** 3 + 0.2 * htf.sin(1.1 * r) * s ** 2,
        # negative constants as operands of neg / sub / abs / sigmoid ("-%s" of a literal read "--2.0f")
        "negatives": (s ** 2 * (-htf.where(r < 1.5, -2.0, -0.5 * s)) + htf.sigmoid(htf.where(r < 2.0, -2.0, 1.0 * s)) * s
                      - htf.abs(htf.where(r < 1.2, -3.0, s)) * s ** 3 - (-0.25) * s),
    }


def test_negative_literals_are_parenthesised():
    """unary minus, abs and sigmoid of a negative constant node emitted ``--2.0f`` (the decrement of an rvalue)."""
    from hoomd_tf_amd import codegen as cg
    for op in ("neg", "abs", "sigmoid", "square", "exp"):
        body = cg.generate_body(cg.Node("mul", (cg.S, cg.Node(op, (cg.const(-2.0),)))))
        assert "--" not in body and "(-2.0f)" in body, body
    body = cg.generate_body(cg.Node("sub", (cg.S, cg.const(-2.0))))
    assert "--" not in body
    with tempfile.TemporaryDirectory() as tmp:
        for k, op in enumerate(("neg", "abs", "sigmoid")):
            src = os.path.join(tmp, "n%d.c" % k)
            with open(src, "w") as f:
                f.write(HARNESS.replace("BODY", cg.generate_body(cg.Node("mul", (cg.S, cg.Node(op, (cg.const(-2.0),)))))))
            subprocess.check_call(["gcc", "-O1", "-o", os.path.join(tmp, "n%d" % k), src, "-lm"])
            out = subprocess.run([os.path.join(tmp, "n%d" % k)], input="1.0 0.0 0.0\n", capture_output=True, text=True, check=True).stdout
            want = {"neg": 2.0, "abs": 2.0, "sigmoid": 1.0 / (1.0 + np.exp(2.0))}[op]
            assert abs(float(out.split()[0]) - want / (1.0 + 1e-7 + 3e-6)) < 1e-5, (op, out)


HARNESS = r"""
#include <math.h>
#include <stdbool.h>
#include <stdio.h>
static float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static float __builtin_amdgcn_sqrtf(float x) { return sqrtf(x); }
static float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
static float __builtin_amdgcn_logf(float x) { return log2f(x); }
static float __builtin_amdgcn_fractf(float x) { return x - floorf(x); }
static float __builtin_amdgcn_sinf(float x) { return sinf(6.283185307179586f * x); }
static float __builtin_amdgcn_cosf(float x) { return cosf(6.283185307179586f * x); }
int main(void) {
    float x, y, z;
    while (scanf("%f %f %f", &x, &y, &z) == 3) {
        const float tx = x + 1e-7f, ty = y + 1e-7f, tz = z + 1e-7f;
        const float r = sqrtf(tx * tx + ty * ty + tz * tz);
        const int cond = r > 3e-6f;
        const float s = cond ? 1.0f / (r + 3e-6f) : 0.0f;
        const float ds = cond ? -(s * s) : 0.0f;
        float e = 0.0f, dedr = 0.0f;
        { BODY }
        printf("%.9g %.9g\n", e, dedr);
    }
    return 0;
}
"""


def test_traced_expressions_and_their_derivatives():
    import hoomd_tf_amd as htf
    from hoomd_tf_amd import codegen as cg
    from hoomd_tf_amd.simmodel import PairExpr, RinvPoly
    rng = np.random.default_rng(0)
    nl, _ = random_nlist(rng, 30, 16, fill=0.6, rmin=0.85, rmax=2.9, dtype=np.float64)
    x = htf.Nlist(torch.from_numpy(nl))
    s = htf.nlist_rinv(x)
    assert isinstance(4.0 * (s ** 12 - s ** 6), RinvPoly)            # the zoo keeps what it can express
    pts = nl.reshape(-1, 4)[:, :3]
    for name, e in _models(htf, x).items():
        assert isinstance(e, PairExpr) and e.lowers(), name
        # reference: torch fp64 value and d/dr of the same expression, slot by slot
        t = torch.from_numpy(pts + 1e-7)
        r = torch.sqrt((t * t).sum(dim=1)).requires_grad_(True)
        ok = r > 3e-6
        sv = torch.where(ok, 1.0 / (torch.where(ok, r, torch.ones_like(r)) + 3e-6), torch.zeros_like(r))
        rn = torch.sqrt((torch.from_numpy(pts) ** 2).sum(dim=1))
        val = cg.evaluate(e.node, sv, r, rn)
        (grad,) = torch.autograd.grad(val.sum(), r)
        # the emitted body, run on the host
        with tempfile.TemporaryDirectory() as tmp:
            src = os.path.join(tmp, "h.c")
            with open(src, "w") as f:
                f.write(HARNESS.replace("BODY", e.body()))
            exe = os.path.join(tmp, "h")
            subprocess.check_call(["gcc", "-O1", "-o", exe, src, "-lm"])
            out = subprocess.run([exe], input="\n".join("%.9g %.9g %.9g" % tuple(p) for p in pts.astype(np.float32)), capture_output=True,
                                 text=True, check=True).stdout
        got = np.array([[float(v) for v in line.split()] for line in out.strip().splitlines()])
        live = pts.any(axis=1)
        scale_e, scale_g = np.abs(val.detach().numpy()).max(), np.abs(grad.numpy()).max()
        assert np.abs(got[:, 0] - val.detach().numpy()).max() < 3e-6 * scale_e, name
        assert np.abs(got[:, 1] - grad.numpy())[live].max() < 1e-5 * scale_g, name
        assert np.all(got[~live] == 0.0), name                         # padded slots: exact zeros


def test_an_energy_that_does_not_vanish_on_padding_keeps_the_torch_route():
    import hoomd_tf_amd as htf
    rng = np.random.default_rng(1)
    nl, _ = random_nlist(rng, 12, 8, fill=0.5, rmin=0.9, rmax=2.5, dtype=np.float64)
    x = htf.Nlist(torch.from_numpy(nl))
    r = htf.safe_norm(x[:, :, :3], axis=2)
    morse = (1.0 - htf.exp(-2.0 * (r - 1.2))) ** 2          # no mask: a padded slot has energy (1 - e^2.4)^2
    assert not morse.lowers()
    assert not (htf.sqrt(htf.nlist_rinv(x)) * r).lowers()   # d sqrt(s) at s = 0: 0 * inf, NaN in TensorFlow as well
    f = htf.compute_nlist_forces(x, htf.reduce_sum(morse, axis=1))
    # == torch autograd of the same energy written in torch ops
    xx = torch.from_numpy(nl).requires_grad_(True)
    t = xx[:, :, :3] + 1e-7
    rr = torch.sqrt((t * t).sum(dim=2))
    en = ((1.0 - torch.exp(-2.0 * (rr - 1.2))) ** 2).sum(dim=1)
    (g,) = torch.autograd.grad(en.sum(), xx)
    np.testing.assert_allclose(f[:, :3].detach().numpy(), 2.0 * g.sum(dim=1)[:, :3].numpy(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(f[:, 3].detach().numpy(), en.detach().numpy(), rtol=1e-12)


@pytest.mark.parametrize("how", ["hiprtc", "hipcc"])
def test_generated_unit_cross_compiles_for_gfx950(how, monkeypatch):
    """Both run-time compilers, no GPU needed: hipRTC in process through the library's own htf_jit_compile (sources handed over as
    text; the default when libhiprtc loads) and `hipcc --genco` as a subprocess."""
    import hoomd_tf_amd as htf
    from hoomd_tf_amd import _lib, codegen as cg
    if how == "hiprtc" and not _lib._ctypes_lib.htf_jit_available():
        pytest.skip("libhiprtc not loadable")
    monkeypatch.setenv("HTF_JIT_COMPILER", how)
    assert cg.compiler() == how
    x = htf.Nlist(torch.zeros((2, 4, 4)))
    e = _models(htf, x)["yukawa"]
    image, key = cg.compile_body(e.body())
    assert (image[:4] == b"\x7fELF" or image.startswith(b"__CLANG_OFFLOAD_BUNDLE__")) and len(key) == 24
    for name in ("htf_jit_rows2_f32_store", "htf_jit_rows2_f64_nostore", "htf_jit_row1v_f32_store", "htf_jit_eval_f32", "htf_jit_eval_f64_virial"):
        assert name.encode() in image
    image2, key2 = cg.compile_body(e.body())
    assert key2 == key and image2 == image                     # from the cache


TYPED_HARNESS = HARNESS.replace('#include <stdio.h>', '#include <stdio.h>\n#define min(a, b) ((a) < (b) ? (a) : (b))\n#define max(a, b) ((a) > (b) ? (a) : (b))') \
    .replace("float x, y, z;", "float x, y, z, tj, ti;").replace('scanf("%f %f %f", &x, &y, &z) == 3', 'scanf("%f %f %f %f %f", &x, &y, &z, &tj, &ti) == 5') \
    .replace("static constexpr", "static const")


def _typed_models(htf, x, positions, ntypes, big):
    """Per-species-pair parameters of a traced energy: tables looked up by (own type, neighbor type)."""
    s = htf.nlist_rinv(x)
    r = htf.safe_norm(x[:, :, :3], axis=2)
    tj = htf.cast(x[:, :, 3], torch.int32)
    ti = htf.cast(positions[:, 3], torch.int32)
    idx = ti[:, None] * ntypes + tj
    rng = np.random.default_rng(7)
    eps = rng.uniform(0.5, 1.5, ntypes * ntypes)
    sig = rng.uniform(0.8, 1.1, ntypes * ntypes)
    models = {
        "lj_table": 4.0 * htf.gather(eps, idx) * ((htf.gather(sig, idx) * s) ** 12 - (htf.gather(sig, idx) * s) ** 6),
        "unlike_only": htf.cast(htf.not_equal(tj, ti[:, None]), torch.float32) * 2.0 * htf.exp(-0.5 * r) * s,
        "neighbor_species": htf.where(tj == 1, 3.0 * s ** 4, 0.5 * s ** 8),
    }
    if big:
        models["wide_table"] = htf.gather(rng.uniform(0.5, 1.5, 64), idx * 3 + 1) * s ** 6     # 64 entries: a constant array of the unit
    return models


def test_typed_expressions_and_their_derivatives():
    import hoomd_tf_amd as htf
    from hoomd_tf_amd import codegen as cg
    from hoomd_tf_amd.simmodel import PairExpr, PositionsInput, TypeExpr
    rng = np.random.default_rng(3)
    ntypes = 3
    nl, _ = random_nlist(rng, 40, 12, fill=0.6, rmin=0.85, rmax=2.9, dtype=np.float64)
    live = nl[:, :, :3].any(axis=2)
    nl[:, :, 3] = np.where(live, rng.integers(0, ntypes, live.shape), 0)
    pos = np.zeros((40, 4))
    pos[:, 3] = rng.integers(0, ntypes, 40)
    x = htf.Nlist(torch.from_numpy(nl))
    P = PositionsInput.wrap(torch.from_numpy(pos))
    assert isinstance(x[:, :, 3], TypeExpr) and isinstance(P[:, 3], TypeExpr)
    assert type(P[:, :3]) is torch.Tensor and type(P * 2.0) is torch.Tensor          # everything else: plain tensors
    assert torch.equal(torch.as_tensor((P[:, 3][:, None] * ntypes + x[:, :, 3]).ad), torch.from_numpy(pos[:, 3:4] * ntypes + nl[:, :, 3]))
    with pytest.raises(ValueError):
        P[:, 3] * htf.nlist_rinv(x)                                                       # [N] against [N, NN]: as torch would
    pts = nl.reshape(-1, 4)
    own = np.repeat(pos[:, 3], 12)
    for name, e in _typed_models(htf, x, P, ntypes, big=True).items():
        assert isinstance(e, PairExpr) and e.lowers(), name
        assert e.reads_own_type == (name != "neighbor_species"), name
        t = torch.from_numpy(pts[:, :3] + 1e-7)
        r = torch.sqrt((t * t).sum(dim=1)).requires_grad_(True)
        ok = r > 3e-6
        sv = torch.where(ok, 1.0 / (torch.where(ok, r, torch.ones_like(r)) + 3e-6), torch.zeros_like(r))
        rn = torch.sqrt((torch.from_numpy(pts[:, :3]) ** 2).sum(dim=1))
        val = cg.evaluate(e.node, sv, r, rn, tj=torch.from_numpy(pts[:, 3]), ti=torch.from_numpy(own))
        (grad,) = torch.autograd.grad(val.sum(), r)
        # == the expression's own torch value through the symbolic layer (what the generic route would differentiate)
        np.testing.assert_allclose(e.torch_value(torch.from_numpy(nl)).reshape(-1).numpy(), val.detach().numpy(), rtol=1e-12, atol=1e-300)
        with tempfile.TemporaryDirectory() as tmp:
            src = os.path.join(tmp, "h.c")
            with open(src, "w") as f:
                f.write(TYPED_HARNESS.replace("BODY", e.body().replace("static constexpr", "static const")))
            exe = os.path.join(tmp, "h")
            subprocess.check_call(["gcc", "-O1", "-o", exe, src, "-lm"])
            rows = np.concatenate([pts.astype(np.float32), own[:, None].astype(np.float32)], axis=1)
            out = subprocess.run([exe], input="\n".join("%.9g %.9g %.9g %.9g %.9g" % tuple(p) for p in rows), capture_output=True,
                                 text=True, check=True).stdout
        got = np.array([[float(v) for v in line.split()] for line in out.strip().splitlines()])
        lv = pts[:, :3].any(axis=1)
        scale_e, scale_g = np.abs(val.detach().numpy()).max(), np.abs(grad.numpy()).max()
        assert np.abs(got[:, 0] - val.detach().numpy()).max() < 3e-6 * scale_e, name
        assert np.abs(got[:, 1] - grad.numpy())[lv].max() < 1e-5 * scale_g, name
        assert np.all(got[~lv] == 0.0), name


def test_typed_unit_cross_compiles_and_a_table_that_leaks_onto_padding_does_not_lower():
    import hoomd_tf_amd as htf
    from hoomd_tf_amd import codegen as cg
    from hoomd_tf_amd.simmodel import PositionsInput
    x = htf.Nlist(torch.zeros((2, 4, 4)))
    P = PositionsInput.wrap(torch.zeros((2, 4)))
    models = _typed_models(htf, x, P, 3, big=True)
    for name in ("lj_table", "wide_table"):
        image, key = cg.compile_body(models[name].body())
        assert b"htf_jit_rows2_f32_store" in image and b"htf_jit_eval_f64_virial" in image, name
    idx = P[:, 3][:, None] * 3 + x[:, :, 3]
    assert not (htf.gather([1.0, 2.0, 3.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0], idx) * htf.exp(-htf.safe_norm(x[:, :, :3], axis=2))).lowers()


TRAIN_HARNESS = r"""
#include <math.h>
#include <stdbool.h>
#include <stdio.h>
static float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static float __builtin_amdgcn_sqrtf(float x) { return sqrtf(x); }
static float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
static float __builtin_amdgcn_logf(float x) { return log2f(x); }
static float __builtin_amdgcn_fractf(float x) { return x - floorf(x); }
static float __builtin_amdgcn_sinf(float x) { return sinf(6.283185307179586f * x); }
static float __builtin_amdgcn_cosf(float x) { return cosf(6.283185307179586f * x); }
struct P { float theta[8]; };
int main(void) {
    struct P p = {{ THETA }};
    float x, y, z;
    while (scanf("%f %f %f", &x, &y, &z) == 3) {
        const float tx = x + 1e-7f, ty = y + 1e-7f, tz = z + 1e-7f;
        const float r = sqrtf(tx * tx + ty * ty + tz * tz);
        const int cond = r > 3e-6f;
        const float s = cond ? 1.0f / (r + 3e-6f) : 0.0f;
        const float ds = cond ? -(s * s) : 0.0f;
        const float tj = 0.0f, ti = 0.0f;
        float e = 0.0f, dedr = 0.0f, dedw[NP], d2edrdw[NP];
        for (int k = 0; k < NP; ++k) dedw[k] = d2edrdw[k] = 0.0f;
        { BODY }
        printf("%.9g %.9g", e, dedr);
        for (int k = 0; k < NP; ++k) printf(" %.9g %.9g", dedw[k], d2edrdw[k]);
        printf("\n");
        (void)tj; (void)ti;
    }
    return 0;
}
"""


def _weighted_models(htf, x, w, a, b):
    """Traced energies with WEIGHTS: a vector weight indexed by element as upstream's LJLayer writes it, scalar Parameters, scalar
    arithmetic on them, weights inside nonlinear functions, comparisons and selects."""
    r = htf.safe_norm(x[:, :, :3], axis=2)
    s = htf.nlist_rinv(x)
    q = (w[1] * s) ** 6
    live = htf.cast(s > 0.0, torch.float32)
    return {
        "lj_layer": w[0] * 2.0 * (q * q - q),                                                # build_examples.py:349-353, on nlist_rinv
        "morse": 0.5 * a * live * ((1.0 - htf.exp(-1.0 * b * (r - 1.122))) ** 2 - 1.0),
        "yukawa_mix": (-a / 2.0) * htf.exp(-1.0 * b * r) * s * w[1] ** 2 + htf.tanh(a * s) * s ** 2,
        "switched": htf.where(r < b / 2.0, w[0] * s ** 4, htf.sqrt(a * s + 0.3) * s ** 3) + htf.minimum(a * s ** 3, w[1] * s ** 2),
        "soft": htf.softplus(a - r) * s / b + htf.sigmoid(b * (1.5 - r)) * s ** 2 * w[0] + htf.cos(a * r) * s ** 3 + htf.erfc(b * r / 4.0) * s,
    }


def test_weights_are_kernel_arguments_and_their_jets_match_double_backward():
    """Round 6: an element of a trainable leaf -- a one-element Parameter, ``w[k]`` of a weight vector, scalar
    arithmetic on such -- meeting a traced expression is WEIGHT k of the generated kernel (``p.theta[k]``), in inference and in
    training: the text does not carry its value (a write costs no recompile).  For training the emitter differentiates in forward
    mode over (r', w_k): value, d/dr', d/dw_k and the mixed d2/(dr' dw_k) -- the loss goes through a force -- each checked here,
    slot by slot, against torch's fp64 autograd (double backward for the mixed term) on the emitted C run in a host harness."""
    import hoomd_tf_amd as htf
    from hoomd_tf_amd import codegen as cg
    from hoomd_tf_amd.simmodel import PairCond, PairExpr
    rng = np.random.default_rng(5)
    nl, _ = random_nlist(rng, 24, 12, fill=0.6, rmin=0.85, rmax=2.9, dtype=np.float64)
    pts = nl.reshape(-1, 4)[:, :3]
    for name in ("lj_layer", "morse", "yukawa_mix", "switched", "soft"):
        x = htf.Nlist(torch.from_numpy(nl))
        w = torch.nn.Parameter(torch.tensor([1.1, 0.95], dtype=torch.float64))
        a = torch.nn.Parameter(torch.tensor(0.7, dtype=torch.float64))
        b = torch.nn.Parameter(torch.tensor(2.3, dtype=torch.float64))
        e = _weighted_models(htf, x, w, a, b)[name]
        assert isinstance(e, PairExpr) and e.folded == () and e.lowers(), name
        els = e.weight_elements
        P = len(els)
        assert 1 <= P <= 4 and "p.theta[" in e.body() and "//@train %d" % P in e.body(), name
        body0 = e.body()
        with torch.no_grad():
            a.mul_(1.5)                                            # a written weight: the same text, the same kernel
        x2 = htf.Nlist(torch.from_numpy(nl))
        assert _weighted_models(htf, x2, w, a, b)[name].body() == body0
        # reference: torch fp64, every slot
        t = torch.from_numpy(pts + 1e-7)
        r = torch.sqrt((t * t).sum(dim=1)).requires_grad_(True)
        ok = r > 3e-6
        sv = torch.where(ok, 1.0 / (torch.where(ok, r, torch.ones_like(r)) + 3e-6), torch.zeros_like(r))
        rn = torch.sqrt((torch.from_numpy(pts) ** 2).sum(dim=1))
        params = [tt.reshape(-1)[i] for tt, i in els]
        val = cg.evaluate(e.node, sv, r, rn, tj=torch.zeros_like(r), ti=torch.zeros_like(r), params=params)
        (gr,) = torch.autograd.grad(val.sum(), r, create_graph=True)
        leaves = [tt for tt, _ in els]
        gw, grw = [], []
        for k, (tt, i) in enumerate(els):
            # d val / d w_k per slot, and d (d val / d r) / d w_k per slot (the slots are independent: one backward per slot sum is
            # not enough for the per-slot value, so a unit perturbation through forward differences of autograd's jvp)
            one = torch.zeros_like(tt)
            one.reshape(-1)[i] = 1.0
            _, jv = torch.autograd.functional.jvp(lambda q: cg.evaluate(e.node, sv.detach(), r.detach(), rn, tj=torch.zeros_like(rn), ti=torch.zeros_like(rn),
                                                                       params=[q.reshape(-1)[ii] if t2 is tt else t2.reshape(-1)[ii] for t2, ii in els]),
                                                  (tt.detach(),), (one,))
            gw.append(jv.numpy())

            def dvdr(q, tt=tt):
                rr = r.detach().clone().requires_grad_(True)
                ok2 = rr > 3e-6
                s2 = torch.where(ok2, 1.0 / (torch.where(ok2, rr, torch.ones_like(rr)) + 3e-6), torch.zeros_like(rr))
                v = cg.evaluate(e.node, s2, rr, rn, tj=torch.zeros_like(rn), ti=torch.zeros_like(rn),
                                params=[q.reshape(-1)[ii] if t2 is tt else t2.reshape(-1)[ii] for t2, ii in els])
                (g,) = torch.autograd.grad(v.sum(), rr, create_graph=True)
                return g
            _, jv2 = torch.autograd.functional.jvp(dvdr, (tt.detach(),), (one,))
            grw.append(jv2.numpy())
        del leaves
        # the emitted training body, run on the host
        train = e.body().split("//@train %d\n" % P)[1]
        with tempfile.TemporaryDirectory() as tmp:
            src = os.path.join(tmp, "h.c")
            theta = ", ".join("%.9gf" % float(tt.reshape(-1)[i]) for tt, i in els)
            with open(src, "w") as f:
                f.write(TRAIN_HARNESS.replace("BODY", train).replace("THETA", theta).replace("NP", str(P)))
            exe = os.path.join(tmp, "h")
            subprocess.check_call(["gcc", "-O1", "-o", exe, src, "-lm"])
            out = subprocess.run([exe], input="\n".join("%.9g %.9g %.9g" % tuple(pp) for pp in pts.astype(np.float32)), capture_output=True,
                                 text=True, check=True).stdout
        got = np.array([[float(v) for v in line.split()] for line in out.strip().splitlines()])
        live = pts.any(axis=1)
        ref = [val.detach().numpy(), gr.detach().numpy()]
        for k in range(P):
            ref += [gw[k], grw[k]]
        for c, rf in enumerate(ref):
            scale = np.abs(rf).max()
            if scale == 0:
                assert np.all(got[:, c] == 0), (name, c)
                continue
            assert np.abs(got[:, c] - rf)[live].max() < 2e-5 * scale, (name, c, np.abs(got[:, c] - rf)[live].max() / scale)
        assert np.all(got[~live] == 0.0), name                     # padded slots: exact zeros in every column
    # comparisons with a weight stay symbolic too; a tensor of per-pair values still goes to torch
    x = htf.Nlist(torch.from_numpy(nl))
    w1 = torch.nn.Parameter(torch.tensor(1.7, dtype=torch.float64))
    r = htf.safe_norm(x[:, :, :3], axis=2)
    assert isinstance(r < w1, PairCond) and (r < w1).folded == ()
    traced = htf.exp(-0.7 * r) * htf.nlist_rinv(x)
    per_pair = torch.from_numpy(rng.uniform(0.5, 1.5, nl.shape[:2]))
    assert isinstance(traced * per_pair, torch.Tensor) and isinstance(traced < per_pair, torch.Tensor)
    assert isinstance(torch.tensor(2.0, dtype=torch.float64) * traced, PairExpr)      # (a plain one-element constant folds)
    # an Add with a hidden constant cannot be recovered from the autograd graph: folded during inference, as before
    e = (w1 + 1.0) * traced
    assert isinstance(e, PairExpr) and len(e.folded) == 1 and e.weight_elements == []
    # the torch value of a weighted expression carries the weights' gradients (the generic route, e.g. with HTF_NO_JIT=1)
    x = htf.Nlist(torch.from_numpy(nl))
    e = w1 * htf.exp(-0.7 * htf.safe_norm(x[:, :, :3], axis=2)) * htf.nlist_rinv(x)
    (g,) = torch.autograd.grad(e.torch_value(x.ad).sum(), w1)
    tt = torch.from_numpy(nl)[:, :, :3] + 1e-7
    rr = torch.sqrt((tt * tt).sum(dim=2))
    ss = torch.where(rr > 3e-6, 1.0 / (rr + 3e-6), torch.zeros_like(rr))
    np.testing.assert_allclose(float(g), float((torch.exp(-0.7 * rr) * ss).sum()), rtol=1e-12)


@pytest.mark.parametrize("how", ["hiprtc", "hipcc"])
def test_unit_with_weights_cross_compiles_with_its_training_sweep(how, monkeypatch):
    import hoomd_tf_amd as htf
    from hoomd_tf_amd import codegen as cg
    monkeypatch.setenv("HTF_JIT_COMPILER", how)
    if cg.compiler() != how:
        pytest.skip("%s not available" % how)
    x = htf.Nlist(torch.zeros((2, 4, 4), dtype=torch.float64))
    w = torch.nn.Parameter(torch.tensor([1.1, 0.95], dtype=torch.float64))
    a = torch.nn.Parameter(torch.tensor(0.7, dtype=torch.float64))
    b = torch.nn.Parameter(torch.tensor(2.3, dtype=torch.float64))
    e = _weighted_models(htf, x, w, a, b)["yukawa_mix"]
    image, key = cg.compile_body(e.body())
    assert b"htf_jit_train_f32" in image and b"htf_jit_train_f64" in image and b"htf_jit_nparams" in image and b"htf_jit_rows2_f32_store" in image
    plain = (htf.exp(-0.7 * htf.safe_norm(x[:, :, :3], axis=2)) * htf.nlist_rinv(x))
    image2, _ = cg.compile_body(plain.body())
    assert b"htf_jit_train_f32" not in image2


def _random_expression(htf, rng, s, r, tj, ti, depth):
    """A random elementwise expression of bounded magnitude on 0.85 <= r <= 3 (denominators kept away from zero, exponents small)."""
    if depth == 0:
        return [lambda: s, lambda: r, lambda: 0.3 * r + 0.2, lambda: s * s, lambda: float(rng.uniform(0.3, 1.5)) * s,
                lambda: htf.gather(rng.uniform(0.5, 1.5, 9), ti[:, None] * 3 + tj) * s][rng.integers(0, 6)]()
    a = _random_expression(htf, rng, s, r, tj, ti, depth - 1)
    b = _random_expression(htf, rng, s, r, tj, ti, depth - 1)
    k = rng.integers(0, 15)
    if k == 0: return a + b
    if k == 1: return a - 0.5 * b
    if k == 2: return a * b
    if k == 3: return a / (1.0 + htf.square(b))
    if k == 4: return htf.exp(-0.3 * htf.abs(a)) * b
    if k == 5: return htf.tanh(a) + b
    if k == 6: return htf.minimum(a, b)
    if k == 7: return htf.maximum(a, 0.5 * b)
    if k == 8: return htf.where(a < b, a, b * 0.7)
    if k == 9: return htf.pow(htf.abs(a) + 0.5, float(rng.uniform(0.5, 2.5))) - b
    if k == 10: return htf.sqrt(htf.square(a) + 1.0) * htf.cast(htf.equal(tj, int(rng.integers(0, 3))), torch.float32) + b
    if k == 11: return htf.erfc(0.5 * a) * b + htf.erf(a)
    if k == 12: return htf.sigmoid(a) * b + htf.softplus(-1.0 * b)
    if k == 13: return htf.cos(a) + htf.sin(0.7 * b) * a
    return htf.log(1.0 + htf.square(a)) + a ** int(rng.integers(2, 5)) * 0.1 + b


def test_random_expressions_against_autograd():
    """Thirty random expression trees (every op of the tracer, types and tables included), each masked by s^2 so that it vanishes on
    padding: the emitted C against torch-fp64 value and derivative, slot by slot."""
    import hoomd_tf_amd as htf
    from hoomd_tf_amd import codegen as cg
    from hoomd_tf_amd.simmodel import PairExpr, PositionsInput
    rng = np.random.default_rng(2026)
    nl, _ = random_nlist(rng, 24, 10, fill=0.7, rmin=0.85, rmax=3.0, dtype=np.float64)
    live = nl[:, :, :3].any(axis=2)
    nl[:, :, 3] = np.where(live, rng.integers(0, 3, live.shape), 0)
    pos = np.zeros((24, 4))
    pos[:, 3] = rng.integers(0, 3, 24)
    x = htf.Nlist(torch.from_numpy(nl))
    P = PositionsInput.wrap(torch.from_numpy(pos))
    s, r = htf.nlist_rinv(x), htf.safe_norm(x[:, :, :3], axis=2)
    tj, ti = x[:, :, 3], P[:, 3]
    pts = nl.reshape(-1, 4)
    own = np.repeat(pos[:, 3], 10)
    rows = np.concatenate([pts.astype(np.float32), own[:, None].astype(np.float32)], axis=1)
    feed = "\n".join("%.9g %.9g %.9g %.9g %.9g" % tuple(p) for p in rows)
    lv = pts[:, :3].any(axis=1)
    done = 0
    with tempfile.TemporaryDirectory() as tmp:
        for trial in range(40):
            e = htf.square(s) * _random_expression(htf, rng, s, r, tj, ti, int(rng.integers(1, 4)))
            assert isinstance(e, PairExpr)
            if not cg.unit_of(e.node)["vanishes"]:
                continue                       # (e.g. log / sqrt of something whose derivative is not finite at s = 0: the torch route)
            # (the emitted C on the host only: building these units for gfx950 -- ~8 s each -- is test_gpu_codegen.py's
            #  test_random_expressions_on_the_device, which also runs them)
            t = torch.from_numpy(pts[:, :3] + 1e-7)
            rr = torch.sqrt((t * t).sum(dim=1)).requires_grad_(True)
            ok = rr > 3e-6
            sv = torch.where(ok, 1.0 / (torch.where(ok, rr, torch.ones_like(rr)) + 3e-6), torch.zeros_like(rr))
            rn = torch.sqrt((torch.from_numpy(pts[:, :3]) ** 2).sum(dim=1))
            val = cg.evaluate(e.node, sv, rr, rn, tj=torch.from_numpy(pts[:, 3]), ti=torch.from_numpy(own))
            (grad,) = torch.autograd.grad(val.sum(), rr)
            src, exe = os.path.join(tmp, "h%d.c" % trial), os.path.join(tmp, "h%d" % trial)
            with open(src, "w") as f:
                f.write(TYPED_HARNESS.replace("BODY", e.body().replace("static constexpr", "static const")))
            subprocess.check_call(["gcc", "-O1", "-o", exe, src, "-lm"])
            out = subprocess.run([exe], input=feed, capture_output=True, text=True, check=True).stdout
            got = np.array([[float(v) for v in line.split()] for line in out.strip().splitlines()])
            scale_e, scale_g = np.abs(val.detach().numpy()).max() + 1e-30, np.abs(grad.numpy()).max() + 1e-30
            assert np.abs(got[:, 0] - val.detach().numpy()).max() < 1e-5 * scale_e, (trial, e.body())
            assert np.abs(got[:, 1] - grad.numpy())[lv].max() < 3e-5 * scale_g, (trial, e.body())
            assert np.all(got[~lv] == 0.0), (trial, e.body())
            done += 1
    assert done >= 25


# --------------------------------------------------------------------------- row functions (round 6)
ROW_HARNESS = r"""
#include <math.h>
#include <stdbool.h>
#include <stdio.h>
static float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static float __builtin_amdgcn_sqrtf(float x) { return sqrtf(x); }
static float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
static float __builtin_amdgcn_logf(float x) { return log2f(x); }
static float __builtin_amdgcn_fractf(float x) { return x - floorf(x); }
static float __builtin_amdgcn_sinf(float x) { return sinf(6.283185307179586f * x); }
static float __builtin_amdgcn_cosf(float x) { return cosf(6.283185307179586f * x); }
int main(void) {
    float rho;
    while (scanf("%f", &rho) == 1) {
        float Fv = 0.0f, dF = 0.0f;
        { BODY }
        printf("%.9g %.9g\n", Fv, dF);
    }
    return 0;
}
"""


def _row_models(htf, x):
    """Per-particle energies that feed a row sum into a nonlinearity -- what VERDICT r5 'missing 5' names: an embedded-atom term
    (Finnis-Sinclair: -A sqrt(rho) + pair repulsion), a coordination-number restraint k (n - n0)^2 on a smooth count, a
    log-density, two embeddings of two different densities, and a product of two different sums (not separable: torch route)."""
    s = htf.nlist_rinv(x)
    r = htf.safe_norm(x[:, :, :3], axis=2)
    live = htf.cast(s > 0.0, torch.float32)
    rho = htf.reduce_sum(htf.exp(-1.7 * r) * s * s, axis=1)
    n = htf.reduce_sum(live * htf.sigmoid(6.0 * (1.5 - r)), axis=1)
    phi = htf.reduce_sum(0.5 * (s ** 12 - s ** 6), axis=1)
    return {
        "finnis_sinclair": -1.3 * htf.sqrt(rho) + phi,
        "coordination": 0.25 * (n - 9.0) ** 2,
        "log_density": htf.log(1.0 + rho) * 0.7 - 0.2 * rho,
        "two_embeddings": -1.1 * htf.sqrt(rho) + htf.tanh(0.1 * n) + phi + 0.3,
        "mixed": rho * n,
    }


def test_row_functions_are_traced_and_split_into_terms_of_one_sum():
    """Round 6 (VERDICT r5 missing 5): ``htf.reduce_sum(pair_expr, axis=1)`` fed into further htf.* arithmetic stays symbolic
    (RowExpr; until now a TypeError): an energy that is a sum of functions of ONE row sum each becomes one generated unit per term,
    its row function -- F and F' of the emitted C, run on the host -- equal to torch-fp64 autograd; the value of the whole
    expression equals the same energy written in torch ops; a product of two different sums keeps the torch route."""
    import hoomd_tf_amd as htf
    from hoomd_tf_amd import codegen as cg
    from hoomd_tf_amd.simmodel import RowExpr, RowFnEnergy
    rng = np.random.default_rng(4)
    nl, _ = random_nlist(rng, 40, 24, fill=0.6, rmin=0.85, rmax=2.9, dtype=np.float64)
    x = htf.Nlist(torch.from_numpy(nl))
    models = _row_models(htf, x)
    want_terms = {"finnis_sinclair": 2, "coordination": 1, "log_density": 1, "two_embeddings": 3, "mixed": None}
    # the same energies in torch ops
    xx = torch.from_numpy(nl)
    t = xx[:, :, :3] + 1e-7
    rr = torch.sqrt((t * t).sum(dim=2))
    ss = torch.where(rr > 3e-6, 1.0 / (rr + 3e-6), torch.zeros_like(rr))
    live = (ss > 0).double()
    rho = (torch.exp(-1.7 * rr) * ss * ss).sum(dim=1)
    n = (live * torch.sigmoid(6.0 * (1.5 - rr))).sum(dim=1)
    phi = (0.5 * (ss ** 12 - ss ** 6)).sum(dim=1)
    ref = {"finnis_sinclair": -1.3 * torch.sqrt(rho) + phi, "coordination": 0.25 * (n - 9.0) ** 2,
           "log_density": torch.log(1.0 + rho) * 0.7 - 0.2 * rho, "two_embeddings": -1.1 * torch.sqrt(rho) + torch.tanh(0.1 * n) + phi + 0.3,
           "mixed": rho * n}
    for name, e in models.items():
        assert isinstance(e, RowExpr), name
        np.testing.assert_allclose(e.torch_value(xx).numpy(), ref[name].numpy(), rtol=1e-12, atol=1e-12)
        groups = e.groups()
        if want_terms[name] is None:
            assert groups is None
            continue
        assert len(groups) == want_terms[name] and all(isinstance(g, RowFnEnergy) for g in groups), name
        total = torch.zeros(len(nl), dtype=torch.float64)
        for g in groups:
            assert "//@row" in g.body() and cg.vanishes_on_padding(g.node)
            row_text = g.body().partition("\n//@row\n")[2]
            rsum = g.torch_value(xx).sum(dim=1).detach().requires_grad_(True)     # rho of this term, fp64
            fv = cg.evaluate(g.row, rsum, None, None, rows=[rsum])
            fv = fv if fv.dim() else fv.expand(len(nl))
            (dfv,) = torch.autograd.grad(fv.sum(), rsum, allow_unused=True)
            dfv = torch.zeros_like(rsum) if dfv is None else dfv
            with tempfile.TemporaryDirectory() as tmp:
                src = os.path.join(tmp, "h.c")
                with open(src, "w") as f:
                    f.write(ROW_HARNESS.replace("BODY", row_text))
                exe = os.path.join(tmp, "h")
                subprocess.check_call(["gcc", "-O1", "-o", exe, src, "-lm"])
                out = subprocess.run([exe], input="\n".join("%.9g" % v for v in rsum.detach().numpy().astype(np.float32)), capture_output=True,
                                     text=True, check=True).stdout
            got = np.array([[float(v) for v in line.split()] for line in out.strip().splitlines()])
            assert np.abs(got[:, 0] - fv.detach().numpy()).max() < 3e-6 * max(1.0, np.abs(fv.detach().numpy()).max()), name
            assert np.abs(got[:, 1] - dfv.numpy()).max() < 1e-5 * max(1.0, np.abs(dfv.numpy()).max()), name
            total = total + fv.detach()
        np.testing.assert_allclose(total.numpy(), ref[name].numpy(), rtol=1e-10, atol=1e-10)   # the terms add up to the energy


def test_row_expression_forces_on_the_torch_route_equal_autograd(monkeypatch):
    """compute_nlist_forces of a row expression with the generated kernels switched off (HTF_NO_JIT=1; also what a product of two
    sums or a training step takes): torch autograd through the traced expression == the same energy written in torch ops."""
    import hoomd_tf_amd as htf
    monkeypatch.setenv("HTF_NO_JIT", "1")
    rng = np.random.default_rng(6)
    nl, _ = random_nlist(rng, 20, 16, fill=0.6, rmin=0.85, rmax=2.9, dtype=np.float64)
    for name in ("finnis_sinclair", "mixed"):
        x = htf.Nlist(torch.from_numpy(nl))
        f = htf.compute_nlist_forces(x, _row_models(htf, x)[name])
        xx = torch.from_numpy(nl).requires_grad_(True)
        t = xx[:, :, :3] + 1e-7
        rr = torch.sqrt((t * t).sum(dim=2))
        ss = torch.where(rr > 3e-6, 1.0 / (rr + 3e-6), torch.zeros_like(rr))
        rho = (torch.exp(-1.7 * rr) * ss * ss).sum(dim=1)
        n = ((ss > 0).double() * torch.sigmoid(6.0 * (1.5 - rr))).sum(dim=1)
        en = -1.3 * torch.sqrt(rho) + (0.5 * (ss ** 12 - ss ** 6)).sum(dim=1) if name == "finnis_sinclair" else rho * n
        (g,) = torch.autograd.grad(en.sum(), xx)
        np.testing.assert_allclose(f[:, :3].detach().numpy(), 2.0 * g.sum(dim=1)[:, :3].numpy(), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(f[:, 3].detach().numpy(), en.detach().numpy(), rtol=1e-12)


def test_unit_with_a_row_function_cross_compiles():
    import hoomd_tf_amd as htf
    from hoomd_tf_amd import codegen as cg
    x = htf.Nlist(torch.zeros((2, 4, 4), dtype=torch.float64))
    g = _row_models(htf, x)["coordination"].groups()[0]
    assert g.lowers()
    image, key = cg.compile_body(g.body())
    assert b"htf_jit_rows2_f32_store" in image and b"htf_jit_tails4_f32_store" in image and b"htf_jit_eval_f64_virial" in image
    assert b"htf_jit_train_f32" not in image


def test_row_expression_arithmetic_corner_cases(monkeypatch):
    """What a row expression does when it meets things that are not scalars: a per-particle tensor, a comparison, a torch
    function, a total over particles -- its torch value from there on (the reference's graph is TF ops either way); scalar
    multiples of one sum are ONE sum; a weight that comes first in a product stays symbolic."""
    import hoomd_tf_amd as htf
    from hoomd_tf_amd.simmodel import RowExpr
    rng = np.random.default_rng(9)
    nl, _ = random_nlist(rng, 12, 8, fill=0.6, rmin=0.85, rmax=2.9, dtype=np.float64)
    x = htf.Nlist(torch.from_numpy(nl))
    s = htf.nlist_rinv(x)
    u = htf.reduce_sum(2.0 * (s ** 12 - s ** 6), axis=1)
    e = u + 0.02 * u * u - (0.5 * u) ** 3 / 7.0
    assert isinstance(e, RowExpr) and len(e.sums) == 1 and len(e.groups()) == 1            # one sum, whatever its prefactors
    tt = torch.from_numpy(nl)[:, :, :3] + 1e-7
    rr = torch.sqrt((tt * tt).sum(dim=2))
    ss = torch.where(rr > 3e-6, 1.0 / (rr + 3e-6), torch.zeros_like(rr))
    uu = (2.0 * (ss ** 12 - ss ** 6)).sum(dim=1)
    np.testing.assert_allclose(e.torch_value(torch.from_numpy(nl)).numpy(), (uu + 0.02 * uu * uu - (0.5 * uu) ** 3 / 7.0).numpy(), rtol=1e-12)
    per_particle = torch.arange(12, dtype=torch.float64)
    assert isinstance(e * per_particle, torch.Tensor) and isinstance(per_particle * e, torch.Tensor)
    np.testing.assert_allclose((per_particle * e).detach().numpy(), (per_particle * e.ad).detach().numpy())
    assert (e > 0.0).dtype == torch.bool and isinstance(torch.exp(e), torch.Tensor) and isinstance(e ** per_particle, torch.Tensor)
    assert htf.reduce_sum(e).dim() == 0                                                       # a total: a torch scalar
    w = torch.nn.Parameter(torch.tensor(0.7, dtype=torch.float64))
    assert isinstance(w * e, RowExpr) and isinstance(e / w, RowExpr) and isinstance(w - e, RowExpr)
    np.testing.assert_allclose((w - e).ad.detach().numpy(), (0.7 - e.ad).detach().numpy(), rtol=1e-12)
    # forces of the whole thing on the torch route (CPU: no kernels) == autograd of the torch-written energy
    monkeypatch.setenv("HTF_NO_JIT", "1")
    f = htf.compute_nlist_forces(x, e)
    xx = torch.from_numpy(nl).requires_grad_(True)
    t2 = xx[:, :, :3] + 1e-7
    r2 = torch.sqrt((t2 * t2).sum(dim=2))
    s2 = torch.where(r2 > 3e-6, 1.0 / (r2 + 3e-6), torch.zeros_like(r2))
    u2 = (2.0 * (s2 ** 12 - s2 ** 6)).sum(dim=1)
    (g,) = torch.autograd.grad((u2 + 0.02 * u2 * u2 - (0.5 * u2) ** 3 / 7.0).sum(), xx)
    np.testing.assert_allclose(f[:, :3].detach().numpy(), 2.0 * g.sum(dim=1)[:, :3].numpy(), rtol=1e-9, atol=1e-10)
