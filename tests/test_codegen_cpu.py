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
    }


HARNESS = r"""
#include <math.h>
#include <stdbool.h>
#include <stdio.h>
static float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static float __builtin_amdgcn_sqrtf(float x) { return sqrtf(x); }
static float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
static float __builtin_amdgcn_logf(float x) { return log2f(x); }
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


def test_generated_unit_cross_compiles_for_gfx950():
    import hoomd_tf_amd as htf
    from hoomd_tf_amd import codegen as cg
    x = htf.Nlist(torch.zeros((2, 4, 4)))
    e = _models(htf, x)["yukawa"]
    image, key = cg.compile_body(e.body())
    assert (image[:4] == b"\x7fELF" or image.startswith(b"__CLANG_OFFLOAD_BUNDLE__")) and len(key) == 24
    for name in ("htf_jit_rows2_f32_store", "htf_jit_rows2_f64_nostore", "htf_jit_row1v_f32_store", "htf_jit_eval_f32", "htf_jit_eval_f64_virial"):
        assert name.encode() in image
    image2, key2 = cg.compile_body(e.body())
    assert key2 == key and image2 == image                     # from the cache
