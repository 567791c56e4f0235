"""Generated kernels for traced pair energies (HTF_POT_JIT) on the GPU: the streaming evaluator, the one-kernel step and its
virial form around a generated body, against torch-fp64 autograd of the SAME expression (the reference's own definition of a
model's forces: tf.gradients of whatever compute() builds, simmodel.py:526-555), at the LJ tolerances; through tfcompute the model
is replayed as the one-kernel step like a built-in closed form."""
import os

import numpy as np
import pytest
import torch

from helpers import random_nlist
from test_codegen_cpu import _models

pytestmark = pytest.mark.gpu


def _ref(htf, e, nl64, virial=False):
    """torch fp64 autograd of the traced expression on the same pair vectors."""
    from hoomd_tf_amd.simmodel import _autograd_nlist_forces
    x = htf.Nlist(torch.from_numpy(nl64))
    en = e.torch_value(x.ad).sum(dim=1)
    out = _autograd_nlist_forces(x, en, virial)
    return [o.detach().numpy() for o in out] if virial else out.detach().numpy()


@pytest.mark.parametrize("name", ["morse", "yukawa", "switched_lj", "mix", "real_power"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_generated_evaluator_matches_autograd(htf, cuda, name, dtype):
    from test_gpu_parity import CONTACTS, assert_forces_close
    rng = np.random.default_rng(5)
    nl, _ = random_nlist(rng, 300, 128, fill=0.7, rmin=0.85, rmax=3.0, dtype=dtype)
    nl64 = nl.astype(np.float32).astype(np.float64)
    e = _models(htf, htf.Nlist(torch.from_numpy(nl64)))[name]
    pot = e.potential()
    x = torch.from_numpy(nl).to(cuda)
    f, v = htf.ops.eval_forces(pot, x, virial=True)
    f0 = htf.ops.eval_forces(pot, x)
    assert torch.equal(f, f0)
    ref, vref = _ref(htf, e, nl64, virial=True)
    # condition scale: sum_j |f_ij| of the row, from the reference's per-pair gradient
    xx = htf.Nlist(torch.from_numpy(nl64))
    (g,) = torch.autograd.grad(e.torch_value(xx.ad).sum(), xx.ad)
    cond = np.abs(2 * g.numpy()[:, :, :3]).sum(axis=(1, 2))
    assert_forces_close("jit_%s_%s" % (name, dtype.__name__), f.cpu().numpy(), ref, cond, cancelling_rows=CONTACTS)
    vcond = (np.linalg.norm(2 * g.numpy()[:, :, :3], axis=2) * np.linalg.norm(nl64[:, :, :3], axis=2) / 2).sum(axis=1)
    assert_forces_close("jit_%s_virial_%s" % (name, dtype.__name__), v.cpu().numpy().reshape(len(nl), 9), vref.reshape(len(nl), 9), vcond,
                        cancelling_rows=CONTACTS)


@pytest.mark.parametrize("name,wire", [("morse", torch.float32), ("yukawa", torch.float32), ("switched_lj", torch.float64)])
def test_traced_model_is_replayed_as_the_one_kernel_step(htf, cuda, name, wire, monkeypatch):
    """A SimModel whose compute() is written with htf.* ops the zoo does not know (Morse, Yukawa, a switched LJ): traced on the
    first step, lowered to HTF_POT_JIT, replayed as the fused one-kernel step (tensor written too) -- forces and energies of an
    MD run equal to the same model forced onto the torch-autograd route (HTF_NO_JIT=1) within fp32 rounding of the row sums."""
    from hoomd_tf_amd import _lib, standin

    class Model(htf.SimModel):
        def compute(self, nlist, positions, box):
            e = _models(htf, nlist)[name]
            return htf.compute_nlist_forces(nlist, htf.reduce_sum(e, axis=1))

    def run(jit):
        monkeypatch.setenv("HTF_NO_JIT", "0" if jit else "1")
        pos, L, a = standin.fcc_positions(6, 0.8442)
        rng = np.random.default_rng(2)
        pos = pos + 0.03 * a * rng.standard_normal(pos.shape)
        pos -= np.round(pos / L) * L
        sysm = standin.System(pos, L, dtype=wire, device=cuda)
        sysm.randomize_velocities(kT=0.5, seed=2)
        sim = standin.Simulation(sysm)
        sim.integrate_nve(0.002)
        tfc = htf.tfcompute(Model(96))
        tfc.attach(sim.nlist_cell(r_buff=0.4, check_period=1), r_cut=2.5)
        sim.run(30, graph=False)
        torch.cuda.synchronize()
        return tfc, sysm.pos.clone(), tfc.force.clone()

    tfc, p1, f1 = run(True)
    assert tfc._plan is not None and tfc._plan.kind == _lib.POT_JIT and tfc.graph_safe()
    tfc0, p0, f0 = run(False)
    assert tfc0._plan is None
    scale = float(f0[:, :3].abs().max())
    assert float((p1[:, :3] - p0[:, :3]).abs().max()) < 2e-4          # 30 steps of the same dynamics
    assert float((f1[:, :3] - f0[:, :3]).abs().max()) < 2e-3 * scale
    assert abs(float(f1[:, 3].double().sum()) - float(f0[:, 3].double().sum())) < 1e-4 * abs(float(f0[:, 3].double().sum())) + 1e-3
