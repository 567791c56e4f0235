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


@pytest.mark.parametrize("name", ["morse", "yukawa", "switched_lj", "mix", "real_power", "ewald_real", "switches", "friedel"])
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


def _typed_inputs(rng, N, NN, ntypes, dtype):
    nl, _ = random_nlist(rng, N, NN, fill=0.7, rmin=0.85, rmax=3.0, dtype=dtype)
    live = nl[:, :, :3].any(axis=2)
    nl[:, :, 3] = np.where(live, rng.integers(0, ntypes, live.shape), 0)
    pos = np.zeros((N, 4), dtype=dtype)
    pos[:, :3] = rng.standard_normal((N, 3))
    pos[:, 3] = rng.integers(0, ntypes, N)
    return nl, pos


@pytest.mark.parametrize("name", ["lj_table", "unlike_only", "neighbor_species", "wide_table"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_typed_evaluator_matches_autograd(htf, cuda, name, dtype):
    """Per-species-pair parameters in a generated kernel: tables looked up by (positions[i, 3], nlist[i, j, 3]) -- the streaming
    evaluator (htf_eval_forces_typed: the positions tensor beside the pair vectors) against torch-fp64 autograd of the same
    expression; an evaluator call WITHOUT the positions is refused for an energy that reads the row particle's type."""
    from test_codegen_cpu import _typed_models
    from test_gpu_parity import CONTACTS, assert_forces_close
    from hoomd_tf_amd.simmodel import PositionsInput
    rng = np.random.default_rng(11)
    ntypes = 3
    nl, pos = _typed_inputs(rng, 300, 128, ntypes, dtype)
    nl64, pos64 = nl.astype(np.float32).astype(np.float64), pos.astype(np.float64)
    e = _typed_models(htf, htf.Nlist(torch.from_numpy(nl64)), PositionsInput.wrap(torch.from_numpy(pos64)), ntypes, big=True)[name]
    pot = e.potential()
    x, P = torch.from_numpy(nl).to(cuda), torch.from_numpy(pos).to(cuda)
    f, v = htf.ops.eval_forces(pot, x, virial=True, positions=P)
    if e.reads_own_type:
        with pytest.raises(ValueError, match="own type"):
            htf.ops.eval_forces(pot, x)
    else:
        assert torch.equal(htf.ops.eval_forces(pot, x), f)
    ref, vref = _ref(htf, e, nl64, virial=True)
    xx = htf.Nlist(torch.from_numpy(nl64))
    (g,) = torch.autograd.grad(e.torch_value(xx.ad).sum(), xx.ad)
    cond = np.abs(2 * g.numpy()[:, :, :3]).sum(axis=(1, 2))
    assert_forces_close("jit_%s_%s" % (name, dtype.__name__), f.cpu().numpy(), ref, cond, cancelling_rows=CONTACTS)
    vcond = (np.linalg.norm(2 * g.numpy()[:, :, :3], axis=2) * np.linalg.norm(nl64[:, :, :3], axis=2) / 2).sum(axis=1)
    assert_forces_close("jit_%s_virial_%s" % (name, dtype.__name__), v.cpu().numpy().reshape(len(nl), 9), vref.reshape(len(nl), 9), vcond,
                        cancelling_rows=CONTACTS)


@pytest.mark.parametrize("wire", [torch.float32, torch.float64])
def test_typed_model_is_replayed_as_the_one_kernel_step(htf, cuda, wire, monkeypatch):
    """A three-species mixture whose compute() looks epsilon and sigma up by species pair (tf.gather on ti * ntypes + tj, the
    way a multi-component model is written against the reference): traced, lowered, replayed as the one-kernel step -- the row
    particle's type from pos.w, the neighbor's from the gathered position -- equal to the torch-autograd route over 30 steps,
    and different from the same run with the types ignored."""
    from hoomd_tf_amd import _lib, standin
    ntypes = 3
    rng = np.random.default_rng(7)
    eps = rng.uniform(0.6, 1.4, (ntypes, ntypes))
    eps = 0.5 * (eps + eps.T)
    sig = rng.uniform(0.85, 1.0, (ntypes, ntypes))
    sig = 0.5 * (sig + sig.T)

    class Mixture(htf.SimModel):
        def compute(self, nlist, positions, box):
            s = htf.nlist_rinv(nlist)
            tj = htf.cast(nlist[:, :, 3], torch.int32)
            ti = htf.cast(positions[:, 3], torch.int32)
            idx = ti[:, None] * ntypes + tj
            q = (htf.gather(sig.reshape(-1), idx) * s) ** 6
            e = 2.0 * htf.gather(eps.reshape(-1), idx) * (q * q - q)
            return htf.compute_nlist_forces(nlist, htf.reduce_sum(e, axis=1))

    def run(jit, typed=True):
        monkeypatch.setenv("HTF_NO_JIT", "0" if jit else "1")
        pos, L, a = standin.fcc_positions(6, 0.8442)
        r = np.random.default_rng(2)
        pos = pos + 0.03 * a * r.standard_normal(pos.shape)
        pos -= np.round(pos / L) * L
        types = r.integers(0, ntypes, len(pos)) if typed else np.zeros(len(pos), dtype=np.int64)
        sysm = standin.System(pos, L, types=types, dtype=wire, device=cuda)
        sysm.randomize_velocities(kT=0.5, seed=2)
        sim = standin.Simulation(sysm)
        sim.integrate_nve(0.002)
        tfc = htf.tfcompute(Mixture(96))
        tfc.attach(sim.nlist_cell(r_buff=0.4, check_period=1), r_cut=2.5)
        sim.run(30, graph=False)
        torch.cuda.synchronize()
        return tfc, sysm.pos.clone(), tfc.force.clone()

    tfc, p1, f1 = run(True)
    assert tfc._plan is not None and tfc._plan.kind == _lib.POT_JIT and tfc.graph_safe()
    tfc0, p0, f0 = run(False)
    assert tfc0._plan is None
    scale = float(f0[:, :3].abs().max())
    assert float((p1[:, :3] - p0[:, :3]).abs().max()) < 2e-4
    assert float((f1[:, :3] - f0[:, :3]).abs().max()) < 2e-3 * scale
    assert abs(float(f1[:, 3].double().sum()) - float(f0[:, 3].double().sum())) < 1e-4 * abs(float(f0[:, 3].double().sum())) + 1e-3
    _, _, f2 = run(True, typed=False)
    assert float((f2[:, :3] - f1[:, :3]).abs().max()) > 0.05 * scale      # (the species matter: one-species forces are elsewhere)


def test_a_model_with_weights_runs_generated_kernels_until_a_weight_is_written(htf, cuda, monkeypatch):
    """Inference MD with a traced model that owns torch Parameters (what every Keras layer upstream does): the weights' present
    values are constants of the generated kernel and the model is replayed as the one-kernel step; writing a weight in place
    (load_weights, an optimizer step) makes the plan stale -- the next step re-traces with the new value, a new kernel -- and the
    forces follow it: equal to the torch-autograd route with the same weights, before and after."""
    from hoomd_tf_amd import _lib, standin

    class Morse(htf.SimModel):
        def setup(self):
            self.depth = torch.nn.Parameter(torch.tensor(0.8))
            self.width = torch.nn.Parameter(torch.tensor(4.0))

        def compute(self, nlist, positions, box):
            r = htf.safe_norm(nlist[:, :, :3], axis=2)
            live = htf.cast(htf.nlist_rinv(nlist) > 0.0, torch.float32)
            x = 1.0 - htf.exp(-1.0 * self.width * (r - 1.122))
            return htf.compute_nlist_forces(nlist, htf.reduce_sum(0.5 * self.depth * live * (x * x - 1.0), axis=1))

    def system():
        pos, L, a = standin.fcc_positions(6, 0.8442)
        rng = np.random.default_rng(2)
        pos = pos + 0.03 * a * rng.standard_normal(pos.shape)
        pos -= np.round(pos / L) * L
        sysm = standin.System(pos, L, dtype=torch.float32, device=cuda)
        sysm.randomize_velocities(kT=0.3, seed=2)
        sim = standin.Simulation(sysm)
        sim.integrate_nve(0.002)
        return sysm, sim

    def run(jit, schedule):
        monkeypatch.setenv("HTF_NO_JIT", "0" if jit else "1")
        sysm, sim = system()
        model = Morse(96)
        tfc = htf.tfcompute(model)
        tfc.attach(sim.nlist_cell(r_buff=0.4, check_period=1), r_cut=2.5)
        out = []
        for depth in schedule:
            with torch.no_grad():
                model.depth.fill_(depth)
            sim.run(10, graph=False)
            torch.cuda.synchronize()
            out.append((tfc._plan, tfc.force.clone(), tfc._plan_is_stale()))
        return tfc, out

    tfc, got = run(True, [0.8, 1.6, 1.6])
    _, ref = run(False, [0.8, 1.6, 1.6])
    plans = [g[0] for g in got]
    assert all(p is not None and p.kind == _lib.POT_JIT for p in plans) and not any(g[2] for g in got)
    assert plans[0] is not plans[1] and plans[1] is plans[2]            # re-traced once, when the weight changed; not again
    assert "0.4" in plans[0].body and "0.8" in plans[1].body               # 0.5 * depth, as a constant of each kernel
    assert len(tfc._plan_folded) >= 2
    for (_, f, _), (p0, f0, _) in zip(got, ref):
        assert p0 is None
        scale = float(f0[:, :3].abs().max())
        assert float((f[:, :3] - f0[:, :3]).abs().max()) < 2e-3 * scale
        assert abs(float(f[:, 3].double().sum()) - float(f0[:, 3].double().sum())) < 1e-4 * abs(float(f0[:, 3].double().sum())) + 1e-3
    assert float((got[1][1][:, 3].double().sum() / got[0][1][:, 3].double().sum())) > 1.5      # (twice the well depth: the energy followed)


def test_random_expressions_on_the_device(htf, cuda):
    """Six random expression trees of the tracer's whole op set (tables by species pair included), masked by s^2: the generated
    streaming evaluator against torch-fp64 autograd of the same tree, forces and energies at the LJ tolerances."""
    from test_codegen_cpu import _random_expression
    from test_gpu_parity import CONTACTS, assert_forces_close
    from hoomd_tf_amd.simmodel import PositionsInput
    rng = np.random.default_rng(77)
    nl, pos = _typed_inputs(rng, 200, 64, 3, np.float32)
    nl64, pos64 = nl.astype(np.float64), pos.astype(np.float64)
    xs = htf.Nlist(torch.from_numpy(nl64))
    P = PositionsInput.wrap(torch.from_numpy(pos64))
    s, r = htf.nlist_rinv(xs), htf.safe_norm(xs[:, :, :3], axis=2)
    x, Pd = torch.from_numpy(nl).to(cuda), torch.from_numpy(pos).to(cuda)
    done = 0
    for trial in range(12):
        e = htf.square(s) * _random_expression(htf, rng, s, r, xs[:, :, 3], P[:, 3], int(rng.integers(1, 4)))
        if not e.lowers():
            continue
        f = htf.ops.eval_forces(e.potential(), x, positions=Pd)
        ref = _ref(htf, e, nl64)
        xx = htf.Nlist(torch.from_numpy(nl64))
        (g,) = torch.autograd.grad(e.torch_value(xx.ad).sum(), xx.ad)
        cond = np.abs(2 * g.numpy()[:, :, :3]).sum(axis=(1, 2))
        assert_forces_close("jit_random_%d" % trial, f.cpu().numpy(), ref, cond, cancelling_rows=CONTACTS)
        done += 1
        if done == 6:
            break
    assert done == 6
