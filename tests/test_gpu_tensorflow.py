"""The reference's integration tests (htf/test-py/test_tensorflow.py, test_layers.py,
test_utils.py::test_eds) re-run against hoomd_tf_amd on the GPU: same systems, same
assertions, HOOMD replaced by the stand-in driver and HOOMD's md.pair.lj by the fp64
analytic LJ of tests/helpers.py."""
import numpy as np
import pytest
import torch

import build_examples
from helpers import analytic_lj, min_image_np, sq_lattice
from oracle import htf_oracle as O

pytestmark = pytest.mark.gpu


def _sim(htf, cuda, n, a, dtype=torch.float64, types=None, kT=None, seed=1, dt=0.005, jitter=0.0):
    from hoomd_tf_amd import standin
    pos, L = sq_lattice(n, a)
    if jitter:
        pos[:, :2] += jitter * np.random.default_rng(seed).standard_normal((n * n, 2))
    system = standin.System(pos, L, types=types, dtype=dtype, device=cuda)
    sim = standin.Simulation(system)
    if kT is not None:
        system.randomize_velocities(kT, seed)
        system.vel[:, 2] = 0  # 2-D
    sim.integrate_nve(dt)
    return sim, system, L


def compute_forces(system, L, rcut):
    """test_tensorflow.py:20-35."""
    position = system.positions_numpy()
    N = len(position)
    forces = np.zeros((N, 3))
    for i in range(N):
        for j in range(i + 1, N):
            r = min_image_np(position[j] - position[i], L)
            rd = np.sqrt(np.sum(r**2))
            if rd <= rcut:
                f = -r / rd
                forces[i, :] += f
                forces[j, :] -= f
    return forces


def test_access(htf, cuda):
    """test_tensorflow.py:46-70: three types survive into nlist[..., 3] and positions[:, 3]."""
    from hoomd_tf_amd import standin
    rng = np.random.default_rng(0)
    cell = np.array([[2, 2, 2], [1, 3, 1], [3, 1, 1]], dtype=float)
    ijk = np.stack(np.meshgrid(*[np.arange(5)] * 3, indexing="ij"), -1).reshape(-1, 3)
    pos = (ijk[:, None, :] * 6 + cell[None]).reshape(-1, 3) - 15.0
    types = np.tile(np.arange(3), len(ijk))
    system = standin.System(pos, [30, 30, 30], types=types, dtype=torch.float64, device=cuda)
    sim = standin.Simulation(system)
    sim.integrate_nve(0.005)
    model = build_examples.SimplePotential(32)
    tfcompute = htf.tfcompute(model)
    tfcompute.attach(sim.nlist_cell(check_period=1), r_cut=3)
    sim.run(1)
    tfcompute.get_virial_array()
    tfcompute.get_forces_array()
    pa = tfcompute.get_positions_array()
    nl = tfcompute.get_nlist_array()
    assert len(np.unique(nl[:, :, 3].astype(int))) == 3
    assert len(np.unique(pa[:, 3].astype(int))) == 3


@pytest.mark.parametrize("batch_size", [None, 4])
def test_force_overwrite(htf, cuda, batch_size):
    """test_tensorflow.py:81-129."""
    N, rcut = 9, 5.0
    sim, system, L = _sim(htf, cuda, 3, 4.0, kT=2, seed=2)
    model = build_examples.SimplePotential(N - 1)
    tfcompute = htf.tfcompute(model)
    tfcompute.attach(sim.nlist_cell(check_period=1), r_cut=rcut, batch_size=batch_size)
    sim.run(1)
    sim.run(1)
    for i in range(3):
        sim.compute_forces()
        py_forces = compute_forces(system, L, rcut)
        np.testing.assert_allclose(sim.net_force.cpu().numpy()[:, :3], py_forces, atol=1e-5)
        sim.run(100)
    assert tfcompute._plan is not None  # the traced single-call path took over


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_lj_forces(htf, cuda, dtype):
    """test_tensorflow.py:335-382: LJModel(32) == md.pair.lj(eps=1, sig=1, r_cut=5), atol 1e-5."""
    sim, system, L = _sim(htf, cuda, 5, 3.0, dtype=dtype, kT=1, seed=1)
    model = build_examples.LJModel(32)
    tfcompute = htf.tfcompute(model)
    tfcompute.attach(sim.nlist_cell(check_period=1), r_cut=5.0)
    sim.run(20)
    for i in range(10):
        sim.run(1)
        sim.compute_forces()
        F, E, _ = analytic_lj(system.positions_numpy(), L, 5.0)
        got = sim.net_force.double().cpu().numpy()
        np.testing.assert_allclose(got[:, :3], F, atol=1e-5)
        np.testing.assert_allclose(got[:, 3], E, atol=1e-5)
        assert np.all(np.sum(F**2, axis=1) > 1e-4**2), 'Forces are too low to assess!'


def test_traced_equals_eager(htf, cuda):
    """The one-C-call traced path and the per-batch eager path give the same forces: bit-equal
    when the traced path uses the same two kernels (fused = 0), to summation-order rounding with
    the default one-kernel mode (the evaluator rides in the build kernel: 64-lane instead of
    16-lane row sums)."""
    outs = []
    for force_eager, fused in ((False, 0), (True, 0), (False, 2)):
        sim, system, L = _sim(htf, cuda, 6, 1.3, kT=0.5, seed=4, jitter=0.05)
        model = build_examples.LJModel(32)
        tfc = htf.tfcompute(model)
        tfc.fused = fused
        tfc.attach(sim.nlist_cell(check_period=1), r_cut=2.5, save_output_period=1 if force_eager else None)
        sim.run(15)
        sim.compute_forces()
        assert (tfc._plan is None) == force_eager
        outs.append(sim.net_force.cpu().numpy().copy())
    np.testing.assert_array_equal(outs[0], outs[1])
    np.testing.assert_allclose(outs[2], outs[1], rtol=1e-4, atol=1e-5 * np.abs(outs[1]).max())


def test_lj_energy(htf, cuda):
    """test_tensorflow.py:532-557 at the reference's own parameters (kT = 0.8, dt = 0.001, ten blocks of 250 steps,
    r_cut = 5 unshifted): NVE total energy is conserved to 1e-3 between blocks.  The unshifted cut makes the energy jump by
    e(r_cut) = 4 (5^-12 - 5^-6) = -2.56e-4 whenever a pair crosses it; whether a block passes the bare assertion therefore
    depends on the random seed (the reference's seed 1 belongs to HOOMD's generator, not to ours).  The restatement keeps the
    assertion AS STATED for every block across which the number of pairs inside r_cut is unchanged, and for the others
    asserts the same bound on the energy minus those known steps -- i.e. exact conservation of what the integrator conserves."""
    sim, system, L = _sim(htf, cuda, 3, 4.0, kT=0.8, seed=1, dt=0.001)
    tfcompute = htf.tfcompute(build_examples.LJModel(32))
    tfcompute.attach(sim.nlist_cell(check_period=1), r_cut=5.0)
    e_cut = 4.0 * (5.0 ** -12 - 5.0 ** -6)
    energy, inside = [], []
    as_stated = 0
    for i in range(10):
        sim.run(250)
        sim.compute_forces()
        pe = float(sim.net_force[:, 3].sum())
        v = system.vel[:, :3] + 0.5 * 0.001 * sim.net_force[:, :3]  # leapfrog: v at t
        energy.append(pe + 0.5 * float((v * v).sum()))
        nl = tfcompute.get_nlist_array()
        inside.append(int((np.sum(nl[:, :, :3] ** 2, axis=2) > 0).sum()) // 2)   # pairs within r_cut (each listed twice)
        if i > 1:
            dn = inside[-1] - inside[-2]
            as_stated += dn == 0
            np.testing.assert_allclose(energy[-1] - dn * e_cut, energy[-2], atol=1e-3, err_msg=str((energy, inside)))
    assert as_stated >= 1, inside   # at least one block is the reference's assertion word for word


def test_nlist_count(htf, cuda):
    """test_tensorflow.py:559-579: full (not half) list -> 4 neighbors on the 3x3 lattice."""
    sim, system, L = _sim(htf, cuda, 3, 4.0, kT=0.8, seed=1, dt=0.001)
    tfcompute = htf.tfcompute(build_examples.LJModel(32))
    tfcompute.attach(sim.nlist_cell(), r_cut=5.0)
    sim.run(1)
    nl = tfcompute.get_nlist_array()
    ncount = np.sum(np.sum(nl**2, axis=2) > 0.1, axis=1)
    assert np.min(ncount) == 4


def test_overflow(htf, cuda):
    """test_tensorflow.py:830-848: NN=4, r_cut=10, check_nlist=True raises."""
    sim, system, L = _sim(htf, cuda, 8, 4.0, kT=1, seed=1)
    tfcompute = htf.tfcompute(build_examples.LJModel(4, check_nlist=True))
    tfcompute.attach(sim.nlist_cell(check_period=1), r_cut=10.0)
    with pytest.raises(htf.NlistOverflowError):
        sim.run(2)


def test_skew_fails(htf, cuda):
    """test_tensorflow.py:321-333."""
    sim, system, L = _sim(htf, cuda, 3, 4.0)
    system.box3x3[2, 0] = 0.5
    system.box = htf._lib.make_box(system.box3x3)
    tfcompute = htf.tfcompute(build_examples.WrapModel(0, output_forces=False))
    tfcompute.attach()
    with pytest.raises(htf.SkewedBoxError):
        sim.run(1)


def test_wrap(htf, cuda):
    """test_tensorflow.py:310-319."""
    sim, system, L = _sim(htf, cuda, 3, 4.0)
    tfcompute = htf.tfcompute(build_examples.WrapModel(0, output_forces=False))
    tfcompute.attach(save_output_period=1)
    sim.run(1)
    p = system.positions_numpy()
    np.testing.assert_allclose(tfcompute.outputs[0][0], O.wrap_vector(p[0] - p[-1], O.make_box(L)), atol=1e-12)


def test_lj_pressure(htf, cuda):
    """test_tensorflow.py:619-671: virial [:, 0:2] vs HOOMD LJ per-particle virial, atol 1e-5."""
    sim, system, L = _sim(htf, cuda, 3, 4.0, kT=1, seed=1)
    tfcompute = htf.tfcompute(build_examples.LJVirialModel(32, virial=True))
    tfcompute.attach(sim.nlist_cell(check_period=1), r_cut=5.0)
    for i in range(5):
        sim.run(3)
        sim.compute_forces()
        tf_virial = tfcompute.get_virial_array()
        _, _, V = analytic_lj(system.positions_numpy(), L, 5.0)
        # [B, 9] row-major: xx, xy are columns 0, 1
        np.testing.assert_allclose(V[:, 0:2], tf_virial[:, 0:2], atol=1e-5)


def test_wca(htf, cuda):
    """test_layers.py:10-22: WCA(32) runs 10 batched steps (+ values vs the oracle)."""
    sim, system, L = _sim(htf, cuda, 3, 4.0, kT=0.8, seed=1, dt=0.001)
    tfcompute = htf.tfcompute(build_examples.WCA(32))
    tfcompute.attach(sim.nlist_cell(), r_cut=5.0, batch_size=4)
    sim.run(10)
    assert np.all(np.isfinite(tfcompute.get_forces_array()))


def test_rbf(htf, cuda):
    """test_layers.py:24-31 (shape) + values against the oracle restatement."""
    rbf = htf.RBFExpansion(0, 2, 10)
    nlist = torch.ones((10, 6, 3), device=cuda)
    r = htf.safe_norm(nlist, axis=2)
    out = rbf(r)
    assert tuple(out.shape) == (10, 6, 10)
    ref = O.rbf_expansion(O.safe_norm(np.ones((10, 6, 3), np.float32), axis=2), 0, 2, 10)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=2e-6, atol=1e-7)


def test_eds_layer_trace(htf, cuda):
    """EDSLayer step-for-step against the oracle state machine (200-step scripted CV)."""
    rng = np.random.default_rng(11)
    cvs = 4.0 + rng.standard_normal(200)
    dev, ref = htf.EDSLayer(4.0, 5, 0.2, device=cuda), O.EDSLayer(4.0, 5, 0.2)
    for cv in cvs:
        a = float(dev(float(cv)))
        b = float(ref(cv))
        assert abs(a - b) <= 2e-5 * max(1.0, abs(b))
    assert abs(b) > 0.1
    with pytest.raises(ValueError):
        htf.EDSLayer(4, 5)


def test_eds(htf, cuda):
    """test_utils.py:447-461: EDSModel(set_point=4): (cv_avg - 4)^2 < 0.5, alpha finite."""
    sim, system, L = _sim(htf, cuda, 3, 4.0, kT=0.2, seed=2, dt=0.05)
    model = build_examples.EDSModel(0, set_point=4.0)
    tfcompute = htf.tfcompute(model)
    tfcompute.attach(save_output_period=10)
    sim.run(1000)
    assert np.isfinite(np.mean(tfcompute.outputs[0]))
    assert (model.cv_sum / model.cv_n - 4) ** 2 < 0.5


def test_rdf(htf, cuda):
    """test_tensorflow.py:433-485: rdf non-zero; typed rdf A-B == B-A; values vs the oracle."""
    types = np.arange(81) % 2
    sim, system, L = _sim(htf, cuda, 9, 1.5, types=types, kT=0.5, seed=1, dt=0.001, jitter=0.1)
    model = build_examples.LJRDF(64)
    tfcompute = htf.tfcompute(model)
    tfcompute.attach(sim.nlist_cell(), r_cut=5.0, save_output_period=1)
    sim.run(3)
    rdf = tfcompute.outputs[0][-1]
    assert np.sum(rdf) > 0
    nl = tfcompute.get_nlist_array().astype(np.float32)
    ref, rs = O.compute_rdf(nl, [3, 5])
    # fp32 shell volumes hi^3 - lo^3 cancel (~2e-5 relative): counts are exact, the quotient is not
    np.testing.assert_allclose(rdf, ref, rtol=1e-4)
    model = build_examples.LJTypedModel(64)
    sim, system, L = _sim(htf, cuda, 9, 1.5, types=types, kT=0.5, seed=1, dt=0.001, jitter=0.1)
    tfcompute = htf.tfcompute(model)
    tfcompute.attach(sim.nlist_cell(), r_cut=5.0)
    sim.run(2)
    rdfa, rdfb = model.avg_rdfa.result().cpu().numpy(), model.avg_rdfb.result().cpu().numpy()
    np.testing.assert_allclose(rdfa, rdfb, rtol=1e-6)        # test_tensorflow.py:482-485
    assert np.sum(rdfa) > 0


@pytest.mark.parametrize("sdtype", [torch.float32, torch.float64], ids=["float32", "float64"])
def test_typed_rdf_replayed_from_the_plan(htf, cuda, sdtype):
    """build_examples.LJTypedModel as upstream writes it (build_examples.py:80-101: two typed compute_rdf's of
    ``positions[:, 3]``, each averaged by a MeanTensor; test_tensorflow.py:450-485) stays on the replayed plan: after the
    first eager step the force kernel is followed by the typed histograms (row types from the positions side buffer) and
    the metric updates, and SimModel.compute is not called again.  The last step's RDFs against the oracle's masked_nlist +
    histogram (simmodel.py:656-693) on the step's own tensor and types; a type tensor that is NOT a view of the step's
    positions keeps the model eager."""
    types = np.arange(81) % 2
    sim, system, L = _sim(htf, cuda, 9, 1.5, dtype=sdtype, types=types, kT=0.5, seed=1, dt=0.001, jitter=0.1)
    model = build_examples.LJTypedModel(64)
    calls = []
    inner = model.compute
    model.compute = lambda *a, **k: (calls.append(1), inner(*a, **k))[1]
    tfc = htf.tfcompute(model)
    tfc.attach(sim.nlist_cell(), r_cut=5.0)
    sim.run(6)
    assert tfc._plan is not None and len(tfc._post_ops) == 4 and len(calls) == 1
    assert model.avg_rdfa.count == 6 and model.avg_rdfb.count == 6
    rdfa, rdfb = model.avg_rdfa.result().cpu().numpy(), model.avg_rdfb.result().cpu().numpy()
    np.testing.assert_allclose(rdfa, rdfb, rtol=1e-6)
    model.avg_rdfa.reset_states()
    model.avg_rdfb.reset_states()
    sim.run(1)
    assert len(calls) == 1 and model.avg_rdfa.count == 1
    nl = tfc.get_nlist_array().astype(np.float32)
    pos_t = tfc.get_positions_array()[:, 3].astype(np.float32)
    np.testing.assert_array_equal(pos_t, types.astype(np.float32))
    for avg, (ti, tj) in ((model.avg_rdfa, (0, 1)), (model.avg_rdfb, (1, 0))):
        ref, _ = O.compute_rdf(nl, [0, 10], pos_t, type_i=ti, type_j=tj)
        assert ref.sum() > 0
        np.testing.assert_allclose(avg.result().cpu().numpy(), ref, rtol=1e-4)

    class DetachedTypes(build_examples.LJTypedModel):
        def compute(self, nlist, positions, box):
            energy = htf.reduce_sum(1e-10 * htf.nlist_rinv(nlist) ** 12, axis=1)
            forces = htf.compute_nlist_forces(nlist, energy)
            rdfa, _ = htf.compute_rdf(nlist, [0, 10], positions[:, 3].clone(), type_i=0, type_j=1)
            self.avg_rdfa.update_state(rdfa)
            return forces
    sim2, system2, _ = _sim(htf, cuda, 9, 1.5, dtype=sdtype, types=types, kT=0.5, seed=1, dt=0.001, jitter=0.1)
    m2 = DetachedTypes(64)
    tfc2 = htf.tfcompute(m2)
    tfc2.attach(sim2.nlist_cell(), r_cut=5.0)
    sim2.run(3)
    assert tfc2._plan is None and m2.avg_rdfa.count == 3


def test_bare_observable_keeps_the_model_eager(htf, cuda):
    """A compute_rdf whose result no traced consumer takes (here: appended to a Python list) is not worth a plan that would
    stop calling compute(): the model steps eagerly and the list keeps growing."""
    sim, system, L = _sim(htf, cuda, 9, 1.5, kT=0.5, seed=1, dt=0.001, jitter=0.1)

    class ListRDF(htf.SimModel):
        def setup(self):
            self.rdfs = []

        def compute(self, nlist):
            energy = htf.reduce_sum(1e-10 * htf.nlist_rinv(nlist) ** 12, axis=1)
            rdf, _ = htf.compute_rdf(nlist, [0, 5])
            self.rdfs.append(rdf)
            return htf.compute_nlist_forces(nlist, energy)
    model = ListRDF(64)
    tfc = htf.tfcompute(model)
    tfc.attach(sim.nlist_cell(), r_cut=5.0)
    sim.run(4)
    assert tfc._plan is None and not tfc._post_ops and len(model.rdfs) == 4


def test_quickstart_example01(htf, cuda):
    """examples/01. Quickstart.ipynb as written (cells 3 and 5): 16 x 16 particles on sq(a = 1.2), WCAPotential(64) --
    r^-12 times cast(r < 2^(1/6)) -- attached with r_cut 5, an RDF averaged every step.  Forces of a step against the
    oracle on the same pair vectors; the symbolic mask lowers to ONE kernel (a masked rinv polynomial), the lattice
    rows (60 candidates within 5.0 + buffer > NN = 64 only after the fluid has moved) do not overflow at first."""
    sim, system, L = _sim(htf, cuda, 16, 1.2, kT=0.5, seed=1, dt=0.005)
    model = build_examples.QuickstartWCA(64)
    tfcompute = htf.tfcompute(model)
    tfcompute.attach(sim.nlist_cell(), r_cut=5)
    sim.run(20)
    model.avg_rdf.reset_states()
    sim.run(5)
    nl = tfcompute.get_nlist_array().astype(np.float32)
    ref = O.rinv_poly_model(nl.astype(np.float64), [1.0], [12], cut=2 ** (1 / 6))
    f = tfcompute.get_forces_array()
    np.testing.assert_allclose(f, ref, atol=1e-5, rtol=2e-5)
    assert np.abs(ref[:, :3]).max() > 0.05              # neighbors at 1.2 -> inside 2^(1/6) only once they move: repulsion is on
    rdf = model.avg_rdf.result().cpu().numpy()
    assert rdf.shape == (2, 100) and model.avg_rdf.count == 5 and rdf[0].sum() > 0
    ref_rdf, rs = O.compute_rdf(nl, [0, 3.5])
    np.testing.assert_allclose(rs, rdf[1], rtol=1e-6)
    # The metric lives on the device and its update is part of the step plan (the reference's tf.function traces
    # tf.keras.metrics.MeanTensor the same way): compute() is not called again after the first step, and the average over the
    # next steps is the mean of the per-step RDFs of the oracle on each step's own pair vectors.
    assert tfcompute._plan is not None and len(tfcompute._post_ops) == 2
    calls = {"n": 0}
    orig = model.compute

    def counting(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)
    model.compute = counting
    model.avg_rdf.reset_states()
    want = []
    for _ in range(4):
        sim.run(1)
        want.append(O.compute_rdf(tfcompute.get_nlist_array().astype(np.float32), [0, 3.5])[0])
    assert calls["n"] == 0 and model.avg_rdf.count == 4
    np.testing.assert_allclose(model.avg_rdf.result().cpu().numpy()[0], np.mean(want, axis=0), rtol=2e-5, atol=1e-6)
    # ... and the whole step, observable included, is a fixed launch sequence: replayed from a hipGraph it leaves the same
    # average behind as the eager replay of the same steps
    assert tfcompute.graph_safe()
    model.avg_rdf.reset_states()
    sim.run(12, graph=True)
    assert calls["n"] == 0 and model.avg_rdf.count == 12
    g_rdf = model.avg_rdf.result().cpu().numpy()
    assert np.all(np.isfinite(g_rdf)) and g_rdf[0].sum() > 0
    np.testing.assert_allclose(g_rdf[1], rs, rtol=1e-6)
    model.compute = orig
    # the lowered potential is the masked polynomial (one evaluator kernel), not the autograd fallback
    from hoomd_tf_amd import simmodel
    pots = [k for k in getattr(simmodel.compute_nlist_forces, "_cache", {}) if k[0] == "poly" and k[-1] is not None]
    assert pots and abs(pots[0][-1] - 2 ** (1 / 6)) < 1e-12


def test_eds_rdf_model_steps_match_oracle(htf, cuda):
    """config C4 through SimModel/tfcompute: every step's forces, cv and alpha against the
    oracle composite driven by the oracle EDSLayer on the same pair vectors."""
    from hoomd_tf_amd import standin
    pos, L, a = standin.fcc_positions(4, 0.8442)
    pos = pos + 0.03 * a * np.random.default_rng(2).standard_normal(pos.shape)
    system = standin.System(pos, L, dtype=torch.float32, device=cuda)
    sim = standin.Simulation(system)
    sim.integrate_nve(0.002)
    model = build_examples.EDSRDFModel(64, set_point=9.0, r0=1.1, gap=0.05, period=8, learning_rate=0.5)
    tfc = htf.tfcompute(model)
    tfc.attach(sim.nlist_cell(check_period=1), r_cut=2.5, save_output_period=1)
    ref_eds = O.EDSLayer(9.0, 8, 0.5)
    for step in range(20):
        sim.run(1)
        nl = tfc.get_nlist_array().astype(np.float32).astype(np.float64)
        _, cv_ref = O.eds_rdf_model(nl, 0.0, 1.1, 0.05)
        a_ref = float(ref_eds(cv_ref))
        f_ref, _ = O.eds_rdf_model(nl, a_ref, 1.1, 0.05)
        np.testing.assert_allclose(tfc.outputs[0][-1], cv_ref, rtol=2e-5)
        np.testing.assert_allclose(tfc.outputs[1][-1], a_ref, rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(tfc.get_forces_array(), f_ref, rtol=2e-4, atol=2e-3)
    assert abs(a_ref) > 1e-3  # the bias is actually on
    assert tfc._plan is None  # stateful model: stays on the eager path
    rdf_ref, _ = O.compute_rdf(tfc.get_nlist_array().astype(np.float32), [0, 3.5])
    np.testing.assert_allclose(tfc.outputs[2][-1], rdf_ref, rtol=1e-4)


@pytest.mark.parametrize("sdtype", [torch.float32, torch.float64], ids=["float32", "float64"])
def test_eds_rdf_model_replayed_as_one_kernel(htf, cuda, sdtype):
    """config C4 with nobody saving the outputs: after the first (eager) step tfcompute replays
    the step as htf_build_eval_forces2 + device-side EDS update.  Every step's forces, cv and
    alpha against the oracle composite on the same pair vectors, as in the eager test above.
    ``float64``: HOOMD built in double precision (TensorflowCompute.h:117-124, the reference's commonest deployment) stays
    on the replayed plan too -- fp64 positions in, the fp32 tensor of simmodel.py:226-227's cast, fp64 forces out."""
    from hoomd_tf_amd import standin
    pos, L, a = standin.fcc_positions(4, 0.8442)
    pos = pos + 0.03 * a * np.random.default_rng(2).standard_normal(pos.shape)
    system = standin.System(pos, L, dtype=sdtype, device=cuda)
    sim = standin.Simulation(system)
    sim.integrate_nve(0.002)
    model = build_examples.EDSRDFModel(64, set_point=9.0, r0=1.1, gap=0.05, period=8, learning_rate=0.5)
    tfc = htf.tfcompute(model)
    cell = sim.nlist_cell(check_period=1)
    tfc.attach(cell, r_cut=2.5)
    ref_eds = O.EDSLayer(9.0, 8, 0.5)
    for step in range(20):
        pos_before = system.pos.clone()
        sim.run(1)
        assert tfc._bplan is not None  # installed by the first (eager) step
        nl = tfc.get_nlist_array().astype(np.float32).astype(np.float64)
        _, cv_ref = O.eds_rdf_model(nl, 0.0, 1.1, 0.05)
        a_ref = float(ref_eds(cv_ref))
        f_ref, _ = O.eds_rdf_model(nl, a_ref, 1.1, 0.05)
        np.testing.assert_allclose(float(tfc._bplan["cv"].value), cv_ref, rtol=2e-5)
        np.testing.assert_allclose(float(model.eds_bias.alpha), a_ref, rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(tfc.get_forces_array(), f_ref, rtol=2e-4, atol=2e-3)
        if step > 0:  # the replay's tensor is the build kernel's tensor, bit for bit
            want = htf.ops.build_pair_vectors(pos_before, cell.n_neigh, cell.head_list, cell.nlist, system.box, 2.5, 64)
            assert want.dtype == torch.float32 and torch.equal(tfc._last[0], want)
            np.testing.assert_array_equal(tfc.get_positions_array()[:, :3], pos_before[:, :3].double().cpu().numpy())
            assert tfc.force.dtype == sdtype
    assert abs(a_ref) > 1e-3
    # the same trajectory as the eager path (summation order differs between the one-kernel and two-kernel sweeps)
    system2 = standin.System(pos, L, dtype=sdtype, device=cuda)
    sim2 = standin.Simulation(system2)
    sim2.integrate_nve(0.002)
    model2 = build_examples.EDSRDFModel(64, set_point=9.0, r0=1.1, gap=0.05, period=8, learning_rate=0.5)
    tfc2 = htf.tfcompute(model2)
    tfc2.attach(sim2.nlist_cell(check_period=1), r_cut=2.5, save_output_period=1)
    sim2.run(20)
    assert tfc2._bplan is None
    np.testing.assert_allclose(system.pos.cpu().numpy(), system2.pos.cpu().numpy(), atol=2e-4)
    np.testing.assert_allclose(float(model.eds_bias.alpha), float(model2.eds_bias.alpha), rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("precision", ["fp32", "split", "split16"])
def test_pair_mlp_model_runs_traced(htf, cuda, precision):
    """config-3 style model through SimModel/tfcompute: traced path, forces against the oracle --
    on the fp32 matrix instruction and on the bf16 one with exactly split operands, same tolerance."""
    from hoomd_tf_amd import standin
    pos, L, a = standin.fcc_positions(5, 0.8442)
    pos = pos + 0.03 * a * np.random.default_rng(0).standard_normal(pos.shape)
    system = standin.System(pos, L, dtype=torch.float32, device=cuda)
    sim = standin.Simulation(system)
    sim.integrate_nve(0.001)
    model = build_examples.PairMLPModel(128, precision=precision)
    tfc = htf.tfcompute(model)
    tfc.attach(sim.nlist_cell(), r_cut=2.5)
    sim.run(5)
    assert tfc._plan is not None
    f = tfc.get_forces_array()
    assert np.all(np.isfinite(f)) and np.abs(f[:, :3]).max() > 1e-3
    nl = tfc.get_nlist_array().astype(np.float64)
    ref = O.pair_mlp_model(nl, model.mlp.params, 0.0, 3.0, "tanh")
    # the buffer holds the pair vectors of the last computeForces call == the last force evaluation
    np.testing.assert_allclose(f, ref, rtol=1e-4, atol=1e-4)


def test_api_errors(htf, cuda):
    """attach()/SimModel error behaviour (tensorflowcompute.py:66-67,95,122-123; simmodel.py:40-42)."""
    from hoomd_tf_amd import standin
    with pytest.raises(AttributeError):
        htf.SimModel(4)
    sim, system, L = _sim(htf, cuda, 3, 4.0)
    with pytest.raises(ValueError):
        htf.tfcompute(build_examples.LJModel(8)).attach()  # nlist required when nneighbor_cutoff > 0
    t = htf.tfcompute(build_examples.LJModel(8))
    t.attach(sim.nlist_cell(), r_cut=3.0)
    with pytest.raises(ValueError):
        t.set_reference_forces(object())
    with pytest.raises(ValueError):
        htf.tfcompute(build_examples.LJModel(8)).attach(sim.nlist_cell(), r_cut=3.0, train=True)
    with pytest.raises(ValueError):
        htf.compute_nlist_forces(torch.zeros(2, 2, 4, device=cuda), torch.zeros(2, device=cuda))
    standin._current["sim"] = None
    with pytest.raises(RuntimeError):
        htf.tfcompute(build_examples.LJModel(8)).attach(None, r_cut=3.0)


def test_log_value_and_hook_rules(htf, cuda):
    """getLogValue('tensorflow') == potential energy (TensorflowCompute.cc:376-395); a hoomd2tf
    compute needs an integrator to hook into (tensorflowcompute.py:183-188)."""
    sim, system, L = _sim(htf, cuda, 4, 1.4, kT=0.3, seed=2)
    tfc = htf.tfcompute(build_examples.LJModel(32))
    tfc.attach(sim.nlist_cell(), r_cut=3.0)
    sim.run(3)
    e = tfc.get_log_value('tensorflow', system.timestep)
    assert abs(e - float(tfc.force[:, 3].double().sum())) < 1e-9 and e != 0.0
    assert abs(float(htf.ops.energy_sum(tfc.force.float())) - e) < 1e-4 * abs(e)
    with pytest.raises(RuntimeError):
        tfc.get_log_value('kinetic', 0)
    tfc.update_coeffs()
    from hoomd_tf_amd import standin
    pos, L2 = sq_lattice(3, 4.0)
    bare = standin.Simulation(standin.System(pos, L2, dtype=torch.float64, device=cuda))
    with pytest.raises(ValueError, match='integrator'):
        htf.tfcompute(build_examples.LJModel(8, output_forces=False)).attach(bare.nlist_cell(), r_cut=5.0)


def test_retrace(htf, cuda):
    """test_tensorflow.py:507-530: a Python attribute read inside compute() is baked into the traced
    step; flipping it has no effect until retrace_compute() drops the installed plan."""
    class Switchable(htf.SimModel):
        def setup(self):
            self.double = False

        def compute(self, nlist, positions, box):
            rinv = htf.nlist_rinv(nlist)
            inv_r6 = rinv**6
            p_energy = (8.0 if self.double else 4.0) / 2.0 * (inv_r6 * inv_r6 - inv_r6)
            return htf.compute_nlist_forces(nlist, htf.reduce_sum(p_energy, axis=1))

    sim, system, L = _sim(htf, cuda, 4, 1.5, kT=None)
    model = Switchable(32)
    tfc = htf.tfcompute(model)
    tfc.attach(sim.nlist_cell(), r_cut=3.0)
    sim.compute_forces()
    sim.compute_forces()
    assert tfc._plan is not None
    f1 = tfc.force.cpu().numpy().copy()
    model.double = True  # without retrace
    sim.compute_forces()
    np.testing.assert_array_equal(tfc.force.cpu().numpy(), f1)
    model.retrace_compute()  # with retrace
    sim.compute_forces()
    np.testing.assert_allclose(tfc.force.cpu().numpy(), 2.0 * f1, rtol=1e-5, atol=1e-5)  # eager step after the retrace: other summation order


def test_sorted(htf, cuda):
    """test_tensorflow.py:850-866: NlistNN(64, dim=32, top_neighs=8) on the 8 x 8 lattice, r_cut 10."""
    sim, system, L = _sim(htf, cuda, 8, 4.0, kT=1.0, seed=1)
    tfc = htf.tfcompute(build_examples.NlistNN(64, dim=32, top_neighs=8))
    tfc.attach(sim.nlist_cell(check_period=1), r_cut=10.0)
    sim.run(10)
    assert np.all(np.isfinite(tfc.force.cpu().numpy()))
