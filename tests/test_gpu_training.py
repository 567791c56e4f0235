"""Online training (FORCE_MODE::hoomd2tf) on the GPU: the loss/gradient sweep
against torch double-backward and finite differences of the oracle, the device optimizers
against the Keras update rules in numpy, and the reference's training tests re-run."""
import numpy as np
import pytest
import torch

import build_examples
from helpers import random_nlist, sq_lattice
from oracle import graph_torch as G
from oracle import htf_oracle as O

pytestmark = pytest.mark.gpu


def _case(seed=0, N=300, NN=64):
    rng = np.random.default_rng(seed)
    nl, _ = random_nlist(rng, N, NN, fill=0.7, rmin=0.9, rmax=2.8, dtype=np.float32)
    nl[0] = 0
    return nl


def test_lj_param_forward_and_loss_gradient(htf, cuda):
    nl = _case()
    nl64 = nl.astype(np.float64)
    labels = O.lj_model(nl64)  # "HOOMD LJ" labels: eps = sig = 1
    theta = [1.2, 0.9]
    w = torch.tensor(theta, dtype=torch.float32, device=cuda)
    pot = htf.Potential.lj_param(*theta, theta=w)
    x = torch.from_numpy(nl).to(cuda)
    # forward == the oracle restatement of example 06's LJLayer
    f = htf.ops.eval_forces(pot, x).cpu().numpy()
    ref = O.lj_param_model(nl64, *theta)
    np.testing.assert_allclose(f, ref, rtol=2e-4, atol=2e-4 * np.abs(ref).max())
    # the device weights are what the kernel reads: changing them changes the next launch
    w[0] = 2.4
    f2 = htf.ops.eval_forces(pot, x).cpu().numpy()
    np.testing.assert_allclose(f2, 2 * f, rtol=1e-5, atol=1e-4)
    w[0] = 1.2
    # loss + gradient through the force
    pred = torch.empty((nl.shape[0], 4), device=cuda)
    accum = htf.ops.train_pair_grad(pot, x, torch.from_numpy(labels).to(cuda), pred=pred).cpu().numpy()
    np.testing.assert_allclose(pred.cpu().numpy(), f, rtol=1e-5, atol=1e-4)
    B = nl.shape[0]
    loss, g = G.mse_grad_wrt_params(lambda n, ww: G.lj_param_forces(n, ww, create_graph=True),
                                    torch.from_numpy(nl64), torch.from_numpy(labels), theta)
    np.testing.assert_allclose(accum[0] / (4 * B), loss, rtol=1e-4)
    np.testing.assert_allclose(accum[1:] / (4 * B), g, rtol=1e-3)
    fd = O.fd_loss_grad(lambda t: O.lj_param_model(nl64, t[0], t[1]), theta, labels)
    np.testing.assert_allclose(accum[1:] / (4 * B), fd, rtol=1e-3)


def test_wca_and_poly_loss_gradients(htf, cuda):
    nl = _case(3, NN=32)
    nl[:, :, :3] *= 0.45  # inside the WCA cut
    nl64 = nl.astype(np.float64)
    x = torch.from_numpy(nl).to(cuda)
    B = nl.shape[0]
    labels = O.wca_model(nl64, 0.55)
    w = torch.tensor([0.5], dtype=torch.float32, device=cuda)
    accum = htf.ops.train_pair_grad(htf.Potential.wca(0.5, theta=w), x, torch.from_numpy(labels).to(cuda)).cpu().numpy()
    # autograd, not finite differences: the cut mask and the clip are piecewise constant in TF's
    # gradient, while a finite difference in sigma sees pairs jump across them
    loss, g = G.mse_grad_wrt_params(lambda n, ww: G.wca_param_forces(n, ww, create_graph=True),
                                    torch.from_numpy(nl64), torch.from_numpy(labels), [0.5])
    np.testing.assert_allclose(accum[0] / (4 * B), loss, rtol=1e-3)
    np.testing.assert_allclose(accum[1:] / (4 * B), g, rtol=5e-3)
    # rinv polynomial with trainable coefficients (linear model): LJ written as 2 s^12 - 2 s^6
    nl = _case(4, NN=32)
    nl64 = nl.astype(np.float64)
    x = torch.from_numpy(nl).to(cuda)
    labels = O.lj_model(nl64)
    c = torch.tensor([1.5, -2.5], dtype=torch.float32, device=cuda)
    accum = htf.ops.train_pair_grad(htf.Potential.rinv_poly([1.5, -2.5], [12, 6], theta=c), x,
                                    torch.from_numpy(labels).to(cuda)).cpu().numpy()
    fd = O.fd_loss_grad(lambda t: O.rinv_poly_model(nl64, list(t), [12, 6]), [1.5, -2.5], labels, h=1e-5)
    np.testing.assert_allclose(accum[1:3] / (4 * nl.shape[0]), fd, rtol=2e-3)
    with pytest.raises(ValueError):
        htf.ops.train_pair_grad(htf.Potential.lj(), x, torch.from_numpy(labels).to(cuda))


@pytest.mark.parametrize("name", ["SGD", "Adam", "Nadam"])
def test_device_optimizers_follow_keras_rules(htf, cuda, name):
    from hoomd_tf_amd import optimizers, _lib
    rng = np.random.default_rng(1)
    theta = np.array([0.9, 1.2], dtype=np.float64)
    dev_theta = torch.tensor(theta, dtype=torch.float32, device=cuda)
    state = torch.zeros(_lib.OPT_STATE_FLOATS, dtype=torch.float32, device=cuda)
    opt = getattr(optimizers, name)(0.01)
    desc = opt.desc(nonneg_mask=0b11)
    ref = {"SGD": None, "Adam": O.KerasAdam(0.01), "Nadam": O.KerasNadam(0.01)}[name]
    for step in range(40):
        g = rng.standard_normal(2)
        accum = torch.tensor([3.0, g[0], g[1]], dtype=torch.float32, device=cuda)
        htf.ops.optimizer_step(dev_theta, accum, 0.5, state, desc)
        theta = theta - 0.01 * 0.5 * g if ref is None else ref.step(theta, 0.5 * g)
        theta = np.maximum(theta, 0.0)
        np.testing.assert_allclose(dev_theta.cpu().numpy(), theta, rtol=2e-4, atol=2e-6)
    assert float(state[19]) == 40 and abs(float(state[18]) - 40 * 1.5) < 1e-3


def _training_sim(htf, cuda, n, a, seed, kT, dt=0.005):
    from hoomd_tf_amd import standin
    pos, L = sq_lattice(n, a)
    pos[:, :2] += 0.1 * np.random.default_rng(seed).standard_normal((n * n, 2))
    system = standin.System(pos, L, dtype=torch.float32, device=cuda)
    sim = standin.Simulation(system)
    system.randomize_velocities(kT, seed)
    system.vel[:, 2] = 0
    sim.integrate_nve(dt)
    return sim, system


def test_trainable(htf, cuda):
    """test_tensorflow.py:155-174 test_trainable: TrainableGraph(16, output_forces=False), Nadam(0.01),
    batch_size=4, labels = the other force (LJ eps 1.1 sig 0.9 upstream; here LJModel): the
    LJ weights move by more than 0.01."""
    sim, system = _training_sim(htf, cuda, 3, 1.3, 2, kT=0.5)
    lj = htf.tfcompute(build_examples.LJModel(16))
    nlist = sim.nlist_cell(check_period=1)
    lj.attach(nlist, r_cut=3.0)
    model = build_examples.TrainableGraph(16, output_forces=False, sig=0.9, eps=1.1)
    model.compile(optimizer=htf.optimizers.Nadam(0.01), loss='MeanSquaredError')
    start = model.get_layer('lj').trainable_weights[0].cpu().numpy().copy()
    tfcompute = htf.tfcompute(model)
    tfcompute.attach(nlist, r_cut=3.0, batch_size=4, train=True)
    sim.run(25)
    end = model.get_layer('lj').trainable_weights[0].cpu().numpy()
    assert np.sum((start - end) ** 2) > 0.01 ** 2, 'No training observed'
    assert np.all(np.isfinite(end)) and np.all(end >= 0)


def test_force_matching_recovers_lj(htf, cuda):
    """examples/06 Force Matching: TrainableLJ(sig 0.9, eps 1.2) trained online against LJ
    (eps = sig = 1) labels approaches them; the loss falls by orders of magnitude."""
    sim, system = _training_sim(htf, cuda, 16, 1.25, 3, kT=0.3, dt=0.002)
    lj = htf.tfcompute(build_examples.LJModel(64))
    nlist = sim.nlist_cell(check_period=1)
    lj.attach(nlist, r_cut=3.0)
    model = build_examples.TrainableGraph(64, output_forces=False, sig=0.9, eps=1.2)
    model.compile(htf.optimizers.Adam(0.01), loss=['MeanSquaredError', None, None])
    tfcompute = htf.tfcompute(model)
    tfcompute.attach(nlist, train=True, r_cut=3.0, save_output_period=5)
    tfcompute.set_reference_forces(lj)
    sim.run(5)
    first = float(tfcompute._opt_state[20])
    sim.run(400)
    w = model.lj.w.cpu().numpy()
    last = float(tfcompute._opt_state[20])
    assert last < 0.02 * first, (first, last)
    # eps and sig trade off along the valley w0 * w1^12 = const of the repulsive wall; 400 Adam
    # steps reach the valley floor (start: 1.2 * 0.9^12 = 0.34) and creep along it towards (1, 1)
    assert abs(w[0] * w[1] ** 12 - 1.0) < 0.2 and abs(w[1] - 1.0) < 0.05, w
    # outputs beyond the loss list: the weights trace and the energy (example 06 plots them)
    assert tfcompute.outputs[0].shape[1] == 2 and tfcompute.outputs[0].shape[0] >= 80
    assert float(model.metrics[0].result()) > 0


def test_force_output(htf, cuda):
    """test_tensorflow.py:400-431: LJModel(32, output_forces=False) 'trained' against the LJ
    reference force: the MSE metric over all four columns stays < 1e-5."""
    sim, system = _training_sim(htf, cuda, 5, 2.0, 1, kT=0.8, dt=0.01)
    nlist = sim.nlist_cell(check_period=1)
    lj = htf.tfcompute(build_examples.LJModel(32))
    lj.attach(nlist, r_cut=3.0)
    lj2 = htf.tfcompute(build_examples.BenchmarkPotential(32))
    lj2.attach(nlist, r_cut=3.0)
    model = build_examples.LJModel(32, output_forces=False)
    model.compile(loss='MeanSquaredError', optimizer='adam')
    tfcompute = htf.tfcompute(model)
    tfcompute.attach(nlist, train=True, r_cut=3.0, period=100)
    tfcompute.set_reference_forces(lj)
    sim.run(300)
    error = float(model.metrics[0].result())
    assert abs(error) < 1e-5
    with pytest.raises(ValueError):
        lj.set_reference_forces(lj2)
    m2 = build_examples.LJModel(8, output_forces=False)
    with pytest.raises(ValueError):
        htf.tfcompute(m2).attach(nlist, train=True, r_cut=3.0)  # not compiled


# ------------------------------------------------------------------ pair-MLP weights (C5b)
def _flat_params(params):
    return np.concatenate([np.asarray(params[k], dtype=np.float64).ravel() for k in ("W1", "b1", "W2", "b2", "W3", "b3")])


@pytest.mark.parametrize("precision", ["fp32", "split16"])
@pytest.mark.parametrize("dims,act,NN", [((32, 64, 64), "tanh", 24), ((12, 20, 28), "tanh", 24), ((32, 64, 64), "linear", 24),
                                         ((32, 64, 64), "tanh", 72), ((32, 64, 64), "tanh", 128)])
def test_pair_mlp_loss_gradient_matches_double_backward(htf, cuda, dims, act, NN, precision):
    """One sweep (value + r-tangent forward, one reverse) == torch's double backward through
    the force, for every one of the 6337 weights; ragged widths exercise the zero padding.
    ``precision``: "fp32" -> mlp_grad_mfma_kernel (fp32 MFMA); "split16" (the default of htf.PairMLP) -> mlp_grad_f16_kernel,
    the same sweep on the fp16 pipeline with hi + lo operands -- same tolerances."""
    from hoomd_tf_amd import initializers
    K, H1, H2 = dims
    # NN = 72 (3 tiles per row) takes the two-pass route (evaluator, then gradient kernel);
    # 1, 2 or 4 tiles per row form the prediction inside the gradient kernel
    nl = _case(5, N=40 if NN == 24 else 21, NN=NN)
    nl64 = nl.astype(np.float64)
    params = initializers.mlp_params(seed=11, K=K, H1=H1, H2=H2)
    rng = np.random.default_rng(4)
    for k in ("b1", "b2", "b3"):  # Keras starts biases at zero; make them count
        params[k] = (0.1 * rng.standard_normal(params[k].shape)).astype(np.float32)
    theta = _flat_params(params)
    labels = O.lj_model(nl64) * 0.05
    w = torch.tensor(theta, dtype=torch.float32, device=cuda)
    pot = htf.Potential.pair_mlp(params, 0.0, 3.0, activation=act, theta=w, precision=precision)
    assert pot.num_params == theta.size == K * H1 + H1 + H1 * H2 + 2 * H2 + 1
    x = torch.from_numpy(nl).to(cuda)
    pred = torch.empty((nl.shape[0], 4), device=cuda)
    accum = htf.ops.train_pair_grad(pot, x, torch.from_numpy(labels).to(cuda), pred=pred).cpu().numpy()
    fwd = lambda n, ww: G.pair_mlp_param_forces(n, ww, dims, act=act, create_graph=True)
    ref_pred = G.pair_mlp_param_forces(torch.from_numpy(nl64), torch.from_numpy(theta), dims, act=act).detach().numpy()
    np.testing.assert_allclose(pred.cpu().numpy(), ref_pred, rtol=1e-4, atol=1e-4 * np.abs(ref_pred).max())
    loss, g = G.mse_grad_wrt_params(fwd, torch.from_numpy(nl64), torch.from_numpy(labels), theta)
    B = nl.shape[0]
    np.testing.assert_allclose(accum[0] / (4 * B), loss, rtol=2e-4)
    got = accum[1:] / (4 * B)
    if NN == 24 and act == "tanh":  # fp64 pair vectors on the wire (HOOMD double build): same sweep
        accum64 = htf.ops.train_pair_grad(pot, torch.from_numpy(nl64).to(cuda), torch.from_numpy(labels).to(cuda)).cpu().numpy()
        np.testing.assert_allclose(accum64, accum, rtol=1e-4, atol=1e-5 * np.abs(accum).max())
    # fp32 accumulation over ~700 pairs against an fp64 reference: tolerance relative to the gradient scale
    assert np.abs(got - g).max() < 2e-4 * np.abs(g).max(), (np.abs(got - g).max(), np.abs(g).max())
    # every block of theta carries signal (no silently-zero slice)
    o = 0
    for n in (K * H1, H1, H1 * H2, H2, H2, 1):
        assert np.abs(got[o:o + n]).max() > 0
        o += n


@pytest.mark.parametrize("precision", ["split16", "fp32"])
@pytest.mark.parametrize("regime", ["residuals 1e-6", "residuals 1e-3", "labels 1e4 x LJ", "labels 1e7 x LJ"])
def test_pair_mlp_gradient_over_the_residual_range(htf, cuda, regime, precision):
    """The split16 sweep carries everything downstream of the reverse seeds (residual x w3 x act') as fp16 hi + lo: unscaled,
    a large residual (early training, close contacts) leaves fp16's range -> inf / NaN gradients, a small one its normal
    numbers -> the gradient loses its bits and vanishes below ~3e-8.  The sweep scales the seeds of a launch by a
    power of two taken from its largest residual and folds it back out exactly.  Checked over thirteen decades of residual,
    every weight against torch's fp64 double backward FED THE SWEEP'S OWN fp32 RESIDUAL (labels_eff = fp64 prediction - (GPU
    prediction - labels)), so that the tolerance stays the fixture's 2e-4 of the gradient scale whatever the residual's size."""
    from hoomd_tf_amd import initializers
    dims, NN = (32, 64, 64), 72
    nl = _case(9, N=25, NN=NN)
    nl64 = nl.astype(np.float64)
    params = initializers.mlp_params(seed=13)
    rng = np.random.default_rng(8)
    for k in ("b1", "b2", "b3"):
        params[k] = (0.1 * rng.standard_normal(params[k].shape)).astype(np.float32)
    theta = _flat_params(params)
    ref_pred = G.pair_mlp_param_forces(torch.from_numpy(nl64), torch.from_numpy(theta), dims, act="tanh").detach().numpy()
    if regime.startswith("residuals"):
        labels = (ref_pred + float(regime.split()[1]) * rng.standard_normal(ref_pred.shape)).astype(np.float32)
    else:
        labels = (float(regime.split()[1]) * O.lj_model(nl64)).astype(np.float32)
    w = torch.tensor(theta, dtype=torch.float32, device=cuda)
    pot = htf.Potential.pair_mlp(params, 0.0, 3.0, activation="tanh", theta=w, precision=precision)
    pred = torch.empty((nl.shape[0], 4), device=cuda)
    accum = htf.ops.train_pair_grad(pot, torch.from_numpy(nl).to(cuda), torch.from_numpy(labels).to(cuda), pred=pred).cpu().numpy()
    assert np.all(np.isfinite(accum)), regime
    resid = pred.cpu().numpy() - labels                      # fp32, as the sweep forms it
    B = nl.shape[0]
    np.testing.assert_allclose(accum[0] / (4 * B), np.mean(resid.astype(np.float64) ** 2), rtol=1e-5)
    labels_eff = ref_pred - resid.astype(np.float64)
    fwd = lambda n, ww: G.pair_mlp_param_forces(n, ww, dims, act="tanh", create_graph=True)
    _, g = G.mse_grad_wrt_params(fwd, torch.from_numpy(nl64), torch.from_numpy(labels_eff), theta)
    got = accum[1:] / (4 * B)
    assert np.abs(g).max() > 0
    err = np.abs(got - g).max() / np.abs(g).max()
    assert err < 2e-4, (regime, precision, err)
    o = 0
    for n in (32 * 64, 64, 64 * 64, 64, 64, 1):   # every block of theta carries signal at every residual size
        assert np.abs(got[o:o + n]).max() > 1e-3 * np.abs(g[o:o + n]).max()
        o += n


@pytest.mark.parametrize("NN,N", [(8, 300), (200, 37), (64, 5)])
def test_pair_mlp_compaction_over_row_shapes(htf, cuda, NN, N):
    """The split16 evaluator and training sweep stage a row 64 slots at a time and cut the stream of live pairs into tiles of 32
    wherever they fall: short rows (many rows per tile), rows longer than two chunks, fewer rows than waves, empty rows in between
    -- forces against the fp64 oracle, the loss gradient against the fp32-MFMA sweep (which walks the rows' own tiles)."""
    from hoomd_tf_amd import initializers
    rng = np.random.default_rng(NN)
    nl, _ = random_nlist(rng, N, NN, fill=0.6, rmin=0.9, rmax=2.8, dtype=np.float32)
    nl[::7] = 0                                   # empty rows between live ones
    params = initializers.mlp_params(seed=21)
    for k in ("b1", "b2", "b3"):
        params[k] = (0.1 * rng.standard_normal(params[k].shape)).astype(np.float32)
    theta = _flat_params(params)
    x = torch.from_numpy(nl).to(cuda)
    ref = O.pair_mlp_model(nl.astype(np.float64), params, 0.0, 3.0, "tanh")
    labels = torch.from_numpy((0.05 * O.lj_model(nl.astype(np.float64))).astype(np.float32)).to(cuda)
    grads = {}
    for precision in ("split16", "fp32"):
        w = torch.tensor(theta, dtype=torch.float32, device=cuda)
        pot = htf.Potential.pair_mlp(params, 0.0, 3.0, activation="tanh", theta=w, precision=precision)
        f = htf.ops.eval_forces(pot, x).cpu().numpy()
        np.testing.assert_allclose(f, ref, rtol=2e-4, atol=2e-4 * max(1.0, np.abs(ref).max()))
        assert np.all(f[::7] == 0)
        pred = torch.empty((N, 4), device=cuda)
        grads[precision] = htf.ops.train_pair_grad(pot, x, labels, pred=pred).cpu().numpy()
        np.testing.assert_allclose(pred.cpu().numpy(), f, rtol=1e-5, atol=1e-5 * max(1.0, np.abs(f).max()))
    a, b = grads["split16"], grads["fp32"]
    np.testing.assert_allclose(a[0], b[0], rtol=1e-5)
    assert np.abs(a[1:] - b[1:]).max() < 1e-4 * np.abs(b[1:]).max()


@pytest.mark.parametrize("outlier", [1e4, 1e6, 1e8])
def test_pair_mlp_gradient_with_residuals_of_mixed_size(htf, cuda, outlier):
    """Round 6 (found by the config-5 test at 1 048 576 rows): the split16 sweep scaled the seeds of a LAUNCH by one power of two,
    taken from its largest residual -- a single outlier row (a close contact under LJ labels) 4e5 x the median pushed everybody
    else's seeds into fp16's subnormals and cost the whole gradient 0.5 % of its largest component.  The sweep now takes its rows
    in windows by residual size, each with its own scale: the gradient of a batch == the gradient of its outlier rows + the
    gradient of the rest, each swept on its own (the sweep is linear in the residuals), to 1e-5 of the REST's largest component
    + 2e-6 of the outliers'; and both against the fp32-MFMA sweep, which has no scale to lose."""
    from hoomd_tf_amd import initializers
    N, NN = 4096, 72
    nl = _case(31, N=N, NN=NN)
    params = initializers.mlp_params(seed=14)
    theta = _flat_params(params)
    x = torch.from_numpy(nl).to(cuda)
    base = (0.05 * O.lj_model(nl.astype(np.float64))).astype(np.float32)
    # residual sizes spread over four decades among the ordinary rows, three rows far above them
    rng = np.random.default_rng(5)
    base *= (10.0 ** rng.uniform(-2.0, 2.0, size=(N, 1))).astype(np.float32)
    big = np.array([17, 1234, 4000])
    scale = np.abs(base).max()
    lab_all = base.copy()
    lab_all[big] = (outlier * scale * rng.standard_normal((3, 4))).astype(np.float32)
    w = torch.tensor(theta, dtype=torch.float32, device=cuda)
    pot = htf.Potential.pair_mlp(params, 0.0, 3.0, activation="tanh", theta=w, precision="split16")
    pred = htf.ops.eval_forces(pot, x)
    # "the rest": the outlier rows' labels equal their prediction (zero residual); "the outliers": everybody else's do
    lab_rest = torch.from_numpy(lab_all).to(cuda)
    lab_rest[torch.from_numpy(big).to(cuda)] = pred[torch.from_numpy(big).to(cuda)]
    lab_out = pred.clone()
    lab_out[torch.from_numpy(big).to(cuda)] = torch.from_numpy(lab_all[big]).to(cuda)
    g_all = htf.ops.train_pair_grad(pot, x, torch.from_numpy(lab_all).to(cuda)).double().cpu().numpy()
    g_rest = htf.ops.train_pair_grad(pot, x, lab_rest).double().cpu().numpy()
    g_out = htf.ops.train_pair_grad(pot, x, lab_out).double().cpu().numpy()
    assert np.all(np.isfinite(g_all)) and np.abs(g_rest[1:]).max() > 0 and np.abs(g_out[1:]).max() > 1e2 * np.abs(g_rest[1:]).max()
    err = np.abs(g_all[1:] - (g_rest[1:] + g_out[1:])).max()
    assert err <= 1e-5 * np.abs(g_rest[1:]).max() + 2e-6 * np.abs(g_out[1:]).max(), (outlier, err, np.abs(g_rest[1:]).max(), np.abs(g_out[1:]).max())
    np.testing.assert_allclose(g_all[0], g_rest[0] + g_out[0], rtol=1e-5)
    w32 = torch.tensor(theta, dtype=torch.float32, device=cuda)
    pot32 = htf.Potential.pair_mlp(params, 0.0, 3.0, activation="tanh", theta=w32, precision="fp32")
    r32 = htf.ops.train_pair_grad(pot32, x, lab_rest).double().cpu().numpy()
    assert np.abs(g_rest[1:] - r32[1:]).max() < 1e-4 * np.abs(r32[1:]).max()


VARIANT_ROUTES = ("valu", "nofuse", "split16-fp32-sweep")


@pytest.mark.parametrize("route", ["valu", "nofuse", "bf16-images", "split-images", "split16-sweep", "split16-fp32-sweep"])
def test_pair_mlp_gradient_alternate_routes(htf, cuda, route, monkeypatch):
    """bf16- / split-image potentials (which train on their own fp32 image set with the fp32-MFMA sweep) and the split16
    potential (the fp16-pipeline sweep, every wave on its own, from the evaluator's images) give the fp32 potential's loss
    gradient.  In a variants build (-DHTF_AB_VARIANTS; test_alternate_routes_against_the_variants_build runs this test on
    one) also: the first-generation VALU kernel (HTF_MLP_TRAIN_VALU), the two-pass matrix-core route
    (HTF_MLP_TRAIN_NOFUSE) and the fp32 sweep for a split16 potential (HTF_MLP_TRAIN_FP32)."""
    from hoomd_tf_amd import initializers
    if route in VARIANT_ROUTES and "libhtf_ab" not in htf._lib.LIB_PATH:
        pytest.skip("a switch of the variants build")
    nl = _case(8, N=33, NN=40)
    params = initializers.mlp_params(seed=12)
    theta = _flat_params(params)
    labels = torch.from_numpy((O.lj_model(nl.astype(np.float64)) * 0.05).astype(np.float32)).to(cuda)
    x = torch.from_numpy(nl).to(cuda)

    def grad(precision="fp32"):
        w = torch.tensor(theta, dtype=torch.float32, device=cuda)
        pot = htf.Potential.pair_mlp(params, 0.0, 3.0, activation="tanh", precision=precision, theta=w)
        return htf.ops.train_pair_grad(pot, x, labels).cpu().numpy()

    base = grad()
    if route == "valu":
        monkeypatch.setenv("HTF_MLP_TRAIN_VALU", "1")
        got = grad()
    elif route == "nofuse":
        monkeypatch.setenv("HTF_MLP_TRAIN_NOFUSE", "1")
        got = grad()
    elif route == "split-images":
        got = grad("split")
    elif route == "split16-sweep":
        got = grad("split16")
    elif route == "split16-fp32-sweep":
        monkeypatch.setenv("HTF_MLP_TRAIN_FP32", "1")
        got = grad("split16")
    else:
        got = grad("bf16")
    scale = np.abs(base[1:]).max()
    assert np.abs(got[1:] - base[1:]).max() < 5e-5 * scale, np.abs(got[1:] - base[1:]).max() / scale
    np.testing.assert_allclose(got[0], base[0], rtol=1e-5)


def test_alternate_routes_against_the_variants_build(htf, cuda):
    """The kernels that lost their A/B (first-generation VALU sweep, two-pass fp32 route, fp32 sweep for split16 potentials) live
    in the variants build only; the alternate-routes test, all of it, in a child process on that library."""
    import os
    import subprocess
    import sys
    from helpers import ROOT, variants_env
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "gradient_alternate_routes"], cwd=ROOT, env=variants_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " passed" in r.stdout and "skipped" not in r.stdout.split("\n")[-2], r.stdout[-3000:] + r.stderr[-2000:]


def test_split16_weight_range_is_checked_after_an_update(htf, cuda):
    """Host-supplied weights are checked against fp16's range at creation; a TRAINED device vector cannot be.  The
    image build flags a weight whose forward operand (x 2 log2 e) leaves the range -- or is not a number -- in a host-mapped
    word, and the next evaluation / training call refuses with a ValueError instead of producing inf / NaN forces."""
    from hoomd_tf_amd import initializers
    params = initializers.mlp_params(seed=3)
    theta = torch.tensor(_flat_params(params), dtype=torch.float32, device=cuda)
    pot = htf.Potential.pair_mlp(params, 0.0, 3.0, activation="tanh", precision="split16", theta=theta)
    x = torch.from_numpy(_case(2, N=16, NN=32)).to(cuda)
    f0 = htf.ops.eval_forces(pot, x)
    assert torch.isfinite(f0).all()
    theta[5] = 3.0e4          # x 2.885 = 8.7e4 > 65504
    pot.refresh()
    torch.cuda.synchronize()
    with pytest.raises(ValueError, match="fp16"):
        htf.ops.eval_forces(pot, x)
    with pytest.raises(ValueError, match="fp16"):   # refused until an image build finds every weight in range again
        htf.ops.train_pair_grad(pot, x, torch.zeros((16, 4), device=cuda))
    # repaired and refreshed: every image build judges the range afresh (the word used to be sticky)
    theta[5] = float(_flat_params(params)[5])
    pot.refresh()
    torch.cuda.synchronize()
    assert torch.equal(htf.ops.eval_forces(pot, x), f0)
    assert torch.isfinite(htf.ops.train_pair_grad(pot, x, torch.zeros((16, 4), device=cuda))).all()
    # the exact three-part bf16 split has no range limit
    theta2 = theta.clone()
    pot2 = htf.Potential.pair_mlp(params, 0.0, 3.0, activation="tanh", precision="split", theta=theta2)
    assert torch.isfinite(htf.ops.eval_forces(pot2, x)).all()


def test_pair_mlp_refresh_tracks_device_weights(htf, cuda):
    """The persistent potential reads the flat device vector: after an in-place change +
    refresh, the MFMA evaluator gives what a freshly built potential gives."""
    layer = htf.PairMLP(32, 64, 64, 0.0, 3.0, activation="tanh", seed=5)
    nl = _case(6, N=64, NN=32)
    x = torch.from_numpy(nl).to(cuda)
    pot = layer.potential()
    f0 = htf.ops.eval_forces(pot, x).cpu().numpy()
    ws = layer.get_weights()
    rng = np.random.default_rng(0)
    ws2 = [w + 0.05 * rng.standard_normal(w.shape).astype(np.float32) for w in ws]
    layer.set_weights(ws2)
    assert layer.potential() is pot
    f1 = htf.ops.eval_forces(pot, x).cpu().numpy()
    fresh = htf.Potential.pair_mlp(dict(zip(("W1", "b1", "W2", "b2", "W3", "b3"), ws2)), 0.0, 3.0, activation="tanh",
                                   precision=layer.precision)
    f2 = htf.ops.eval_forces(fresh, x).cpu().numpy()
    assert np.abs(f1 - f0).max() > 1e-3
    np.testing.assert_array_equal(f1, f2)
    for a, b in zip(layer.get_weights(), ws2):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("name", ["SGD", "Adam", "Nadam"])
def test_vector_optimizer_follows_keras_rules(htf, cuda, name):
    from hoomd_tf_amd import optimizers
    rng = np.random.default_rng(2)
    P = 700
    theta = rng.standard_normal(P)
    dev_theta = torch.tensor(theta, dtype=torch.float32, device=cuda)
    state = torch.zeros(htf.ops.optimizer_state_floats(P), dtype=torch.float32, device=cuda)
    opt = getattr(optimizers, name)(0.01)
    desc = opt.desc(0, (0.0,))
    ref = {"SGD": None, "Adam": O.KerasAdam(0.01), "Nadam": O.KerasNadam(0.01)}[name]
    for step in range(25):
        g = rng.standard_normal(P)
        accum = torch.tensor(np.concatenate([[3.0], g]), dtype=torch.float32, device=cuda)
        htf.ops.optimizer_step(dev_theta, accum, 0.5, state, desc)
        theta = theta - 0.01 * 0.5 * g if ref is None else ref.step(theta, 0.5 * g)
        np.testing.assert_allclose(dev_theta.cpu().numpy(), theta, rtol=3e-4, atol=3e-6)
    assert float(state[19]) == 25 and abs(float(state[18]) - 25 * 1.5) < 1e-3


def test_pair_mlp_force_matching_online(htf, cuda):
    """C5b: PairMLPModel trained online (hoomd2tf) against LJ reference forces every step; the
    loss falls, the weights on the device move, and inference afterwards uses them."""
    sim, system = _training_sim(htf, cuda, 12, 1.25, 4, kT=0.3, dt=0.002)
    lj = htf.tfcompute(build_examples.LJModel(48))
    nlist = sim.nlist_cell(check_period=1)
    lj.attach(nlist, r_cut=3.0)
    model = build_examples.PairMLPModel(48, output_forces=False, activation='tanh', seed=9)
    model.compile(htf.optimizers.Adam(0.003), loss='MeanSquaredError')
    start = np.concatenate([w.ravel() for w in model.mlp.get_weights()])
    tfcompute = htf.tfcompute(model)
    tfcompute.attach(nlist, train=True, r_cut=3.0)
    tfcompute.set_reference_forces(lj)
    sim.run(3)
    first = float(tfcompute._opt_state[20])
    sim.run(150)
    last = float(tfcompute._opt_state[20])
    assert np.isfinite(last) and last < 0.5 * first, (first, last)
    end = np.concatenate([w.ravel() for w in model.mlp.get_weights()])
    assert np.abs(end - start).max() > 1e-3 and np.all(np.isfinite(end))
    # the evaluator now predicts with the trained weights
    ws = dict(zip(("W1", "b1", "W2", "b2", "W3", "b3"), model.mlp.get_weights()))
    rng = np.random.default_rng(1)
    probe, _ = random_nlist(rng, 32, 48, fill=0.6, rmin=0.9, rmax=2.8, dtype=np.float32)
    x = torch.from_numpy(probe).to(cuda)
    got = htf.ops.eval_forces(model.mlp.potential(), x).cpu().numpy()
    want = htf.ops.eval_forces(htf.Potential.pair_mlp(ws, 0.0, 3.0, activation='tanh'), x).cpu().numpy()
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("precision", ["fp32", "split16"])
@pytest.mark.parametrize("seed", range(6))
def test_pair_mlp_gradient_random_shapes(htf, cuda, seed, precision):
    """Random widths (zero-padded operand blocks), NN (fused / two-pass route), activation and label
    precision against torch double backward."""
    from hoomd_tf_amd import initializers
    rng = np.random.default_rng(500 + seed)
    K, H1, H2 = int(rng.integers(2, 33)), int(rng.integers(1, 65)), int(rng.integers(1, 65))
    NN = int(rng.choice([5, 17, 33, 64, 90, 128]))
    act = "tanh" if rng.integers(0, 2) else "linear"
    N = int(rng.integers(3, 40))
    nl = _case(600 + seed, N=N, NN=NN)
    nl64 = nl.astype(np.float64)
    params = initializers.mlp_params(seed=20 + seed, K=K, H1=H1, H2=H2)
    for k in ("b1", "b2", "b3"):
        params[k] = (0.1 * rng.standard_normal(params[k].shape)).astype(np.float32)
    theta = _flat_params(params)
    labels = O.lj_model(nl64) * 0.05
    lab = torch.from_numpy(labels if rng.integers(0, 2) else labels.astype(np.float32)).to(cuda)
    w = torch.tensor(theta, dtype=torch.float32, device=cuda)
    pot = htf.Potential.pair_mlp(params, 0.0, 3.0, activation=act, theta=w, precision=precision)
    accum = htf.ops.train_pair_grad(pot, torch.from_numpy(nl).to(cuda), lab).cpu().numpy()
    dims = (K, H1, H2)
    loss, g = G.mse_grad_wrt_params(lambda n, ww: G.pair_mlp_param_forces(n, ww, dims, act=act, create_graph=True),
                                    torch.from_numpy(nl64), torch.from_numpy(labels), theta)
    np.testing.assert_allclose(accum[0] / (4 * N), loss, rtol=3e-4)
    got = accum[1:] / (4 * N)
    assert np.abs(got - g).max() < 3e-4 * max(np.abs(g).max(), 1e-6), (K, H1, H2, NN, act, np.abs(got - g).max(), np.abs(g).max())


@pytest.mark.parametrize("precision", ["split16", "fp32"])
def test_pair_mlp_gradient_full_size(htf, cuda, precision):
    """The training sweep at the size bench.py times it (C5b: 131 072 x 128, every persistent wave busy, every wave's partial
    in the reduction) against torch's fp64 double backward (tensorflowcompute.py:347-370 train_on_batch; test_tensorflow.py:400-431).
    The batch is 64 row-permuted replicas of one 2 048-row block of the C3 fcc box's own pair vectors, so the reference stays a
    2 048-row double backward: sum of squared residuals and every one of the 6 337 weight gradients == 64 x the block's, the
    prediction on 512 sampled rows == the fp64 oracle.  Twice: labels = 0.05 x LJ (residuals O(1)) and x 1e4 (the per-launch
    power-of-two seed scaling of the split16 sweep at size).  As in test_pair_mlp_gradient_over_the_residual_range the reference
    is fed the sweep's own fp32 residual of the block, so the tolerance is the fixture's 2e-4 of the gradient scale."""
    from hoomd_tf_amd import initializers, standin
    from test_gpu_parity import _jittered, _record
    dims, NN, R, BLOCK = (32, 64, 64), 128, 64, 2048
    sysm, nlc, L = _jittered(standin, cuda, "fcc", 32, 3)
    assert sysm.N == R * BLOCK
    pv = htf.ops.build_pair_vectors(sysm.pos, nlc.n_neigh, nlc.head_list, nlc.nlist, sysm.box, 3.0, NN)
    rng = np.random.default_rng(17)
    first = int(rng.integers(0, sysm.N - BLOCK))
    block = pv[first:first + BLOCK].clone()
    del pv
    perm = torch.from_numpy(rng.permutation(R * BLOCK)).to(cuda)
    src_row = perm % BLOCK                                    # batch row i is block row src_row[i]
    x = block[src_row].contiguous()
    assert x.shape == (R * BLOCK, NN, 4)
    blk = block.cpu().numpy()
    blk64 = blk.astype(np.float64)
    live = int((np.abs(blk[:, :, :3]).sum(axis=2) > 0).sum())
    assert live > 80 * BLOCK                                  # a liquid-density row: ~95 of 128 slots
    params = initializers.mlp_params(seed=11)
    for k in ("b1", "b2", "b3"):
        params[k] = (0.1 * rng.standard_normal(params[k].shape)).astype(np.float32)
    theta = _flat_params(params)
    ref_pred = G.pair_mlp_param_forces(torch.from_numpy(blk64), torch.from_numpy(theta), dims).detach().numpy()
    oracle_pred = O.pair_mlp_model(blk64, params, 0.0, 3.0, "tanh")
    np.testing.assert_allclose(ref_pred, oracle_pred, rtol=1e-5, atol=1e-5)   # the two restatements (fp32 RBF centres in one)
    fwd = lambda n, ww: G.pair_mlp_param_forces(n, ww, dims, create_graph=True)
    for scale in (0.05, 500.0):
        lab_blk = (scale * O.lj_model(blk64)).astype(np.float32)
        labels = torch.from_numpy(lab_blk).to(cuda)[src_row].contiguous()
        w = torch.tensor(theta, dtype=torch.float32, device=cuda)
        pot = htf.Potential.pair_mlp(params, 0.0, 3.0, activation="tanh", theta=w, precision=precision)
        pred = torch.empty((R * BLOCK, 4), device=cuda)
        accum = htf.ops.train_pair_grad(pot, x, labels, pred=pred)
        again = htf.ops.train_pair_grad(pot, x, labels)
        assert torch.equal(accum, again)                     # fixed combination order: deterministic at size
        accum = accum.cpu().numpy().astype(np.float64)
        assert np.all(np.isfinite(accum))
        # prediction: every replica of a block row is the same row.  The fp32 sweep walks a row's own tiles -> identical bits;
        # split16 cuts the stream of live pairs into 32-pair tiles ACROSS rows, so a row's fp32 partial sums depend on where
        # its pairs fall in the stream -> equal to rounding.  512 sampled rows against the oracle.
        p = pred.cpu().numpy()
        inv = torch.argsort(perm).cpu().numpy()               # batch positions ordered by perm value
        first_replica = p[inv[:BLOCK]]                        # perm values 0..BLOCK-1: block rows in order
        tiled = first_replica[src_row.cpu().numpy()]
        if precision == "fp32":
            assert np.array_equal(p, tiled)
        else:
            assert np.abs(p - tiled).max() <= 2e-6 * max(1.0, np.abs(p).max()), np.abs(p - tiled).max()
        rows = rng.choice(BLOCK, 512, replace=False)
        np.testing.assert_allclose(first_replica[rows], oracle_pred[rows], rtol=5e-5, atol=2e-5 * max(1.0, np.abs(oracle_pred).max()))
        resid = first_replica.astype(np.float32) - lab_blk    # fp32, as the sweep forms it
        want_loss = np.sum((p.astype(np.float32) - lab_blk[src_row.cpu().numpy()]).astype(np.float64) ** 2)
        loss_err = abs(accum[0] - want_loss) / want_loss
        assert loss_err < 2e-4, (scale, accum[0], want_loss)
        _, g = G.mse_grad_wrt_params(fwd, torch.from_numpy(blk64), torch.from_numpy(ref_pred - resid.astype(np.float64)), theta)
        want = g * (4 * BLOCK) * R                            # mse_grad_wrt_params returns the MEAN's gradient over the block
        err = np.abs(accum[1:] - want).max() / np.abs(want).max()
        _record("train_full_size_%s_labels_x%g" % (precision, scale), max_abs_err=np.abs(accum[1:] - want).max(),
                max_ratio_strict=err / 2e-4, max_ref=np.abs(want).max(), loss_rel_err=loss_err)
        assert err < 2e-4, (precision, scale, err)
        o = 0
        for n in (32 * 64, 64, 64 * 64, 64, 64, 1):           # every block of theta carries signal
            assert np.abs(accum[1 + o:1 + o + n]).max() > 1e-3 * np.abs(want[o:o + n]).max()
            o += n


# --------------------------------------------------------------------------- the training sweep from the index list (round 6)
@pytest.mark.parametrize("kind", ["lj_param", "wca", "rinv_poly", "traced"])
@pytest.mark.parametrize("NN,wire", [(96, torch.float32), (40, torch.float32), (96, torch.float64)])
def test_training_sweep_from_the_index_list_equals_the_tensor_sweep(htf, cuda, kind, NN, wire):
    """htf_train_pair_grad_list -- positions gathered through HOOMD's index list, pair vectors formed in registers, no [N, NN, 4]
    tensor -- against htf_train_pair_grad on the tensor htf_build_pair_vectors writes from the same arrays: prediction, summed
    squared residual and every weight's gradient equal to the rounding of a row's fp32 sums (their ORDER is all that differs).  NN =
    40 at r_cut 2.5 overflows every row of a liquid: the reference's slot wrap keeps the LAST NN neighbors, in both sweeps.
    (The tensor sweep itself is pinned against fp64 double backward by the tests above.)"""
    from hoomd_tf_amd import standin
    pos, L, a = standin.fcc_positions(6, 0.8442)
    rng = np.random.default_rng(12)
    pos = pos + 0.04 * a * rng.standard_normal(pos.shape)
    pos -= np.round(pos / L) * L
    sysm = standin.System(pos, L, dtype=wire, device=cuda)
    nl = standin.CellNlist(sysm, r_cut=2.5, r_buff=0.4)
    nl.build()
    if kind == "lj_param":
        pot = htf.Potential.lj_param(0.9, 1.05, theta=torch.tensor([0.9, 1.05], device=cuda))
    elif kind == "wca":
        pot = htf.Potential.wca(0.95, theta=torch.tensor([0.95], device=cuda))
    elif kind == "rinv_poly":
        pot = htf.Potential.rinv_poly([1.5, -2.5], [12, 6], theta=torch.tensor([1.5, -2.5], device=cuda))
    else:
        x0 = htf.Nlist(torch.zeros((2, 4, 4), device=cuda))
        w = torch.nn.Parameter(torch.tensor([0.8, 1.05], device=cuda))
        q = (w[1] * htf.nlist_rinv(x0)) ** 6
        e = htf.reduce_sum(w[0] * 2.0 * (q * q - q), axis=1)
        assert e.lowers()
        pot = e.layer.potential(cuda)
    pv = htf.ops.build_pair_vectors(sysm.pos, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, 2.5, NN)
    kept = int(((pv[:, :, :3] != 0).any(dim=2)).sum(dim=1).min())
    assert (kept == NN) if NN == 40 else (kept < NN)                       # NN = 40: every row is full (and overflowed)
    labels = (0.05 * torch.randn((sysm.N, 4), generator=torch.Generator().manual_seed(3))).to(cuda)
    p0, p1 = torch.empty((sysm.N, 4), device=cuda), torch.empty((sysm.N, 4), device=cuda)
    a0 = htf.ops.train_pair_grad(pot, pv, labels, pred=p0)
    a1 = htf.ops.train_pair_grad_list(pot, sysm.pos, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, 2.5, NN, labels, pred=p1)
    assert torch.equal(a1, htf.ops.train_pair_grad_list(pot, sysm.pos, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, 2.5, NN, labels))
    scale = float(p0.abs().max())
    assert float((p0 - p1).abs().max()) <= 2e-6 * scale + 1e-7, float((p0 - p1).abs().max()) / scale
    a0, a1 = a0.double().cpu().numpy(), a1.double().cpu().numpy()
    assert abs(a0[0] - a1[0]) <= 2e-6 * abs(a0[0])
    assert np.abs(a0[1:] - a1[1:]).max() <= 2e-5 * np.abs(a0[1:]).max() + 1e-6, (a0, a1)


def test_training_through_tfcompute_from_the_list_walks_the_tensor_sweeps_weights(htf, cuda, monkeypatch):
    """examples/06's LJLayer trained by force matching at every MD step: the replayed training step from the index list (default)
    against the same with HTF_TRAIN_FROM_TENSOR=1 (pair vectors written, then swept): same weights after 40 steps to fp32
    rounding of the sums, and get_nlist_array() still returns the step's tensor (built on demand)."""
    from hoomd_tf_amd import standin
    import build_examples

    def run(from_tensor):
        monkeypatch.setenv("HTF_TRAIN_FROM_TENSOR", "1" if from_tensor else "0")
        pos, L, a = standin.fcc_positions(6, 0.8442)
        rng = np.random.default_rng(2)
        pos = pos + 0.03 * a * rng.standard_normal(pos.shape)
        pos -= np.round(pos / L) * L
        sysm = standin.System(pos, L, dtype=torch.float32, device=cuda)
        sysm.randomize_velocities(kT=0.5, seed=2)
        sim = standin.Simulation(sysm)
        sim.integrate_nve(0.002)
        nlist = sim.nlist_cell(check_period=1)
        lj = htf.tfcompute(build_examples.LJModel(96))
        lj.attach(nlist, r_cut=2.5)
        model = build_examples.TrainableGraph(96, output_forces=False, sig=0.8, eps=1.05)
        model.compile(htf.optimizers.Adam(0.01), loss='MeanSquaredError')
        tfc = htf.tfcompute(model)
        tfc.attach(nlist, train=True, r_cut=2.5)
        tfc.set_reference_forces(lj)
        sim.run(40)
        torch.cuda.synchronize()
        return tfc, model.lj.w.detach().cpu().numpy().copy(), float(tfc._opt_state[20])

    tfc, w1, l1 = run(False)
    assert tfc._tplan is not None and tfc._tplan["from_list"]
    nl_arr = tfc.get_nlist_array()
    assert nl_arr.shape == (tfc.system.N, 96, 4) and np.abs(nl_arr).max() > 0
    tfc0, w0, l0 = run(True)
    assert tfc0._tplan is not None and not tfc0._tplan["from_list"]
    assert np.abs(w1 - w0).max() < 2e-5 and abs(l1 - l0) < 1e-4 * abs(l0)
