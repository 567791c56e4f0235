"""Closed-form oracle (oracle/htf_oracle.py) vs the op-for-op torch graph with
autograd (oracle/graph_torch.py): an independent check of every hand-derived
gradient, in fp64 (tight) and fp32 (round-off level)."""
import numpy as np
import pytest
import torch

from helpers import random_nlist
from oracle import graph_torch as G
from oracle import htf_oracle as O


def _nl(dtype, seed=0, N=48, NN=16, rmin=0.85, exact_zero_t=False):
    rng = np.random.default_rng(seed)
    nl, _ = random_nlist(rng, N, NN, fill=0.6, rmin=rmin, rmax=2.5, ntypes=2, dtype=dtype)
    # edge rows: r' just below / above the 3e-6 mask, exact zeros, tiny negatives
    nl[0, :, :] = 0
    nl[1, 0, :3] = [1.6e-6, 1.6e-6, 1.6e-6]   # r' ~ 2.9e-6 < delta -> masked
    nl[1, 1, :3] = [1.8e-6, 1.8e-6, 1.8e-6]   # r' ~ 3.3e-6 > delta -> 1/(r'+delta) ~ 1.6e5
    if exact_zero_t:
        # t == 0 exactly: forward is 0; the TF gradient is 0 * (0.5/0) = NaN upstream
        # (measure-zero input, excluded from gradient parity)
        nl[2, 0, :3] = [-1e-7, -1e-7, -1e-7]
    return nl


def _tol(dtype):
    return dict(rtol=1e-10, atol=1e-9) if dtype == np.float64 else dict(rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_lj_vs_autograd(dtype):
    nl = _nl(dtype)
    nl[1, :2] = 0  # s^13 of 1.6e5 overflows fp32: keep that probe for rinv itself
    f, v = O.lj_model(nl, virial=True)
    tf_, tv = G.lj_model(torch.from_numpy(nl), virial=True)
    scale = np.abs(f).max()
    np.testing.assert_allclose(f / scale, tf_.numpy() / scale, **_tol(dtype))
    np.testing.assert_allclose(v / scale, tv.numpy() / scale, **_tol(dtype))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_rinv_edges(dtype):
    nl = _nl(dtype, exact_zero_t=True)
    s = O.nlist_rinv(nl)
    ts = G.nlist_rinv(torch.from_numpy(nl)).numpy()
    np.testing.assert_allclose(s, ts, rtol=1e-6)
    assert s[0].max() == 0 and s[1, 0] == 0 and s[1, 1] > 1e5 and s[2, 0] == 0
    nl = _nl(dtype)
    f = O.benchmark_potential(nl)
    tf_ = G.benchmark_potential(torch.from_numpy(nl)).numpy()
    assert np.all(np.isfinite(f))
    scale = np.abs(f).max()
    np.testing.assert_allclose(f / scale, tf_ / scale, **_tol(dtype))


@pytest.mark.parametrize("sigma", [0.5, 1.0, 2.0])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_wca_vs_autograd(dtype, sigma):
    nl = _nl(dtype, seed=3, rmin=0.3)
    nl[1, :2] = 0
    # r exactly at the cut 2^(1/3) sigma (mask uses strict <) and e straddling the clip at 10
    rc = sigma * 2 ** (1 / 3)
    nl[3, 0, :3] = [rc, 0, 0]
    r10 = sigma * 10 ** (-1 / 6)
    nl[3, 1, :3] = [r10 * 1.001, 0, 0]
    nl[3, 2, :3] = [r10 * 0.999, 0, 0]
    f = O.wca_model(nl, sigma)
    tf_ = G.wca_model(torch.from_numpy(nl), sigma).numpy()
    scale = max(np.abs(f).max(), 1.0)
    np.testing.assert_allclose(f / scale, tf_ / scale, **_tol(dtype))
    e = O.wca_pair_energy(nl, sigma)
    assert e.max() <= 10 and e.min() >= 0
    assert e[3, 2] == 10.0 and 9 < e[3, 1] < 10


@pytest.mark.parametrize("act", ["tanh", "linear"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_pair_mlp_vs_autograd(dtype, act):
    nl = _nl(dtype, seed=5)
    params = O.make_mlp_params(seed=3, K=8, H1=16, H2=16, bias_scale=0.1)
    f = O.pair_mlp_model(nl, params, low=0.0, high=3.0, act=act)
    tf_ = G.pair_mlp_model(torch.from_numpy(nl), params, 0.0, 3.0, act).numpy()
    scale = max(np.abs(f).max(), 1.0)
    np.testing.assert_allclose(f / scale, tf_ / scale, **_tol(dtype))
    # padded rows carry no energy / force
    assert np.all(f[0] == 0)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_eds_rdf_composite_vs_autograd(dtype):
    """config C4 closed form (LJ + alpha * soft-RDF CV) against the torch graph."""
    nl = _nl(dtype, seed=9)
    nl[1, :2] = 0
    f, cv = O.eds_rdf_model(nl, 0.7, 1.1, 0.05)
    tf_, tcv = G.eds_rdf_model(torch.from_numpy(nl), 0.7, 1.1, 0.05)
    scale = np.abs(f).max()
    np.testing.assert_allclose(f / scale, tf_.numpy() / scale, **_tol(dtype))
    np.testing.assert_allclose(cv, float(tcv), rtol=1e-5)
    # alpha = 0 reduces to LJModel; the gauss model alone reproduces the bias part
    f0, _ = O.eds_rdf_model(nl, 0.0, 1.1, 0.05)
    np.testing.assert_allclose(f0, O.lj_model(nl), rtol=1e-6, atol=1e-6)
    fg = O.gauss_model(nl, 1.1, 0.05, 1.0)
    np.testing.assert_allclose((f - f0)[:, :3], 0.7 * fg[:, :3], rtol=2e-3 if dtype == np.float32 else 1e-9, atol=1e-4 if dtype == np.float32 else 1e-9)


def test_rinv_poly_is_lj():
    nl = _nl(np.float64)
    nl[1, :2] = 0
    a = O.rinv_poly_model(nl, [2.0, -2.0], [12, 6], virial=True)
    b = O.lj_model(nl, virial=True)
    np.testing.assert_allclose(a[0], b[0], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-12, atol=1e-12)


def test_rbf_values():
    x = np.linspace(0, 3, 17, dtype=np.float32)
    a = O.rbf_expansion(x, 0, 2, 10)
    b = G.rbf_expansion(torch.from_numpy(x), 0, 2, 10).numpy()
    np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-7)
    # centre k responds with exactly 1 at x = c_k
    c, gap = O.rbf_centers(0, 2, 10)
    np.testing.assert_allclose(np.diag(O.rbf_expansion(c, 0, 2, 10)), 1.0)


def test_eds_trace_matches_float64_adam():
    """EDSLayer fp32 trace stays close to an fp64 run of the same state machine and
    only changes alpha on steps n == period-1."""
    rng = np.random.default_rng(11)
    cvs = 4.0 + rng.standard_normal(200)
    e32, e64 = O.EDSLayer(4.0, 5, 0.2), O.EDSLayer(4.0, 5, 0.2, dtype=np.float64)
    prev = 0.0
    for i, cv in enumerate(cvs):
        a32, a64 = e32(cv), e64(cv)
        assert abs(a32 - a64) < 1e-4
        if i % 5 != 4:
            assert a64 == prev
        prev = a64
    assert prev != 0.0


def test_compute_rdf_and_mask():
    rng = np.random.default_rng(2)
    nl, cnt = random_nlist(rng, 64, 16, fill=0.5, rmin=0.5, rmax=4.0, ntypes=2)
    rdf, rs = O.compute_rdf(nl, [0, 3.5], nbins=10)
    assert rdf.shape == (10,) and rs.shape == (10,)
    # hand count for one interior bin: bins are (3.5/12) wide, bin b covers [b, b+1)*w
    r = np.sqrt(np.sum(nl[..., :3].astype(np.float32) ** 2, axis=2)).ravel()
    w = np.float32(3.5) / 12
    b = 5
    count = np.sum((np.floor(12 * (r / np.float32(3.5))) == b))
    shell = np.linspace(0, 3.5, 11).astype(np.float32)
    np.testing.assert_allclose(rdf[b - 1], count / (shell[b] ** 3 - shell[b - 1] ** 3), rtol=1e-6)
    types = (np.arange(64) % 2).astype(np.float32)
    ab, _ = O.compute_rdf(nl, [0, 3.5], types, nbins=10, type_i=0, type_j=1)
    assert ab.shape == (10,)
    m = O.masked_nlist(nl, types, 0, 1)
    assert m.shape[0] == 32 and np.all(m[..., 3][m[..., 3] != 0] == 1)


def test_positions_radial_model_vs_autograd_and_by_hand():
    """a15: compute_positions_forces of BenchmarkNonlistModel's energy (build_examples.py:59-64)."""
    import torch
    rng = np.random.default_rng(4)
    p = rng.uniform(-3, 3, size=(40, 4))
    p[:, 3] = rng.integers(0, 3, size=40)
    p[7] = 0.0                                   # |p| = 0: divide_no_nan
    ref = O.positions_radial_model(p)
    t = torch.tensor(p, requires_grad=True)
    n = torch.sqrt((t * t).sum(dim=1))
    e = torch.where(n > 0, 1.0 / torch.where(n > 0, n, torch.ones_like(n)), torch.zeros_like(n))
    (g,) = torch.autograd.grad(e.sum(), t)
    ok = np.arange(40) != 7                      # autograd through sqrt at 0 is NaN (TF's norm gradient too)
    np.testing.assert_allclose(ref[ok, :3], -g.numpy()[ok, :3], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(ref[:, 3], e.detach().numpy(), rtol=1e-12)
    assert np.all(ref[7] == 0.0) and np.all(np.isnan(g.numpy()[7, :3]))
    # by hand: p = (3, 0, 4, type 0): |p| = 5, e = 1/5, F = p / |p|^3;  p = (1, 2, 2, type 4): |p| = 5 as well
    f = O.positions_radial_model(np.array([[3.0, 0.0, 4.0, 0.0], [1.0, 2.0, 2.0, 4.0]]))
    np.testing.assert_allclose(f, [[3 / 125, 0.0, 4 / 125, 0.2], [1 / 125, 2 / 125, 2 / 125, 0.2]], rtol=1e-14)


@pytest.mark.parametrize("act", [None, "tanh"])
def test_topk_mlp_model_vs_autograd(act):
    """Example 08 (build_examples.py:199-218): the oracle's closed-form backward through sort -> Dense x 3
    against torch.autograd of the op-for-op graph (torch.sort is stable with stable=True)."""
    import torch
    from helpers import random_nlist
    rng = np.random.default_rng(8)
    nl, _ = random_nlist(rng, 40, 24, fill=0.6, rmin=0.8, rmax=3.0, dtype=np.float64)
    nl[0] = 0
    nl[1, 3:] = 0                                   # fewer real neighbors (3) than K: zeros enter the top 8
    K, H = 8, 16
    params = {"W1": rng.normal(0, 0.4, (K, H)), "b1": rng.normal(0, 0.1, H), "W2": rng.normal(0, 0.3, (H, H)),
              "b2": rng.normal(0, 0.1, H), "W3": rng.normal(0, 0.3, (H, 1)), "b3": rng.normal(0, 0.1, 1)}
    ref, g = O.topk_mlp_model(nl, params, act=act, return_grad=True)
    x = torch.tensor(nl, requires_grad=True)
    tt = x[:, :, :3] + 1e-7
    r = torch.sqrt((tt * tt).sum(dim=2))
    rinv = torch.where(r > 3e-6, 1.0 / (r + 3e-6), torch.zeros_like(r))
    top = torch.sort(rinv, dim=1, descending=True, stable=True)[0][:, :K]
    W = {k: torch.tensor(v) for k, v in params.items()}
    h = top @ W["W1"] + W["b1"]
    h = torch.tanh(h) if act == "tanh" else h
    h = h @ W["W2"] + W["b2"]
    h = torch.tanh(h) if act == "tanh" else h
    e = (h @ W["W3"] + W["b3"])[:, 0]
    (gx,) = torch.autograd.grad(e.sum(), x)
    np.testing.assert_allclose(g, gx.numpy(), rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(ref[:, :3], 2.0 * gx.numpy()[:, :, :3].sum(axis=1), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(ref[:, 3], e.detach().numpy(), rtol=1e-12)


def test_topk_mlp_tie_order_on_a_lattice():
    """Equidistant neighbors take different first-layer weights: the lower slot gets the earlier rank."""
    nl = np.zeros((1, 6, 4))
    nl[0, 0, :3] = (0, 1.0, 0)
    nl[0, 1, :3] = (1.0, 0, 0)       # same distance as slot 0
    nl[0, 2, :3] = (0, 0, 0.5)       # closest: rank 0
    K, H = 3, 2
    params = {"W1": np.array([[1.0, 0.0], [0.0, 2.0], [0.0, -3.0]]), "b1": np.zeros(H), "W2": np.eye(H), "b2": np.zeros(H),
              "W3": np.ones((H, 1)), "b3": np.zeros(1)}
    f, g = O.topk_mlp_model(nl, params, return_grad=True)
    # E = 1 * s2 + 2 * s0 - 3 * s1 with s = 1/r: slot 0 (rank 1) takes weight 2, slot 1 (rank 2) weight -3
    assert g[0, 0, 1] == pytest.approx(2.0 * -1.0, rel=1e-4) and g[0, 1, 0] == pytest.approx(-3.0 * -1.0, rel=1e-4)
    assert g[0, 2, 2] == pytest.approx(1.0 * -4.0, rel=1e-4)


def test_product_initializer_equals_the_oracle_copy():
    """hoomd_tf_amd/initializers.py duplicates oracle.make_mlp_params so that the product never imports oracle/:
    the two must stay the same function."""
    from hoomd_tf_amd.initializers import mlp_params
    for kw in (dict(seed=3), dict(seed=7, K=8, H1=16, H2=24), dict(seed=3, bias_scale=0.2)):
        a, b = mlp_params(**kw), O.make_mlp_params(**kw)
        assert sorted(a) == sorted(b)
        for k in a:
            np.testing.assert_array_equal(a[k], b[k])


def test_deferred_rebuild_rule():
    """standin.DeferredRebuildRule (host arithmetic of the multi-rank rebuild decision): fed the displacement of the
    PREVIOUS check, it asks for a rebuild before a linearly growing displacement can exceed r_buff / 2 at the next
    opportunity to act, never when nothing has been measured, and counts a measurement beyond the limit as dangerous."""
    from hoomd_tf_amd.standin import DeferredRebuildRule
    for growth in (0.013, 0.03, 0.049, 0.09):
        rule = DeferredRebuildRule(0.2)
        assert not rule.decide()
        d, k, fired = 0.0, 0, None
        while fired is None and k < 200:
            k += 1                      # check k: displacement is now k * growth; the value of check k - 1 arrives
            if k > 1:
                rule.push((k - 1) * growth)
            if rule.decide():
                fired = k
        assert fired is not None
        assert fired * growth <= 0.2 + 1e-12, (growth, fired)      # the list was never used beyond its guarantee
        assert (fired + 1) * growth > 0.2 - 2 * growth              # ... and not rebuilt more than two periods early
        assert rule.dangerous == 0
    rule = DeferredRebuildRule(0.2)
    rule.push(0.25)
    assert rule.decide() and rule.dangerous == 1
    rule.reset()
    assert not rule.decide()
