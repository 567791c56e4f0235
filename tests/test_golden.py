"""Committed golden vectors (tests/golden/golden_v1.npz, made by tests/golden/make_golden.py):
CPU: the oracle still reproduces them; GPU: the HIP path reproduces them."""
import os

import numpy as np
import pytest
import torch

from helpers import ROOT
from oracle import htf_oracle as O


@pytest.fixture(scope="module")
def G():
    with np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz")) as z:
        return {k: z[k] for k in z.files}


def test_oracle_reproduces_golden(G):
    box = O.make_box(G["L"])
    for NN in (16, 64):
        pv = O.prepare_neighbors(G["pos"], G["types"], G["n_neigh"], G["head_list"], G["nlist"], box, 2.5, NN)
        np.testing.assert_array_equal(pv, G["pv64_NN%d" % NN])
    x = G["pv32_NN64"].astype(np.float64)
    f, v = O.lj_model(x, virial=True)
    np.testing.assert_allclose(f, G["lj_force"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(v, G["lj_virial"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(O.wca_model(x, 0.5), G["wca05_force"], rtol=1e-12, atol=1e-12)
    params = {k[4:]: G[k] for k in G if k.startswith("mlp_") and k[4:] in ("W1", "b1", "W2", "b2", "W3", "b3")}
    np.testing.assert_allclose(O.pair_mlp_model(x, params, 0.0, 3.0, "tanh"), G["mlp_tanh_force"], rtol=1e-10, atol=1e-10)
    eds = O.EDSLayer(4.0, 5, 0.2)
    np.testing.assert_allclose([eds(c) for c in G["eds_cv"]], G["eds_alpha"], rtol=1e-6, atol=1e-7)
    assert np.sum(G["pv64_NN16"][..., :3] ** 2, axis=2).min() > 0  # NN=16 rows are full: the wrap case


@pytest.mark.gpu
def test_hip_path_reproduces_golden(G, htf, cuda):
    from test_gpu_parity import assert_forces_close
    for hdt, tdt, key in ((np.float64, torch.float64, "pv64"), (np.float32, torch.float32, "pv32")):
        p4 = htf.ops.stuff_types(torch.from_numpy(G["pos"].astype(hdt)).to(cuda), torch.from_numpy(G["types"]).to(cuda), tdt)
        args = (p4, torch.from_numpy(G["n_neigh"].astype(np.int32)).to(cuda),
                torch.from_numpy(G["head_list"].astype(np.int32)).to(cuda),
                torch.from_numpy(G["nlist"].astype(np.int32)).to(cuda), O.make_box(G["L"], dtype=hdt), 2.5)
        for NN in (16, 64):
            pv = htf.ops.build_pair_vectors(*args, NN, out_dtype=tdt)
            np.testing.assert_array_equal(pv.cpu().numpy(), G["%s_NN%d" % (key, NN)])
    x = torch.from_numpy(G["pv32_NN64"]).to(cuda)
    x64 = G["pv32_NN64"].astype(np.float64)
    cond = np.abs(2 * O._grad_from_dEds(*_lj_dEds(x64))).sum(axis=(1, 2))
    f, v = htf.ops.eval_forces(htf.Potential.lj(), x, virial=True)
    assert_forces_close("golden_lj", f.cpu().numpy(), G["lj_force"], cond)
    assert_forces_close("golden_ljv", v.cpu().numpy(), G["lj_virial"], 3 * cond)
    assert_forces_close("golden_wca05", htf.ops.eval_forces(htf.Potential.wca(0.5), x).cpu().numpy(), G["wca05_force"], atol=2e-5)
    assert_forces_close("golden_wca10", htf.ops.eval_forces(htf.Potential.wca(1.0), x).cpu().numpy(), G["wca10_force"], atol=1e-4, rtol=5e-5)
    assert_forces_close("golden_rinv", htf.ops.eval_forces(htf.Potential.rinv_poly([1.0], [1]), x).cpu().numpy(), G["rinv_force"])
    simple = htf.ops.eval_forces(htf.Potential.simple(), x).cpu().numpy()
    assert_forces_close("golden_simple", simple[:, :3], G["simple_force"])
    params = {k: G["mlp_" + k] for k in ("W1", "b1", "W2", "b2", "W3", "b3")}
    for act in ("tanh", "linear"):
        f = htf.ops.eval_forces(htf.Potential.pair_mlp(params, 0.0, 3.0, activation=act), x)
        assert_forces_close("golden_mlp_" + act, f.cpu().numpy(), G["mlp_%s_force" % act], atol=1e-4, rtol=1e-4)
    e = torch.from_numpy(G["edge_nlist"]).to(cuda)
    np.testing.assert_allclose(htf.ops.nlist_rinv(e).cpu().numpy(), G["edge_rinv"], rtol=1e-6)
    assert_forces_close("golden_edge_wca", htf.ops.eval_forces(htf.Potential.wca(0.5), e).cpu().numpy(), G["edge_wca05_force"], atol=1e-4, rtol=1e-4)
    r = htf.safe_norm(torch.ones((10, 6, 3), device=cuda), axis=2)
    np.testing.assert_allclose(htf.RBFExpansion(0, 2, 10)(r).cpu().numpy(), G["rbf_ones"], rtol=2e-6, atol=1e-7)
    eds = htf.EDSLayer(4.0, 5, 0.2, device=cuda)
    got = np.array([float(eds(float(c))) for c in G["eds_cv"]])
    np.testing.assert_allclose(got, G["eds_alpha"], rtol=2e-5, atol=2e-5)
    rdf, _ = htf.compute_rdf(x, [0, 3.5], nbins=20)
    np.testing.assert_allclose(rdf.cpu().numpy(), G["rdf_0_35"], rtol=1e-4)
    tt = torch.from_numpy(G["types"].astype(np.float32)).to(cuda)
    rdf, _ = htf.compute_rdf(x, [0, 3.5], tt, nbins=20, type_i=0, type_j=1)
    np.testing.assert_allclose(rdf.cpu().numpy(), G["rdf_typed"], rtol=1e-4)


def _lj_dEds(x64):
    s, t, rp, cond = O._rinv_and_grad_factor(x64)
    inv_r6 = s ** 6
    return 2.0 * (2.0 * inv_r6 - 1.0) * (6.0 * s ** 5), s, t, rp, cond
