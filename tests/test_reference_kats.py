"""Pin the CPU oracle against every known answer the reference's own tests state for
this path.  The reference tests need HOOMD + TF to *run*; each states
a checkable answer that is reproduced here on the same system.
"""
import numpy as np
import pytest

from helpers import analytic_lj, bcc_lattice, brute_nlist, min_image_np, sq_lattice
from oracle import htf_oracle as O


def reference_test_compute_forces(pos, L, rcut):
    """The numpy oracle that lives in the reference's test file
    (test_tensorflow.py:20-35): F_i = -sum_j r_ij/|r_ij| over pairs with
    |r| <= rcut, r = min_image(pos[j] - pos[i])."""
    N = len(pos)
    forces = np.zeros((N, 3))
    for i in range(N):
        for j in range(i + 1, N):
            r = min_image_np(pos[j] - pos[i], L)
            rd = np.sqrt(np.sum(r ** 2))
            if rd <= rcut:
                f = -r / rd
                forces[i] += f
                forces[j] -= f
    return forces


def _system(n, a, jitter, seed, rcut, shuffle=True, dtype=np.float64):
    pos, L = sq_lattice(n, a)
    rng = np.random.default_rng(seed)
    pos[:, :2] += jitter * rng.standard_normal((n * n, 2))
    pos = pos.astype(dtype)
    nn, head, nl = brute_nlist(pos, L, rcut + 0.4, shuffle_seed=seed if shuffle else None)
    return pos, L, O.make_box(L, dtype=dtype), nn, head, nl


@pytest.mark.parametrize("batch", [0, 4])
@pytest.mark.parametrize("hdt", [np.float64, np.float32])
def test_force_overwrite(batch, hdt):
    """test_tensorflow.py:81-129 test_force_overwrite[_batched]: SimplePotential(NN=8),
    3x3 sq a=4, r_cut=5, batch_size None / 4, atol 1e-5."""
    N, NN, rcut = 9, 8, 5.0
    pos, L, box, nn, head, nl = _system(3, 4.0, 0.15, 2, rcut, dtype=hdt)
    types = np.zeros(N, dtype=np.int32)
    f, _ = O.compute_forces(pos, types, nn, head, nl, box, rcut, NN, O.simple_potential,
                            batch_size=batch)
    ref = reference_test_compute_forces(pos.astype(np.float64), L, rcut)
    np.testing.assert_allclose(f[:, :3], ref, atol=1e-5)
    assert np.all(f[:, 3] == 0)


def test_prepare_neighbors_loops_vs_vectorised():
    pos, L, box, nn, head, nl = _system(5, 1.3, 0.1, 7, 2.0)
    types = np.arange(25) % 3
    for NN in (4, 8, 32):
        a = O.prepare_neighbors_loops(pos, types, nn, head, nl, box, 2.0, NN)
        b = O.prepare_neighbors(pos, types, nn, head, nl, box, 2.0, NN)
        np.testing.assert_array_equal(a, b)
    a = O.prepare_neighbors_loops(pos, types, nn, head, nl, box, 2.0, 8, offset=10, batch_size=7)
    b = O.prepare_neighbors(pos, types, nn, head, nl, box, 2.0, 8, offset=10, batch_size=7)
    np.testing.assert_array_equal(a, b)


def test_lj_forces():
    """test_tensorflow.py:335-382 test_lj_forces: LJModel(32), 5x5 sq a=3, r_cut=5:
    per-particle force == HOOMD pair.lj(eps=1, sig=1, r_cut=5), atol 1e-5, forces
    non-trivial (> 1e-4)."""
    N, NN, rcut = 25, 32, 5.0
    for seed in range(10):  # the reference checks 10 consecutive NVT frames
        pos, L, box, nn, head, nl = _system(5, 3.0, 0.08, seed, rcut)
        types = np.zeros(N, dtype=np.int32)
        f, _ = O.compute_forces(pos, types, nn, head, nl, box, rcut, NN, O.lj_model)
        F, E, _ = analytic_lj(pos, L, rcut)
        np.testing.assert_allclose(f[:, :3], F, atol=1e-5)
        assert np.all(np.sum(F ** 2, axis=1) > 1e-4 ** 2)
        # test_force_output (:400-431): the energy column equals HOOMD's per-particle
        # LJ energy too (MSE over all four columns < 1e-5).
        assert np.mean((np.concatenate([F, E[:, None]], 1) - f) ** 2) < 1e-5
        np.testing.assert_allclose(f[:, 3], E, atol=1e-5)


def test_lj_pressure():
    """test_tensorflow.py:619-671: LJVirialModel(32, virial=True), 3x3 sq a=4,
    r_cut=5: TF virial [:, 0:2] (xx, xy) == HOOMD per-particle virial, atol 1e-5."""
    N, NN, rcut = 9, 32, 5.0
    pos, L, box, nn, head, nl = _system(3, 4.0, 0.1, 1, rcut)
    types = np.zeros(N, dtype=np.int32)
    f, vir = O.compute_forces(pos, types, nn, head, nl, box, rcut, NN,
                              lambda x: O.lj_model(x, virial=True), virial=True)
    _, _, V = analytic_lj(pos, L, rcut)
    v6 = vir.reshape(6, N).T
    np.testing.assert_allclose(v6[:, 0:2], V[:, 0:2], atol=1e-5)
    # all six agree here because every pair sits in the attractive branch (r ~ 4);
    # for repulsive pairs the reference's |F| makes the sign differ
    np.testing.assert_allclose(v6, V, atol=1e-5)


def test_nlist_count():
    """test_tensorflow.py:559-579: 3x3 sq a=4, r_cut=5 -> exactly 4 neighbours each
    (full list, not half)."""
    pos, L, box, nn, head, nl = _system(3, 4.0, 0.0, 0, 5.0)
    buf = O.prepare_neighbors(pos, np.zeros(9, np.int32), nn, head, nl, box, 5.0, 32)
    ncount = np.sum(np.sum(buf ** 2, axis=2) > 0.1, axis=1)
    assert np.min(ncount) == 4 and np.max(ncount) == 4


def test_overflow():
    """test_tensorflow.py:830-848: LJModel(4, check_nlist=True), 8x8 sq a=4,
    r_cut=10 -> 'Neighbor list is full!' (count of dx>0 entries not < NN)."""
    pos, L = sq_lattice(8, 4.0)
    rng = np.random.default_rng(1)
    pos[:, :2] += 0.05 * rng.standard_normal((64, 2))
    nn, head, nl = brute_nlist(pos, L, 10.0, shuffle_seed=3)
    buf = O.prepare_neighbors(pos, np.zeros(64, np.int32), nn, head, nl, O.make_box(L), 10.0, 4)
    assert not (O.check_nlist_count(buf) < 4)
    # and a roomy list passes
    buf = O.prepare_neighbors(pos, np.zeros(64, np.int32), nn, head, nl, O.make_box(L), 10.0, 64)
    assert O.check_nlist_count(buf) < 64


def test_access_types():
    """test_tensorflow.py:46-70 test_access: 3 types survive into nlist[...,3]."""
    pos, L = sq_lattice(5, 2.0)
    types = np.arange(25) % 3
    nn, head, nl = brute_nlist(pos, L, 3.0)
    buf = O.prepare_neighbors(pos, types, nn, head, nl, O.make_box(L), 3.0, 32)
    real = np.sum(buf[..., :3] ** 2, axis=2) > 0
    assert len(np.unique(buf[..., 3][real].astype(int))) == 3


def test_compute_nlist_kats():
    """test_utils.py:187-270: 10 particles on the diagonal."""
    N = 10
    positions = np.tile(np.arange(N, dtype=np.float32).reshape(-1, 1), (1, 3))
    box = [100., 100., 100.]
    nl = O.compute_nlist(positions, 100., 9, box, sorted=True)
    np.testing.assert_array_almost_equal(nl[0, 0, :], [1, 1, 1, 1])
    np.testing.assert_array_almost_equal(nl[-1, -1, :], [-9, -9, -9, 0])
    ext = np.concatenate([positions, np.zeros((N, 1), np.float32)], axis=1)
    nl = O.compute_nlist(ext, 100., 9, box, sorted=True, return_types=True)
    np.testing.assert_array_almost_equal(nl[0, 0, :], [1, 1, 1, 0])
    nl = O.compute_nlist(positions, 5.5, 9, box, sorted=True)
    np.testing.assert_array_almost_equal(nl[0, 0, :], [1, 1, 1, 1])
    np.testing.assert_array_almost_equal(nl[-1, -1, :], [0, 0, 0, 0])


def test_nlist_compare():
    """test_utils.py:401-430: the plugin's pair vectors and utils.compute_nlist give
    the same multiset of r per particle (bcc 4^3 a=4, r_cut=5, NN=32, 5 decimals)."""
    pos, L = bcc_lattice(4, 4.0)
    rng = np.random.default_rng(5)
    pos = pos + 0.05 * rng.standard_normal(pos.shape)
    nn, head, nl = brute_nlist(pos, L, 5.4, shuffle_seed=1)
    buf = O.prepare_neighbors(pos, np.zeros(len(pos), np.int32), nn, head, nl, O.make_box(L), 5.0, 32)
    r = np.sqrt(np.sum(buf[..., :3] ** 2, axis=2))
    cn = O.compute_nlist(pos, 5.0, 32, L)
    cr = np.sqrt(np.sum(cn[..., :3] ** 2, axis=2))
    np.testing.assert_array_almost_equal(np.sort(r, axis=1), np.sort(cr, axis=1), decimal=5)


def test_rbf_shape():
    """test_layers.py:24-31: RBFExpansion(0,2,10) on safe_norm(ones(10,6,3)) ->
    (10,6,10).  Shape only upstream; values are 'parity unpinned'."""
    r = O.safe_norm(np.ones((10, 6, 3), np.float32), axis=2)
    out = O.rbf_expansion(r, 0, 2, 10)
    assert out.shape == (10, 6, 10)


def test_eds_converges():
    """test_utils.py:447-461 (statistical pin): EDSModel(set_point=4): harmonic
    (cv-5)^2 + alpha*cv; equilibrium cv = 5 - alpha/2, so alpha -> 2 gives cv 4.
    Driven here by overdamped relaxation + noise instead of HOOMD NVE."""
    eds = O.EDSLayer(4.0, 5, 1 / 5)
    rng = np.random.default_rng(2)
    cv, cvs = 5.0, []
    for _ in range(4000):
        a = float(eds(cv))
        cv += 0.2 * (-(2 * (cv - 5) + a)) + 0.05 * rng.standard_normal()
        cvs.append(cv)
    assert np.isfinite(a)
    assert (np.mean(cvs[2000:]) - 4) ** 2 < 0.5
