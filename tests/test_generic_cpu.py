"""The generic (torch autograd) route of compute_nlist_forces, SURVEY 8(f)-3, on CPU tensors:
same dataflow as simmodel.py:526-578 for an energy written in plain torch ops, checked
against the numpy oracle's hand-derived LJ forces/virial."""
import numpy as np
import pytest
import torch

import build_examples
from helpers import random_nlist
from oracle import htf_oracle as O


def _nl(seed=0, N=30, NN=12):
    rng = np.random.default_rng(seed)
    nl, _ = random_nlist(rng, N, NN, fill=0.7, rmin=0.9, rmax=2.8, dtype=np.float64)
    return nl


def test_autograd_route_matches_oracle_lj():
    import hoomd_tf_amd as htf
    nl = _nl()
    model = build_examples.TorchLJModel(12, virial=True, dtype=torch.float64)
    f, v = model([htf.Nlist(torch.from_numpy(nl)), None, None])
    ref_f, ref_v = O.lj_model(nl, virial=True)
    np.testing.assert_allclose(f.detach().numpy(), ref_f, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(v.detach().numpy(), ref_v, rtol=1e-9, atol=1e-9)


def test_autograd_route_energy_ranks_and_errors():
    import hoomd_tf_amd as htf
    nl = htf.Nlist(torch.from_numpy(_nl(1)))
    r = torch.norm(nl[:, :, :3], dim=2)
    e_pair = torch.where(r > 0, torch.exp(-r), torch.zeros_like(r))     # [N, NN]: summed per particle
    e_part = e_pair.sum(dim=1)                                           # [N]
    e_tot = e_pair.sum()                                                 # scalar: tiled to every particle
    f2 = htf.compute_nlist_forces(nl, e_pair).detach().numpy()
    f1 = htf.compute_nlist_forces(nl, e_part).detach().numpy()
    f0 = htf.compute_nlist_forces(nl, e_tot).detach().numpy()
    np.testing.assert_allclose(f2, f1, atol=1e-12)
    np.testing.assert_allclose(f0[:, :3], f1[:, :3], atol=1e-12)
    np.testing.assert_allclose(f0[:, 3], np.full(len(f0), float(e_tot)), atol=1e-12)
    # operators on the wrapper give torch tensors
    assert isinstance(nl[:, :, :3] * 2.0, torch.Tensor) and isinstance(nl[:, 0], torch.Tensor)
    # 'Did you put them in wrong order?' (simmodel.py:537-541)
    other = htf.Nlist(torch.from_numpy(_nl(2)))
    with pytest.raises(ValueError, match="wrong order"):
        htf.compute_nlist_forces(other, e_part)
    with pytest.raises(ValueError, match="wrong order"):
        htf.compute_nlist_forces(nl, torch.ones(3, dtype=torch.float64))


def test_mol_model_construction_rules():
    """test_tensorflow.py:722-727, 759-775, 806-820."""
    import hoomd_tf_amd as htf
    from hoomd_tf_amd.simmodel import _make_reverse_indices
    with pytest.raises(TypeError):
        build_examples.LJMolModel(MN=1, mol_indices=[1, 1, 4, 24], nneighbor_cutoff=10)
    with pytest.raises(ValueError):
        build_examples.LJMolModel(MN=2, mol_indices=[[0, 1, 2]], nneighbor_cutoff=10)

    class NoMol(htf.MolSimModel):
        def compute(self, nlist):
            return nlist

    class BadArgs(htf.MolSimModel):
        def mol_compute(self, nlist):
            return nlist

    with pytest.raises(AttributeError):
        NoMol(1, [[1]], 0)
    with pytest.raises(AttributeError):
        BadArgs(1, [[1]], 0)
    mi = [[1, 2, 0, 0, 0], [3, 0, 0, 0, 0], [4, 5, 7, 8, 9]]
    rmi = _make_reverse_indices(mi)
    assert rmi[0] == [0, 0] and rmi[1] == [0, 1] and rmi[2] == [1, 0] and rmi[8] == [2, 4]
    m = build_examples.LJMolModel(MN=3, mol_indices=[[0, 1, 2], [3, 4], [5, 6, 7], [8]], nneighbor_cutoff=8)
    assert m.mol_indices == [[1, 2, 3], [4, 5, 0], [6, 7, 8], [9, 0, 0]]
    # the molecule gather: [M, MN, NN, 4] with zeros in the padded slots, gradient back to nlist
    raw = _nl(3, N=9, NN=8)
    raw[:, :, 3] = 0  # "assume particle (w) is 0": the model takes the norm over all 4 components
    nl = htf.Nlist(torch.from_numpy(raw))
    pos = torch.zeros((9, 4), dtype=torch.float64)
    f = m([nl, pos, None])[0].detach().numpy()
    x = raw[:, :, :3]
    r = np.sqrt((x ** 2).sum(-1))
    live = r > 0
    rs = np.where(live, r, 1.0)
    dedr = np.where(live, 2.0 * (-12.0 * rs ** -13 + 6.0 * rs ** -7), 0.0)  # d/dr of 4/2 (r^-12 - r^-6)
    want = (2.0 * dedr / rs)[..., None] * x
    np.testing.assert_allclose(f[:, :3], want.sum(1), rtol=1e-9, atol=1e-9)
    e_tot = np.where(live, 2.0 * (rs ** -12 - rs ** -6), 0.0).sum()
    np.testing.assert_allclose(f[:, 3], np.full(9, e_tot), rtol=1e-12)  # scalar energy tiled (a13)


def test_find_molecules():
    import hoomd_tf_amd as htf

    class S:
        N = 6
        bonds = [(0, 3), (3, 5), (1, 2)]

    assert htf.find_molecules(S()) == [[0, 3, 5], [1, 2], [4]]


def test_pair_mask_is_a_tensor_everywhere_but_the_fast_path():
    """``htf.norm(nlist[:, :, :3], axis=2) < cut`` stays symbolic only for the product with a rinv polynomial (example 01);
    generic model code gets a bool tensor from every other use: torch functions, boolean operators for
    ``a < r < b`` shells, ``.to``, indexing, returning it as an output."""
    import hoomd_tf_amd as htf
    from hoomd_tf_amd.simmodel import PairMask
    raw = torch.from_numpy(_nl(3))
    nl = htf.Nlist(raw)
    r = htf.norm(nl[:, :, :3], axis=2)
    want = torch.sqrt((raw[:, :, :3] ** 2).sum(dim=2))
    m = r < 2.0
    assert isinstance(m, PairMask) and m.shape == want.shape and m.dtype == torch.bool
    assert torch.equal(torch.where(m, torch.ones_like(want), torch.zeros_like(want)), (want < 2.0).to(want.dtype))
    assert torch.equal(m.to(torch.float64), (want < 2.0).double()) and torch.equal(m.float(), (want < 2.0).float())
    assert torch.equal(~m, ~(want < 2.0))
    shell = (r < 2.0) & (r > 1.2)
    assert torch.equal(shell, (want < 2.0) & (want > 1.2))
    assert torch.equal((r < 2.0) * (r < 1.5), want < 1.5)
    assert torch.equal((r < 2.0) | (r > 2.5), (want < 2.0) | (want > 2.5))
    assert torch.equal(m[:, 0], (want < 2.0)[:, 0])
    assert torch.equal(m * want, (want < 2.0).float() * want) and torch.equal(want * m, (want < 2.0).float() * want)
    assert torch.equal(htf.cast(m).tensor(), want < 2.0)
    assert torch.equal(torch.logical_and(m, want > 1.0), (want < 2.0) & (want > 1.0))


def test_cast_mask_takes_part_in_arithmetic():
    """``tf.cast(r < cut, tf.float32)`` in model code: the cast mask still lowers when it multiplies a rinv
    polynomial, and is a float tensor for everything else -- ``1.0 - cast(mask)``, ``-mask``, ``mask ** 2``, ``mask * mask``."""
    import hoomd_tf_amd as htf
    from hoomd_tf_amd.simmodel import PairMask
    raw = torch.from_numpy(_nl(4))
    nl = htf.Nlist(raw)
    r = htf.norm(nl[:, :, :3], axis=2)
    want = (torch.sqrt((raw[:, :, :3] ** 2).sum(dim=2)) < 2.0).to(torch.float32)
    m = htf.cast(r < 2.0, torch.float32)
    assert isinstance(m, PairMask) and m.dtype == torch.float32
    assert torch.equal(1.0 - m, 1.0 - want) and torch.equal(m - 0.5, want - 0.5)
    assert torch.equal(-m, -want) and torch.equal(m ** 2, want ** 2)
    assert torch.equal(m * m, want) and (m * m).dtype == torch.float32
    assert torch.equal(m + m, 2 * want) and torch.equal(m / 2.0, want / 2.0)
    assert torch.equal(torch.sum(m), want.sum())
    assert torch.equal(htf.cast(r < 2.0, torch.float64) * 3.0, want.double() * 3.0)
    assert htf.cast(r < 2.0).dtype == torch.bool and htf.cast(r < 2.0, torch.bool).dtype == torch.bool


def test_neighbor_type_comparisons_are_tensors_to_torch_code():
    """``nlist[:, :, 3] == k`` (simmodel.py:661-693 masks by ``tf.equal(nlist[:, :, 3], type_j)`` and feeds the result to
    ordinary ops) stays symbolic for htf.where / htf.cast, and is the bool tensor it stands for to everything torch."""
    import hoomd_tf_amd as htf
    from hoomd_tf_amd.simmodel import PairCond
    raw = _nl(3)
    rng = np.random.default_rng(4)
    raw[:, :, 3] = rng.integers(0, 3, raw.shape[:2]) * (np.abs(raw[:, :, :3]).sum(axis=2) > 0)
    t = torch.from_numpy(raw)
    nl = htf.Nlist(t)
    want = t[:, :, 3] == 1
    c = nl[:, :, 3] == 1
    assert isinstance(c, PairCond) and c.shape == want.shape and c.dtype == torch.bool
    a, b = torch.ones_like(t[:, :, 0]), torch.zeros_like(t[:, :, 0])
    assert torch.equal(torch.where(c, a, b), torch.where(want, a, b))
    assert torch.equal(c.float(), want.float()) and torch.equal(c.to(torch.float64), want.double())
    assert torch.equal(a[c], a[want]) and torch.equal(t[:, :, 0][nl[:, :, 3] != 1], t[:, :, 0][~want])
    other = t[:, :, 0] > 0
    assert torch.equal(c & other, want & other) and torch.equal(other & c, want & other)
    assert torch.equal(c | other, want | other) and torch.equal(~c, ~want) and torch.equal(c ^ other, want ^ other)
    assert torch.equal(torch.logical_and(c, other), want & other) and int(c.sum()) == int(want.sum())
    assert torch.equal((nl[:, :, 3] < 2).float(), (t[:, :, 3] < 2).float()) and torch.equal(c * 2.0, want.float() * 2.0)
    # htf.where with tensor branches falls through to torch; with constants it stays a traced expression
    assert torch.equal(htf.where(c, a, b), torch.where(want, a, b))
    assert not isinstance(htf.where(c, 1.0, 0.0), torch.Tensor)
    # a per-type masked energy written the reference's way (masked_nlist by hand) on the autograd route
    r = torch.norm(nl[:, :, :3], dim=2)
    e = torch.where(c & (r > 0), torch.exp(-r), torch.zeros_like(r))
    f = htf.compute_nlist_forces(nl, e).detach().numpy()
    nl2 = htf.Nlist(t.clone())
    r2 = torch.norm(nl2[:, :, :3], dim=2)
    e2 = torch.where(want & (r2 > 0), torch.exp(-r2), torch.zeros_like(r2))
    f2 = htf.compute_nlist_forces(nl2, e2).detach().numpy()
    assert np.abs(f2[:, :3]).max() > 0
    np.testing.assert_array_equal(f, f2)
