"""Randomised cross-check of the three traced modes of htf_compute_forces (two kernels / one kernel
with the tensor / one kernel without it) over system size, density, NN (overflowing and not),
precision, batching, row ranges, virial and potential, with the particles moving between calls so
that the per-row live counts (delta zero-fill) change from step to step."""
import os

import numpy as np
import pytest
import torch

from helpers import brute_nlist

pytestmark = pytest.mark.gpu


def _potentials(htf):
    return [htf.Potential.lj(), htf.Potential.wca(1.0), htf.Potential.rinv_poly([1.0, -0.5], [12, 6]),
            htf.Potential.lj_param(1.1, 0.95), htf.Potential.simple()]


@pytest.mark.parametrize("seed", range(int(os.environ.get("HTF_FUZZ_SEEDS", "12"))))
def test_modes_agree_on_random_systems(htf, cuda, seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(3, 8))
    dim3 = bool(rng.integers(0, 2))
    a = float(rng.uniform(1.0, 1.6))
    if dim3:
        g = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
        L = np.array([n * a] * 3)
    else:
        g = np.stack(np.meshgrid(np.arange(n * 2), np.arange(n * 2), indexing="ij"), -1).reshape(-1, 2).astype(np.float64)
        g = np.concatenate([g, np.zeros((len(g), 1))], axis=1)
        L = np.array([2 * n * a, 2 * n * a, 4.0])
    pos = (g + 0.5) * a - L / 2
    pos[:, :3 if dim3 else 2] += 0.1 * a * rng.standard_normal((len(pos), 3 if dim3 else 2))
    pos -= np.round(pos / L) * L
    N = len(pos)
    r_cut = float(rng.uniform(1.3, min(2.6, 0.45 * L[:2].min())))
    NN = int(rng.choice([4, 8, 24, 64, 128, 160]))
    tdt = torch.float64 if rng.integers(0, 2) else torch.float32
    batch = int(rng.choice([0, 0, 7, 33]))
    pots = _potentials(htf)
    pot = pots[int(rng.integers(0, len(pots)))]
    virial = bool(rng.integers(0, 2)) and pot.kind != htf._lib.POT_SIMPLE
    types = np.zeros(N, dtype=np.int32)
    outs = {}
    moves = [0.05 * a * rng.standard_normal(pos.shape) * ([1, 1, 1] if dim3 else [1, 1, 0]) for _ in range(2)]
    for mode in (0, 2, 1):
        p = pos.copy()
        ctx = htf.Context(r_cut=r_cut, nneighs=NN, batch_size=batch, scalar_dtype=tdt, virial=virial, max_n=N, fused=mode)
        ctx.set_potential(pot)
        res = []
        for step in range(3):
            nn, head, nl = brute_nlist(p, L, r_cut + 0.3, shuffle_seed=seed)
            p4 = htf.ops.stuff_types(torch.from_numpy(p).to(cuda), torch.from_numpy(types).to(cuda), tdt)
            dnn, dhead, dnl = (torch.from_numpy(x.astype(np.int32)).to(cuda) for x in (nn, head, nl))
            force = torch.zeros((N, 4), dtype=tdt, device=cuda)
            vir = torch.zeros(6 * N, dtype=tdt, device=cuda) if virial else None
            arr = ctx.make_arrays(p4, N, dnn, dhead, dnl, htf._lib.make_box(np.array([-L / 2, L / 2, [0, 0, 0]])), force, vir, N)
            if batch == 0 and step == 1 and N > 20:  # a step computed in two row ranges
                n1 = int(rng.integers(1, N - 1))
                ctx.compute_forces(step, arr, rows=(0, n1))
                ctx.compute_forces(step, arr, rows=(n1, N - n1))
            else:
                ctx.compute_forces(step, arr)
            torch.cuda.synchronize()
            nb = ctx.nlist_buffer(N if batch == 0 else batch, cuda).clone() if mode != 1 else None
            res.append((force.clone(), vir.clone() if virial else None, nb))
            if step < 2:
                p = p + moves[step]
                p -= np.round(p / L) * L
        outs[mode] = res
    for step in range(3):
        f0, v0, t0 = outs[0][step]
        f2, v2, t2 = outs[2][step]
        f1, v1, _ = outs[1][step]
        assert torch.equal(t0, t2), "pair-vector tensor differs between the two-kernel and the one-kernel mode"
        scale = max(1.0, float(f0.abs().max()))
        assert float((f2 - f0).abs().max()) <= 5e-5 * scale, (seed, step, float((f2 - f0).abs().max()), scale)
        assert float((f1 - f2).abs().max()) <= 5e-5 * scale
        if virial:
            vs = max(1.0, float(v0.abs().max()))
            assert float((v2 - v0).abs().max()) <= 1e-4 * vs and float((v1 - v2).abs().max()) <= 1e-4 * vs


# --------------------------------------------------------------------------- crafted row lengths, every kernel form
_LENGTHS = [0, 1, 2, 63, 64, 65, 100, 127, 128, 129, 150, 191, 192, 193, 230]


def _crafted_lists(pos, L, r_cut, rng):
    """HOOMD-layout lists whose row LENGTHS are drawn from the edges of the kernels' trip structure (64-slot trips, the
    128-entry straight-line part, the 192-entry limit of the merged tail): a row's true neighbors within r_cut + 0.3 first
    (shuffled), cut short or padded with far particles, which prepareNeighbors drops again (.cc:359)."""
    nn0, head0, nl0 = brute_nlist(pos, L, r_cut + 0.3, shuffle_seed=5)
    N = len(pos)
    rows = []
    for i in range(N):
        want = int(rng.choice(_LENGTHS))
        mine = nl0[head0[i]:head0[i] + nn0[i]].astype(np.int64)
        if want <= len(mine):
            row = mine[:want]
        else:
            far = rng.integers(0, N, size=want - len(mine))
            row = rng.permutation(np.concatenate([mine, far]))
        rows.append(row[row != i] if want else row)
    nn = np.array([len(r) for r in rows], dtype=np.uint32)
    head = np.concatenate([[0], np.cumsum(nn)[:-1]]).astype(np.uint32)
    flat = np.concatenate(rows).astype(np.uint32)
    return nn, head, flat


@pytest.mark.parametrize("hdt", [np.float32, np.float64])
@pytest.mark.parametrize("NN", [64, 128])
def test_crafted_row_lengths_body(htf, cuda, hdt, NN):
    """One-kernel step on lists with crafted row lengths against the oracle: tensor bit for bit (overflowing rows with the
    reference's slot wrap), forces within the stated tolerance, tensor written or not.  Run as is (the default forms) and, by
    test_crafted_row_lengths_every_form, in child processes with each rows-per-wave form forced."""
    from oracle import htf_oracle as O

    def assert_forces_close(got, ref, _nl):
        # structural errors (a wrong slot, a dropped tail) are O(1); fp32 summation order is 1e-7 of the largest pair term
        scale = max(1.0, float(np.abs(ref).max()))
        err = np.abs(got - ref)
        assert np.all(np.isfinite(got)) and float(err.max()) <= 2e-5 * scale, (float(err.max()), scale)

    rng = np.random.default_rng(77)
    n, a = 9, 1.12  # 729 particles, 9 = 3 * 3: groups of 2, 3 and 4 rows all end ragged somewhere
    g = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
    L = np.array([n * a] * 3)
    pos = (g + 0.5) * a - L / 2 + 0.06 * a * rng.standard_normal((n ** 3, 3))
    pos -= np.round(pos / L) * L
    pos = pos.astype(hdt).astype(np.float64)
    N, r_cut = len(pos), 3.0
    nn, head, nl = _crafted_lists(pos, L, r_cut, rng)
    types = np.zeros(N, dtype=np.int32)
    tdt = torch.float64 if hdt == np.float64 else torch.float32
    p4 = htf.ops.stuff_types(torch.from_numpy(pos).to(cuda), torch.from_numpy(types).to(cuda), tdt)
    dnn, dhead, dnl = (torch.from_numpy(x.astype(np.int32)).to(cuda) for x in (nn, head, nl))
    box = O.make_box(L, dtype=hdt)
    want = O.prepare_neighbors(pos.astype(hdt), types, nn, head, nl, box, r_cut, NN).astype(np.float32)
    for pot, model in ((htf.Potential.lj(), O.lj_model), (htf.Potential.wca(1.0), lambda t: O.wca_model(t, 1.0))):
        pv = torch.full((N, NN, 4), 7.0, device=cuda)
        f_t = htf.ops.fused_forces(pot, p4, dnn, dhead, dnl, box, r_cut, NN, pair_vectors=pv)
        f_n = htf.ops.fused_forces(pot, p4, dnn, dhead, dnl, box, r_cut, NN)
        assert np.array_equal(pv.cpu().numpy(), want), "pair-vector tensor"
        ref = model(want.astype(np.float64))
        for f in (f_t, f_n):
            assert_forces_close(f.cpu().numpy().astype(np.float64), ref, want.astype(np.float64))
        assert float((f_t - f_n).abs().max()) <= 1e-4 * max(1.0, float(f_t.abs().max()))
    # the same through the context (delta zero-fill of the previous step's rows: second call after the lists change)
    ctx = htf.Context(r_cut=r_cut, nneighs=NN, scalar_dtype=tdt, max_n=N, fused=2)
    ctx.set_potential(htf.Potential.lj())
    force = torch.zeros((N, 4), dtype=tdt, device=cuda)
    for rep in range(2):
        nn2, head2, nl2 = _crafted_lists(pos, L, r_cut, np.random.default_rng(100 + rep))
        d2 = [torch.from_numpy(x.astype(np.int32)).to(cuda) for x in (nn2, head2, nl2)]
        ctx.compute_forces(rep, ctx.make_arrays(p4, N, d2[0], d2[1], d2[2], box, force))
        w2 = O.prepare_neighbors(pos.astype(hdt), types, nn2, head2, nl2, box, r_cut, NN).astype(np.float32)
        assert np.array_equal(ctx.nlist_buffer(N, cuda).cpu().numpy(), w2), "context tensor, call %d" % rep
        assert_forces_close(force.cpu().numpy().astype(np.float64), O.lj_model(w2.astype(np.float64)), w2.astype(np.float64))


@pytest.mark.parametrize("form", ["HTF_FUSED_TAILS=2", "HTF_FUSED_TAILS=3", "HTF_FUSED_TAILS=4", "HTF_FUSED_TAILS=0", "HTF_FUSED_ROWS=4"])
def test_crafted_row_lengths_every_form(htf, cuda, form):
    """The defaults pick a rows-per-wave form by batch size; a child process per form runs the crafted-length test with that
    form forced at this (small) size -- the merged-tail kernels the headline configurations run included."""
    import subprocess
    import sys
    from helpers import variants_env
    k, v = form.split("=")
    env = variants_env(**{k: v})   # (the switches exist in the variants build only)
    if k == "HTF_FUSED_ROWS":
        env["HTF_FUSED_TAILS"] = "0"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "crafted_row_lengths_body"],
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("hdt", [np.float32, np.float64])
@pytest.mark.parametrize("NN", [64, 128])
def test_crafted_row_lengths_c4_body(htf, cuda, hdt, NN):
    """Config C4's one-kernel sweep (htf_build_eval_forces2: tensor, LJ + Gaussian forces, CV partials, compute_rdf histogram) on
    the crafted row lengths against htf_build_pair_vectors -> htf_eval_forces2 and the oracle's tensor: tensor and histogram
    bit for bit (overflowing rows un-binned and redone with the slot wrap), forces and CV to summation order."""
    from oracle import htf_oracle as O
    rng = np.random.default_rng(78)
    n, a = 9, 1.12
    g = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
    L = np.array([n * a] * 3)
    pos = (g + 0.5) * a - L / 2 + 0.06 * a * rng.standard_normal((n ** 3, 3))
    pos -= np.round(pos / L) * L
    pos = pos.astype(hdt).astype(np.float64)
    N, r_cut = len(pos), 3.0
    nn, head, nl = _crafted_lists(pos, L, r_cut, rng)
    types = np.zeros(N, dtype=np.int32)
    tdt = torch.float64 if hdt == np.float64 else torch.float32
    p4 = htf.ops.stuff_types(torch.from_numpy(pos).to(cuda), torch.from_numpy(types).to(cuda), tdt)
    dnn, dhead, dnl = (torch.from_numpy(x.astype(np.int32)).to(cuda) for x in (nn, head, nl))
    box = O.make_box(L, dtype=hdt)
    want = O.prepare_neighbors(pos.astype(hdt), types, nn, head, nl, box, r_cut, NN).astype(np.float32)
    for base in (htf.Potential.lj(), htf.Potential.wca(1.0)):
        gauss = htf.Potential.gauss(1.1, 0.05, 1.0)
        pv = htf.ops.build_pair_vectors(p4, dnn, dhead, dnl, box, r_cut, NN)
        assert np.array_equal(pv.cpu().numpy(), want)
        hist0 = torch.zeros(102, dtype=torch.int32, device=cuda)
        part0 = torch.zeros(htf.ops.num_partials(N, NN), device=cuda)
        fa0, fb0 = htf.ops.eval_forces2(base, gauss, pv, partials=part0, rdf=(0.0, 3.5, hist0), out_dtype=tdt)
        hist1 = torch.zeros(102, dtype=torch.int32, device=cuda)
        part1 = torch.zeros(htf.ops.num_partials_fused(N), device=cuda)
        pv1 = torch.full_like(pv, 3.0)
        fa1, fb1 = htf.ops.build_eval_forces2(base, gauss, p4, dnn, dhead, dnl, box, r_cut, NN, partials=part1,
                                              rdf=(0.0, 3.5, hist1), pair_vectors=pv1)
        assert np.array_equal(pv1.cpu().numpy(), want), "tensor"
        assert torch.equal(hist1, hist0) and int(hist0.sum()) == N * NN, "histogram"
        for x, y in ((fa1, fa0), (fb1, fb0)):
            assert float((x - y).abs().max()) <= 2e-5 * max(1.0, float(y.abs().max()))
        assert abs(float(part1.sum()) - float(part0.sum())) <= 2e-5 * max(1.0, abs(float(part0.sum())))
        fa2, fb2 = htf.ops.build_eval_forces2(base, gauss, p4, dnn, dhead, dnl, box, r_cut, NN)  # nothing stored, no histogram
        assert float((fa2 - fa1).abs().max()) <= 2e-6 * max(1.0, float(fa1.abs().max()))
        assert float((fb2 - fb1).abs().max()) <= 2e-6 * max(1.0, float(fb1.abs().max()))


@pytest.mark.parametrize("form", ["HTF_FUSED2_ROWS=2", "HTF_FUSED2_ROWS=4", "HTF_FUSED2_ROWS=0", "HTF_FUSED2_COMPACT=0"])
def test_crafted_row_lengths_c4_every_form(htf, cuda, form):
    """The C4 sweep's forms (two / four rows per wave with merged tails; the compacting persistent kernel; the same without
    compaction), each forced in a child process."""
    import subprocess
    import sys
    from helpers import variants_env
    k, v = form.split("=")
    env = variants_env(**{k: v})
    if k == "HTF_FUSED2_COMPACT":
        env["HTF_FUSED2_ROWS"] = "0"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "crafted_row_lengths_c4_body"],
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
