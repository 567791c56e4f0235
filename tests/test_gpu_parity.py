"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the
same seeded inputs.  Run on the MI355X box with ``pytest -m gpu``.

Stated tolerances (DESIGN.md "Parity"):
  * pair vectors: BIT-EXACT (the build kernel is compiled -ffp-contract=off and
    performs the oracle's operations in the oracle's order);
  * forces / energies / virials (fp32 compute): |d| <= 1e-5 + 2e-5 * |ref| against
    the fp64 oracle evaluated on the same fp32 inputs, plus a condition term
    2e-6 * sum_j |f_ij| for rows where large pair forces cancel (fp32 summation
    order differs between a wave reduction and TF's reduce_sum).
"""
import json
import os

import numpy as np
import pytest
import torch

from helpers import ROOT, brute_nlist, fcc_lattice, random_nlist, sq_lattice
from oracle import htf_oracle as O

pytestmark = pytest.mark.gpu

STATS = {}


def _record(name, **kw):
    STATS[name] = {k: float(v) for k, v in kw.items()}
    import multiprocessing
    if multiprocessing.current_process().name != "MainProcess" or os.environ.get("HTF_STATS_NO_FILE"):
        return   # a rank process / child of a test: its records travel back to the test, which writes the file
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_stats.json"), "w") as f:
        json.dump(STATS, f, indent=1, sort_keys=True)


def _cond_scale(nl64, model_pair_forces):
    """sum_j |f_ij| per row: the condition number scale of the row sum."""
    return np.abs(model_pair_forces).sum(axis=(1, 2))


def assert_forces_close(name, got, ref, cond=None, atol=1e-5, rtol=2e-5, ctol=2e-6, cancelling_rows=None):
    """SURVEY 8(c)'s bound, |d| <= atol + rtol |ref|, asserted as it stands.  The condition term
    ctol * sum_j |f_ij| is added ONLY for call sites that name why they need it (``cancelling_rows``:
    rows whose pair forces are orders of magnitude larger than their sum, where the fp32 summation order
    of a 64-lane wave reduction and of the oracle's fp64 sum differ by more than the strict bound).
    Both ratios are recorded (gpurun_out/parity_stats.json)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape
    assert np.all(np.isfinite(got)), name
    strict = atol + rtol * np.abs(ref)
    err = np.abs(got - ref)
    rec = dict(max_abs_err=err.max(), max_ratio_strict=(err / strict).max(), max_ref=np.abs(ref).max())
    bound = strict
    if cond is not None:
        with_cond = strict + ctol * np.asarray(cond).reshape(-1, *([1] * (ref.ndim - 1)))
        rec["max_ratio_with_condition_term"] = (err / with_cond).max()
        if cancelling_rows:
            bound = with_cond
            rec["condition_term_used"] = 1.0
    _record(name, **rec)
    assert np.all(err <= bound), "%s: worst err/bound = %.3g (max abs err %.3g)%s" % (
        name, (err / bound).max(), err.max(), "" if cancelling_rows or cond is None else
        "; with the condition term: %.3g" % rec["max_ratio_with_condition_term"])


# The synthetic fixtures below (_nlist_case / random_nlist) place ~95 neighbors per row uniformly in direction with
# contacts down to r = 0.55-0.9: single pair forces reach 10^2-10^3 while the row sums are O(1-10).  fp32 row sums
# taken in a different ORDER (a 16- or 64-lane butterfly here, reduce_sum there) then differ by ~1e-7 * sum_j |f_ij|,
# an order of magnitude above SURVEY 8(c)'s bound, whichever implementation -- TensorFlow's included -- is "right".
# Those call sites name this reason and add the condition term; the same kernels are held to the bound AS STATED on
# the benchmark's own input in test_liquid_configuration_bounds (energies as stated; forces against an independent
# fp32 evaluation's own error).
CONTACTS = "synthetic rows with contacts at r = 0.55-0.9: pair forces 100x the row sum, fp32 summation order"


def _pair_forces_lj(nl64):
    s, t, rp, cond = O._rinv_and_grad_factor(nl64)
    inv_r6 = s ** 6
    dEds = 2.0 * (2.0 * inv_r6 - 1.0) * (6.0 * s ** 5)
    return 2.0 * O._grad_from_dEds(dEds, s, t, rp, cond)


# --------------------------------------------------------------------------- pair vectors
def _system(n, a, jitter, seed, r_list, hdt, three_d=False, ntypes=3):
    if three_d:
        pos, L = fcc_lattice(n, a)
    else:
        pos, L = sq_lattice(n, a)
    rng = np.random.default_rng(seed)
    d = 3 if three_d else 2
    pos[:, :d] += jitter * rng.standard_normal((len(pos), d))
    pos = pos.astype(hdt)
    types = rng.integers(0, ntypes, size=len(pos)).astype(np.int32)
    nn, head, nl = brute_nlist(pos, L, r_list, shuffle_seed=seed)
    return pos, types, L, nn, head, nl


def _to_dev(htf, pos, types, nn, head, nl, hdt, cuda):
    tdt = torch.float64 if hdt == np.float64 else torch.float32
    p4 = htf.ops.stuff_types(torch.from_numpy(pos).to(cuda), torch.from_numpy(types).to(cuda), tdt)
    return (p4, torch.from_numpy(nn.astype(np.int32)).to(cuda), torch.from_numpy(head.astype(np.int32)).to(cuda),
            torch.from_numpy(nl.astype(np.int32)).to(cuda))


@pytest.mark.parametrize("hdt", [np.float32, np.float64])
@pytest.mark.parametrize("NN", [4, 8, 32, 128])
def test_pair_vectors_bit_exact(htf, cuda, hdt, NN):
    """prepareNeighbors parity incl. the overflow wrap (NN=4: Q > 2*NN rows exist)."""
    pos, types, L, nn, head, nl = _system(4, 1.6, 0.08, 11, 3.4, hdt, three_d=True)
    box = O.make_box(L, dtype=hdt)
    ref = O.prepare_neighbors(pos, types, nn, head, nl, box, 3.0, NN)
    p4, dnn, dhead, dnl = _to_dev(htf, pos, types, nn, head, nl, hdt, cuda)
    mc = torch.zeros(1, dtype=torch.int32, device=cuda)
    out = htf.ops.build_pair_vectors(p4, dnn, dhead, dnl, box, 3.0, NN, max_count=mc,
                                     out_dtype=torch.float64 if hdt == np.float64 else torch.float32)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    # kept-count side channel == the largest number of neighbors within r_cut
    assert int(mc.item()) == int(np.sum(np.sum(O.prepare_neighbors(
        pos, types, nn, head, nl, box, 3.0, 512)[..., :3] ** 2, axis=2) > 0, axis=1).max())
    # fp32 destination from fp64 positions == the reference's tf.cast of the fp64 buffer
    if hdt == np.float64:
        out32 = htf.ops.build_pair_vectors(p4, dnn, dhead, dnl, box, 3.0, NN, out_dtype=torch.float32)
        np.testing.assert_array_equal(out32.cpu().numpy(), ref.astype(np.float32))


def test_pair_vectors_batches_and_ragged(htf, cuda):
    pos, types, L, nn, head, nl = _system(6, 2.0, 0.2, 5, 3.0, np.float32)
    # make it ragged: particle 7 has no neighbors at all, 9 keeps only one
    nn = nn.copy()
    nn[7] = 0
    nn[9] = 1
    box = O.make_box(L, dtype=np.float32)
    p4, dnn, dhead, dnl = _to_dev(htf, pos, types, nn, head, nl, np.float32, cuda)
    for off, bs in ((0, 36), (0, 4), (4, 4), (32, 4), (35, 1), (10, 0)):
        ref = O.prepare_neighbors(pos, types, nn, head, nl, box, 2.6, 16, offset=off, batch_size=bs)
        out = htf.ops.build_pair_vectors(p4, dnn, dhead, dnl, box, 2.6, 16, offset=off, batch_size=bs)
        np.testing.assert_array_equal(out.cpu().numpy(), ref)


def test_pair_vectors_fixed_pitch_and_cutoff_edge(htf, cuda):
    """HOOMD's fixed-stride head list, and r == r_cut is KEPT (TensorflowCompute.cc:359)."""
    pos, L = sq_lattice(4, 2.0, dtype=np.float32)  # exact lattice: neighbors at exactly r = 2.0
    types = np.zeros(16, np.int32)
    nn, head, nl = brute_nlist(pos, L, 2.5, pitch=12)
    box = O.make_box(L, dtype=np.float32)
    ref = O.prepare_neighbors(pos, types, nn, head, nl, box, 2.0, 8)
    assert np.all(np.sum(np.sum(ref[..., :3] ** 2, axis=2) > 0, axis=1) == 4)
    p4, dnn, dhead, dnl = _to_dev(htf, pos, types, nn, head, nl, np.float32, cuda)
    out = htf.ops.build_pair_vectors(p4, dnn, dhead, dnl, box, 2.0, 8)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)


def test_reference_force_overwrite_on_gpu(htf, cuda):
    """test_tensorflow.py:81-129 through the whole device path (ctx driver, batch 4)."""
    from test_reference_kats import reference_test_compute_forces
    pos, types, L, nn, head, nl = _system(3, 4.0, 0.15, 2, 5.4, np.float64, ntypes=1)
    box = O.make_box(L)
    p4, dnn, dhead, dnl = _to_dev(htf, pos, types, nn, head, nl, np.float64, cuda)
    for bs in (0, 4):
        ctx = htf.Context(r_cut=5.0, nneighs=8, batch_size=bs, scalar_dtype=torch.float64, max_n=9)
        ctx.set_potential(htf.Potential.simple())
        force = torch.full((9, 4), 7.0, dtype=torch.float64, device=cuda)
        ctx.compute_forces(0, ctx.make_arrays(p4, 9, dnn, dhead, dnl, box, force))
        ref = reference_test_compute_forces(pos, L, 5.0)
        np.testing.assert_allclose(force.cpu().numpy()[:, :3], ref, atol=1e-5)
        assert np.all(force.cpu().numpy()[:, 3] == 0)


@pytest.mark.parametrize("rows", ["1", "2", "8"])
def test_rows_per_wave_variants_are_bit_identical(htf, cuda, rows):
    """The rows-per-wave launch geometries kept for A/B runs (HTF_BUILD_ROWS / HTF_FUSED_ROWS; defaults 4
    and 2) must produce the same tensors and forces: the bit-exact pair-vector tests and the
    fused-vs-two-kernel test are re-run in a child process under each setting."""
    import subprocess
    import sys
    from helpers import variants_env
    env = variants_env(HTF_BUILD_ROWS=rows, HTF_FUSED_ROWS=rows if rows != "8" else "4")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "pair_vectors_bit_exact or ragged_row_lengths or fused_matches_two_kernel"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("fused", [0, 2])
def test_profiler_scopes(htf, cuda, fused):
    """htf_profile_*: the build / evaluator scopes of TensorflowCompute.cc:164-168,196-206 as hipEvent
    brackets.  Every batch, or every k-th one; the one-kernel step is a single scope (eval)."""
    pos, types, L, nn, head, nl = _system(3, 4.0, 0.15, 2, 5.4, np.float64, ntypes=1)
    box = O.make_box(L)
    p4, dnn, dhead, dnl = _to_dev(htf, pos, types, nn, head, nl, np.float64, cuda)
    ctx = htf.Context(r_cut=5.0, nneighs=8, scalar_dtype=torch.float64, max_n=9, fused=fused)
    ctx.set_potential(htf.Potential.lj())
    force = torch.zeros((9, 4), dtype=torch.float64, device=cuda)
    arrays = ctx.make_arrays(p4, 9, dnn, dhead, dnl, box, force)
    ctx.compute_forces(0, arrays)
    want = force.clone()
    assert ctx.profile_read() == (0.0, 0.0, 0)  # off by default
    ctx.profile_enable(True)
    for ts in range(6):
        ctx.compute_forces(ts, arrays)
    b, e, n = ctx.profile_read()
    assert n == 6 and e > 0 and (b > 0) == (fused == 0)
    assert ctx.profile_read()[2] == 0  # read resets
    ctx.profile_enable(3)
    for ts in range(7):
        ctx.compute_forces(ts, arrays)
    b, e, n = ctx.profile_read()
    assert n == 3 and e > 0  # batches 0, 3, 6
    ctx.profile_enable(False)
    ctx.compute_forces(0, arrays)
    assert ctx.profile_read()[2] == 0
    assert torch.equal(force, want)


@pytest.mark.parametrize("hdt", [np.float32, np.float64])
@pytest.mark.parametrize("fused", [False, True])
def test_pair_vectors_ragged_row_lengths(htf, cuda, hdt, fused):
    """Rows of 0, 1, 63..65, 191..193 and 300 list entries side by side, batch sizes that are not a
    multiple of the rows a wave takes: the multi-row fast path (<= 192 entries, no overflow), its
    single-row fallback and the overflow wrap must all give the oracle's tensor bit for bit --
    through the build kernel and through the one-kernel build+evaluate."""
    rng = np.random.default_rng(21)
    lens = [5, 0, 64, 1, 192, 193, 63, 300, 65, 191, 2, 0, 130, 7, 128, 129, 256, 3, 77, 190, 1, 64, 12]
    N = len(lens)
    L = np.array([9.0, 10.0, 11.0])
    pos = ((rng.random((N, 3)) - 0.5) * L).astype(hdt)
    types = rng.integers(0, 3, N).astype(np.int32)
    nn = np.array(lens, dtype=np.int64)
    head = np.concatenate([[0], np.cumsum(nn)[:-1]]).astype(np.int64)
    nl = np.concatenate([rng.integers(0, N, n) for n in lens]).astype(np.int64)
    box = O.make_box(L, dtype=hdt)
    p4, dnn, dhead, dnl = _to_dev(htf, pos, types, nn, head, nl, hdt, cuda)
    for NN, r_cut in ((128, 4.0), (256, 100.0), (16, 100.0)):
        for offset, bs in ((0, N), (1, N - 1), (3, 13), (0, 5), (6, 2)):
            ref = O.prepare_neighbors(pos, types, nn, head, nl, box, r_cut, NN, offset=offset, batch_size=bs)
            if fused:
                pv = torch.full((bs, NN, 4), 7.0, dtype=torch.float32, device=cuda)
                htf.ops.fused_forces(htf.Potential.lj(), p4, dnn, dhead, dnl, box, r_cut, NN, offset=offset, batch_size=bs,
                                     pair_vectors=pv)
                np.testing.assert_array_equal(pv.cpu().numpy(), ref.astype(np.float32))
                # the C4 sweep (row-pipelined persistent kernel): same tensor, histogram and forces as
                # build -> eval_forces2 on these rows
                # (base potentials that stay finite on these random rows, self pairs included: r^-1 takes the compacting
                #  persistent kernel, WCA -- for fp32 positions -- the rows-per-wave form with merged tails, whose generic
                #  single-row routine and overflow un-binning these rows exercise)
                for lj in (htf.Potential.rinv_poly([1.0], [1]), htf.Potential.wca(1.0)):
                    gauss = htf.Potential.gauss(1.1, 0.05, 1.0)
                    pv2 = torch.full((bs, NN, 4), 7.0, dtype=torch.float32, device=cuda)
                    h1 = torch.zeros(102, dtype=torch.int32, device=cuda)
                    fa1, fb1 = htf.ops.build_eval_forces2(lj, gauss, p4, dnn, dhead, dnl, box, r_cut, NN, offset=offset,
                                                          batch_size=bs, rdf=(0.0, 3.5, h1), pair_vectors=pv2)
                    np.testing.assert_array_equal(pv2.cpu().numpy(), ref.astype(np.float32))
                    h0 = torch.zeros(102, dtype=torch.int32, device=cuda)
                    fa0, fb0 = htf.ops.eval_forces2(lj, gauss, pv2, rdf=(0.0, 3.5, h0), out_dtype=p4.dtype)
                    assert torch.equal(h1, h0)
                    for a, b in ((fa1, fa0), (fb1, fb0)):
                        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))
            else:
                mc = torch.zeros(1, dtype=torch.int32, device=cuda)
                out = htf.ops.build_pair_vectors(p4, dnn, dhead, dnl, box, r_cut, NN, offset=offset, batch_size=bs, max_count=mc,
                                                 out_dtype=torch.float64 if hdt == np.float64 else torch.float32)
                np.testing.assert_array_equal(out.cpu().numpy(), ref)
                if hdt == np.float64:
                    out32 = htf.ops.build_pair_vectors(p4, dnn, dhead, dnl, box, r_cut, NN, offset=offset, batch_size=bs,
                                                       out_dtype=torch.float32)
                    np.testing.assert_array_equal(out32.cpu().numpy(), ref.astype(np.float32))
                # kept-count side channel: the largest number of list entries within r_cut among the batch rows
                want = 0
                for i in range(offset, offset + bs):
                    d = pos[nl[head[i]:head[i] + nn[i]]].astype(np.float64) - pos[i].astype(np.float64)
                    d -= np.round(d / L) * L
                    want = max(want, int(((d * d).sum(axis=1) <= r_cut * r_cut).sum()))  # self pairs and repeats count
                assert int(mc.item()) == want


# --------------------------------------------------------------------------- evaluators
def _nlist_case(seed, N=512, NN=128, dtype=np.float32, rmin=0.9):
    rng = np.random.default_rng(seed)
    nl, cnt = random_nlist(rng, N, NN, fill=0.74, rmin=rmin, rmax=3.0, ntypes=2, dtype=dtype)
    nl[0] = 0  # fully padded row
    return nl


@pytest.mark.parametrize("NN", [8, 32, 64, 128, 160])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_lj_forces_energy_virial(htf, cuda, NN, dtype):
    nl = _nlist_case(NN, N=300, NN=NN, dtype=dtype)
    nl64 = nl.astype(np.float32).astype(np.float64)  # the model casts the wire dtype to fp32
    ref_f, ref_v = O.lj_model(nl64, virial=True)
    cond = _cond_scale(nl64, _pair_forces_lj(nl64))
    pot = htf.Potential.lj()
    f, v = htf.ops.eval_forces(pot, torch.from_numpy(nl).to(cuda), virial=True)
    assert f.dtype == (torch.float64 if dtype == np.float64 else torch.float32)
    assert_forces_close("lj_f_NN%d_%s" % (NN, dtype.__name__), f.cpu().numpy(), ref_f, cond, cancelling_rows=CONTACTS)
    assert_forces_close("lj_v_NN%d_%s" % (NN, dtype.__name__), v.cpu().numpy(), ref_v, cond * 3.0, cancelling_rows=CONTACTS)
    f2 = htf.ops.eval_forces(pot, torch.from_numpy(nl).to(cuda))
    np.testing.assert_array_equal(f2.cpu().numpy(), f.cpu().numpy())  # virial flag must not change forces
    # against the fp32 restatement too (what TF itself would produce)
    ref32 = O.lj_model(nl.astype(np.float32))
    assert_forces_close("lj_f32_NN%d_%s" % (NN, dtype.__name__), f.cpu().numpy(), ref32, cond, ctol=4e-6, cancelling_rows=CONTACTS)


@pytest.mark.parametrize("sigma", [0.5, 1.0])
def test_wca_forces(htf, cuda, sigma):
    nl = _nlist_case(3, N=400, NN=128, rmin=0.55 * sigma)
    # clip and cut edges (layers.py:96-98)
    r10 = sigma * 10 ** (-1 / 6)
    nl[1, 0, :3] = [r10 * 1.002, 0, 0]
    nl[1, 1, :3] = [0, r10 * 0.998, 0]
    nl[1, 2, :3] = [0, 0, sigma * 2 ** (1 / 3) * 1.001]
    nl[1, 3, :3] = [0, 0, -sigma * 2 ** (1 / 3) * 0.999]
    nl64 = nl.astype(np.float64)
    ref = O.wca_model(nl64, sigma)
    f = htf.ops.eval_forces(htf.Potential.wca(sigma), torch.from_numpy(nl).to(cuda))
    s, t, rp, cond = O._rinv_and_grad_factor(nl64)
    pf = 2 * O._grad_from_dEds(6 * sigma ** 6 * s ** 5, s, t, rp, cond)
    assert_forces_close("wca_%g" % sigma, f.cpu().numpy(), ref, _cond_scale(nl64, pf), cancelling_rows=CONTACTS)


def test_rinv_poly_and_benchmark_potential(htf, cuda):
    nl = _nlist_case(4, N=200, NN=64)
    nl64 = nl.astype(np.float64)
    f = htf.ops.eval_forces(htf.Potential.rinv_poly([1.0], [1]), torch.from_numpy(nl).to(cuda))
    assert_forces_close("benchmark_potential", f.cpu().numpy(), O.benchmark_potential(nl64))
    # LJ written as a polynomial == the dedicated LJ kernel within tolerance
    f = htf.ops.eval_forces(htf.Potential.rinv_poly([2.0, -2.0], [12, 6]), torch.from_numpy(nl).to(cuda))
    assert_forces_close("poly_lj", f.cpu().numpy(), O.lj_model(nl64), _cond_scale(nl64, _pair_forces_lj(nl64)), cancelling_rows=CONTACTS)
    # example 01's r^-12: e = rinv^12
    f, v = htf.ops.eval_forces(htf.Potential.rinv_poly([1.0], [12]), torch.from_numpy(nl).to(cuda), virial=True)
    rf, rv = O.rinv_poly_model(nl64, [1.0], [12], virial=True)
    s, t, rp, cond = O._rinv_and_grad_factor(nl64)
    c = _cond_scale(nl64, 2 * O._grad_from_dEds(12 * s ** 11, s, t, rp, cond))
    assert_forces_close("poly_r12_f", f.cpu().numpy(), rf, c, cancelling_rows=CONTACTS)
    assert_forces_close("poly_r12_v", v.cpu().numpy(), rv, 3 * c, cancelling_rows=CONTACTS)


def test_rinv_poly_with_hard_cut_example01(htf, cuda):
    """examples/01. Quickstart.ipynb cell 3: pair_energy = tf.cast(tf.norm(nlist[:, :, :3], axis=2) < 2**(1/6), tf.float32)
    * nlist_rinv(nlist)**12 -- WCA as a truncated r^-12; the mask carries no gradient.  Evaluator and fused forms against
    the oracle, and the mask bit at the cut itself: slots at the largest fp32 below 2^(1/6) contribute, at the cut do not."""
    cut = 2 ** (1 / 6)
    nl = _nlist_case(14, N=300, NN=64, rmin=0.85)
    # a row of slots straddling the cut within an ulp (x axis only: the norm is then |x| exactly)
    c32 = np.float32(cut)
    below, above = np.nextafter(c32, np.float32(0)), np.nextafter(c32, np.float32(2))
    nl[1] = 0
    nl[1, :4, 0] = [below, c32, above, -below]
    nl64 = nl.astype(np.float64)
    pot = htf.Potential.rinv_poly([1.0], [12], cut=cut)
    f, v = htf.ops.eval_forces(pot, torch.from_numpy(nl).to(cuda), virial=True)
    rf, rv = O.rinv_poly_model(nl64, [1.0], [12], virial=True, cut=cut)
    s, t, rp, cond = O._rinv_and_grad_factor(nl64)
    r32 = np.sqrt((nl[:, :, :3] ** 2).sum(axis=2, dtype=np.float32))
    g = 2 * O._grad_from_dEds(np.where(r32 < c32, 12 * s ** 11, 0.0), s, t, rp, cond)
    c = _cond_scale(nl64, g)
    assert_forces_close("poly_r12_cut_f", f.cpu().numpy(), rf, c, cancelling_rows=CONTACTS)
    assert_forces_close("poly_r12_cut_v", v.cpu().numpy(), rv, 3 * c, cancelling_rows=CONTACTS)
    # the straddling row: exactly the two slots below the cut count (their energies: rinv^12 at r = below)
    e_one = float(O.rinv_poly_model(nl64[1:2, :1], [1.0], [12], cut=cut)[0, 3])
    np.testing.assert_allclose(float(f[1, 3]), 2 * e_one, rtol=1e-5)
    assert e_one > 0.2
    # the unmasked potential differs (every slot inside r_cut contributes), so the mask is really applied
    f0 = htf.ops.eval_forces(htf.Potential.rinv_poly([1.0], [12]), torch.from_numpy(nl).to(cuda))
    assert float((f0[:, 3] - f[:, 3]).abs().max()) > 1e-3


def test_simple_potential(htf, cuda):
    nl = _nlist_case(5, N=100, NN=32)
    f = htf.ops.eval_forces(htf.Potential.simple(), torch.from_numpy(nl).to(cuda))
    ref = O.compute_outputs(O.simple_potential(nl.astype(np.float64)), np.float64)
    assert_forces_close("simple", f.cpu().numpy(), ref)


def test_rinv_mask_edges(htf, cuda):
    """nlist_rinv's where(r' > 3e-6) mask, slot by slot (simmodel.py:627-635)."""
    nl = np.zeros((4, 8, 4), np.float32)
    nl[1, 0, :3] = [1.6e-6, 1.6e-6, 1.6e-6]
    nl[1, 1, :3] = [1.8e-6, 1.8e-6, 1.8e-6]
    nl[2, 0, :3] = [-1e-7, -1e-7, -1e-7]
    nl[3, 0, :3] = [1.0, 0, 0]
    s = htf.ops.nlist_rinv(torch.from_numpy(nl).to(cuda)).cpu().numpy()
    ref = O.nlist_rinv(nl)
    np.testing.assert_allclose(s, ref, rtol=1e-6)
    assert s[0].max() == 0 and s[1, 0] == 0 and s[1, 1] > 1e5 and s[2, 0] == 0
    f = htf.ops.eval_forces(htf.Potential.rinv_poly([1.0], [1]), torch.from_numpy(nl).to(cuda)).cpu().numpy()
    assert np.all(np.isfinite(f))


# --------------------------------------------------------------------------- pair-MLP (MFMA)
@pytest.mark.parametrize("act", ["tanh", "linear"])
@pytest.mark.parametrize("NN", [128, 40])
def test_pair_mlp_fp32_mfma(htf, cuda, act, NN):
    """C3 potential: RBF(0,3,32) -> 64 -> 64 -> 1 on the fp32 matrix cores vs the fp64
    oracle on the same fp32 inputs.  Non-zero biases and asymmetric random weights so a
    transposed / permuted operand cannot pass; NN=40 exercises the ragged last tile."""
    from hoomd_tf_amd.initializers import mlp_params
    params = mlp_params(seed=3, K=32, H1=64, H2=64, bias_scale=0.2)
    nl = _nlist_case(7, N=257, NN=NN, rmin=0.6)
    nl[1] = 0
    nl[1, 5, :3] = [1.0, -0.5, 0.25]      # a lone real slot after padding (tile skip must not drop it)
    nl64 = nl.astype(np.float64)
    ref, g = O.pair_mlp_model(nl64, params, 0.0, 3.0, act, return_grad=True)
    pot = htf.Potential.pair_mlp(params, 0.0, 3.0, activation=act)
    f = htf.ops.eval_forces(pot, torch.from_numpy(nl).to(cuda))
    cond = np.abs(2 * g).sum(axis=(1, 2))
    assert_forces_close("mlp_%s_NN%d" % (act, NN), f.cpu().numpy(), ref, cond, atol=2e-5, rtol=5e-5, ctol=5e-6)
    assert np.all(f.cpu().numpy()[0] == 0)
    f64 = htf.ops.eval_forces(pot, torch.from_numpy(nl64).to(cuda))
    assert f64.dtype == torch.float64
    # the fp64-wire instantiation casts to fp32 on load: same arithmetic up to fma contraction choices
    assert_forces_close("mlp64_%s_NN%d" % (act, NN), f64.cpu().numpy(), ref, cond, atol=2e-5, rtol=5e-5, ctol=5e-6)


@pytest.mark.parametrize("act", ["tanh", "linear"])
@pytest.mark.parametrize("NN", [128, 40])
@pytest.mark.parametrize("precision", ["split", "split16"])
def test_pair_mlp_split_operands(htf, cuda, act, NN, precision):
    """HTF_MLP_SPLIT: fp32 operands split exactly into three bf16 parts, six partial products
    per multiply on the bf16 matrix pipeline; HTF_MLP_SPLIT16: hi + lo in fp16 (2^-22), three partial
    products on the fp16 pipeline.  Held to the SAME tolerances against the fp64 oracle
    as the fp32-MFMA path (same inputs, same weights), and its error must be of the fp32 path's
    size, not bf16's."""
    from hoomd_tf_amd.initializers import mlp_params
    params = mlp_params(seed=3, K=32, H1=64, H2=64, bias_scale=0.2)
    nl = _nlist_case(7, N=257, NN=NN, rmin=0.6)
    nl[1] = 0
    nl[1, 5, :3] = [1.0, -0.5, 0.25]
    ref, g = O.pair_mlp_model(nl.astype(np.float64), params, 0.0, 3.0, act, return_grad=True)
    x = torch.from_numpy(nl).to(cuda)
    pot = htf.Potential.pair_mlp(params, 0.0, 3.0, activation=act, precision=precision)
    fs = htf.ops.eval_forces(pot, x)
    cond = np.abs(2 * g).sum(axis=(1, 2))
    assert_forces_close("mlp_%s_%s_NN%d" % (precision, act, NN), fs.cpu().numpy(), ref, cond, atol=2e-5, rtol=5e-5, ctol=5e-6)
    assert np.all(fs.cpu().numpy()[0] == 0)
    f32 = htf.ops.eval_forces(htf.Potential.pair_mlp(params, 0.0, 3.0, activation=act), x)
    f16 = htf.ops.eval_forces(htf.Potential.pair_mlp(params, 0.0, 3.0, activation=act, precision="bf16"), x)
    c = 1e-3 + cond[:, None]
    es = (np.abs(fs.cpu().numpy() - ref) / c).max()
    e32 = (np.abs(f32.cpu().numpy() - ref) / c).max()
    e16 = (np.abs(f16.cpu().numpy() - ref) / c).max()
    _record("mlp_%s_vs_fp32_%s_NN%d" % (precision, act, NN), split=es, fp32=e32, bf16=e16)
    assert es < 4 * e32 + 1e-7 and es < 1e-3 * e16, (es, e32, e16)
    assert torch.equal(fs, htf.ops.eval_forces(pot, x))  # deterministic
    # fp64 wire: cast on load, as the other precisions
    f64 = htf.ops.eval_forces(pot, torch.from_numpy(nl.astype(np.float64)).to(cuda))
    assert_forces_close("mlp64_%s_%s_NN%d" % (precision, act, NN), f64.cpu().numpy(), ref, cond, atol=2e-5, rtol=5e-5, ctol=5e-6)


@pytest.mark.parametrize("act", ["tanh", "linear"])
def test_pair_mlp_bf16_operands(htf, cuda, act):
    """Reduced-precision variant (bf16 MFMA operands, fp32 accumulation): NOT a parity
    claim -- bf16 carries 8 mantissa bits, so forces agree with the fp64 oracle only to
    ~1e-2 of the row scale.  Checked: the operand permutation is right (a wrong image
    would be O(1) off), determinism, and that the fp32 path is strictly closer."""
    from hoomd_tf_amd.initializers import mlp_params
    params = mlp_params(seed=3, K=32, H1=64, H2=64, bias_scale=0.2)
    nl = _nlist_case(7, N=257, NN=128, rmin=0.6)
    ref, g = O.pair_mlp_model(nl.astype(np.float64), params, 0.0, 3.0, act, return_grad=True)
    x = torch.from_numpy(nl).to(cuda)
    f16 = htf.ops.eval_forces(htf.Potential.pair_mlp(params, 0.0, 3.0, activation=act, precision="bf16"), x)
    f32 = htf.ops.eval_forces(htf.Potential.pair_mlp(params, 0.0, 3.0, activation=act), x)
    cond = np.abs(2 * g).sum(axis=(1, 2))[:, None]
    e16 = np.abs(f16.cpu().numpy() - ref) / (1e-3 + cond)
    e32 = np.abs(f32.cpu().numpy() - ref) / (1e-3 + cond)
    _record("mlp_bf16_" + act, max_abs_err=np.abs(f16.cpu().numpy() - ref).max(), max_ratio=e16.max(), max_ref=np.abs(ref).max())
    assert e16.max() < 2e-2, e16.max()
    assert e32.max() < e16.max()
    assert torch.equal(f16, htf.ops.eval_forces(htf.Potential.pair_mlp(params, 0.0, 3.0, activation=act, precision="bf16"), x))


def test_pair_mlp_padded_widths(htf, cuda):
    """K < 32 / H < 64 are zero-padded into the fixed 32/64 MFMA tiling."""
    from hoomd_tf_amd.initializers import mlp_params
    params = mlp_params(seed=11, K=8, H1=16, H2=24, bias_scale=0.3)
    nl = _nlist_case(9, N=64, NN=32, rmin=0.7)
    ref, g = O.pair_mlp_model(nl.astype(np.float64), params, 0.5, 2.5, "tanh", return_grad=True)
    f = htf.ops.eval_forces(htf.Potential.pair_mlp(params, 0.5, 2.5), torch.from_numpy(nl).to(cuda))
    assert_forces_close("mlp_padded", f.cpu().numpy(), ref, np.abs(2 * g).sum(axis=(1, 2)), atol=2e-5, rtol=5e-5, ctol=5e-6)
    with pytest.raises(ValueError):
        htf.Potential.pair_mlp(mlp_params(seed=1, K=40), 0, 3)


@pytest.mark.parametrize("precision", ["fp32", "split16", "split"])
def test_pair_mlp_full_size_rows(htf, cuda, precision):
    """BASELINE configs[2] at its own size and on its own input: the C3 fcc box (131 072 particles, NN 128), pair vectors from
    the build kernel, through the MFMA evaluator in every full-precision form -- ``split16`` is the one bench.py times
    (pair_mlp_kernel<tanh, float, SPLIT16>: persistent blocks, two per CU) -- 512 sampled rows of forces, energies AND the
    virial against the fp64 oracle (layers.py:46-49 + Keras Dense, simmodel.py:509-555), at the fixture tolerances; then
    synthetic rows with 70-124 live slots of random directions (every tile-skip pattern), and determinism."""
    from hoomd_tf_amd import standin
    from hoomd_tf_amd.initializers import mlp_params
    sysm, nlc, L = _jittered(standin, cuda, "fcc", 32, 3)
    N, NN = sysm.N, 128
    assert N == 131072
    pv = htf.ops.build_pair_vectors(sysm.pos, nlc.n_neigh, nlc.head_list, nlc.nlist, sysm.box, 3.0, NN)
    params = mlp_params(seed=3)
    pot = htf.Potential.pair_mlp(params, 0.0, 3.0, precision=precision)
    f, v = htf.ops.eval_forces(pot, pv, virial=True)
    f0 = htf.ops.eval_forces(pot, pv)
    assert torch.equal(f, f0) and torch.equal(f0, htf.ops.eval_forces(pot, pv))
    rows = np.random.default_rng(8).choice(N, 512, replace=False)
    sub = pv[torch.from_numpy(rows).to(cuda)].cpu().numpy().astype(np.float64)
    ref, gg = O.pair_mlp_model(sub, params, 0.0, 3.0, "tanh", return_grad=True)
    cond = np.abs(2 * gg).sum(axis=(1, 2))
    assert_forces_close("mlp_c3box_%s" % precision, f.cpu().numpy()[rows], ref, cond, atol=2e-5, rtol=5e-5, ctol=5e-6)
    vref = O.compute_virial(sub, 2.0 * gg)
    vcond = (np.linalg.norm(2 * gg, axis=2) * np.linalg.norm(sub[:, :, :3], axis=2) / 2).sum(axis=1)
    assert_forces_close("mlp_c3box_virial_%s" % precision, v.cpu().numpy()[rows], vref, vcond, atol=2e-5, rtol=5e-5, ctol=5e-6)
    # whole-system property: the energy column of every row against the sampled mean (a dropped or doubled tile shows)
    e_all = f[:, 3].double()
    assert torch.isfinite(f).all() and abs(float(e_all.mean()) - float(ref[:, 3].mean())) < 0.05 * abs(float(ref[:, 3].mean())) + 0.05
    del pv, f, v, f0
    g = torch.Generator(device="cuda").manual_seed(5)
    cnt = torch.randint(70, 125, (N, 1), generator=g, device=cuda)
    d = torch.randn(N, NN, 3, generator=g, device=cuda)
    d = d / d.norm(dim=2, keepdim=True)
    r = 0.8 + 2.2 * torch.rand(N, NN, 1, generator=g, device=cuda) ** (1 / 3)
    nl = torch.zeros(N, NN, 4, device=cuda)
    nl[..., :3] = d * r
    nl *= (torch.arange(NN, device=cuda)[None, :, None] < cnt[:, :, None])
    fs = htf.ops.eval_forces(pot, nl)
    assert torch.equal(fs, htf.ops.eval_forces(pot, nl))
    rows = torch.randint(0, N, (96,), generator=g, device=cuda)
    sub = nl[rows].cpu().numpy().astype(np.float64)
    ref, gg = O.pair_mlp_model(sub, params, 0.0, 3.0, "tanh", return_grad=True)
    assert_forces_close("mlp_full_rows_%s" % precision, fs[rows].cpu().numpy(), ref, np.abs(2 * gg).sum(axis=(1, 2)),
                        atol=2e-5, rtol=5e-5, ctol=5e-6)


# --------------------------------------------------------------------------- config C4 pieces
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_gauss_potential_and_eval2(htf, cuda, dtype):
    """HTF_POT_GAUSS alone, and base + Gaussian in one pass with the deterministic CV sum."""
    nl = _nlist_case(13, N=333, NN=128, dtype=dtype, rmin=0.8)
    nl64 = nl.astype(np.float32).astype(np.float64)
    x = torch.from_numpy(nl).to(cuda)
    pg = htf.Potential.gauss(1.1, 0.05, 1.0)
    ref_g = O.gauss_model(nl64, 1.1, 0.05, 1.0)
    _, g = O.gauss_pair_terms(nl64, 1.1, 0.05)
    cond_g = np.abs(2 * g).sum(axis=(1, 2))
    assert_forces_close("gauss_" + dtype.__name__, htf.ops.eval_forces(pg, x).cpu().numpy(), ref_g, cond_g, ctol=4e-6, cancelling_rows=CONTACTS)
    n = htf.ops.num_partials(nl.shape[0], 128)
    partials = torch.zeros(n, device=cuda)
    fa, fb = htf.ops.eval_forces2(htf.Potential.lj(), pg, x, partials=partials)
    assert_forces_close("eval2_lj_" + dtype.__name__, fa.cpu().numpy(), O.lj_model(nl64), _cond_scale(nl64, _pair_forces_lj(nl64)), cancelling_rows=CONTACTS)
    assert_forces_close("eval2_gauss_" + dtype.__name__, fb.cpu().numpy(), ref_g, cond_g, ctol=4e-6, cancelling_rows=CONTACTS)
    cv = torch.zeros(1, device=cuda)
    htf.ops.reduce_partials(partials, n, 1.0 / nl.shape[0], cv)
    np.testing.assert_allclose(float(cv), ref_g[:, 3].sum() / nl.shape[0], rtol=2e-6)
    # EDS-biased assembly == the oracle composite at the same alpha
    alpha = torch.tensor([0.7], device=cuda)
    out = htf.ops.bias_combine(fa.clone(), fb, alpha, cv)
    ref, rcv = O.eds_rdf_model(nl64, 0.7, 1.1, 0.05)
    assert_forces_close("eds_rdf_" + dtype.__name__, out.cpu().numpy(), ref, _cond_scale(nl64, _pair_forces_lj(nl64)) + 0.7 * cond_g, cancelling_rows=CONTACTS)
    with pytest.raises(ValueError):
        htf.ops.eval_forces2(htf.Potential.lj(), htf.Potential.lj(), x)
    # compute_rdf fused into the same sweep == the stand-alone histogram kernel == the oracle
    hist = torch.zeros(22, dtype=torch.int32, device=cuda)
    htf.ops.eval_forces2(htf.Potential.lj(), pg, x, rdf=(0.0, 3.5, hist))
    from hoomd_tf_amd.simmodel import rdf_from_histogram
    rdf_fused, _ = rdf_from_histogram(hist, 0.0, 3.5)
    rdf_plain, _ = htf.compute_rdf(x, [0, 3.5], nbins=20)
    assert torch.equal(rdf_fused, rdf_plain)
    ref_rdf, _ = O.compute_rdf(nl.astype(np.float32), [0, 3.5], nbins=20)
    np.testing.assert_allclose(rdf_fused.cpu().numpy(), ref_rdf, rtol=1e-4)


def test_slot_at_minus_norm_delta_contributes_zero(htf, cuda):
    """x = y = z = -1e-7 exactly: safe_norm's r' = |x + 1e-7| is 0, 1 / r' is inf and the rsq-based forward's r' is 0 * inf =
    NaN.  TensorFlow's sqrt gradient is NaN there (the one excluded input); the kernels return an exact
    zero for that slot -- for every closed form, the Gaussian channel (whose force chain meets the NaN: ADVICE r3) and both
    outputs of the two-potential sweep -- so one such slot cannot poison a row sum or the CV partial."""
    nl = _nlist_case(17, N=130, NN=64, rmin=0.8)
    nl[:, 63] = 0
    want = torch.from_numpy(nl).to(cuda)
    nl[5, 63, :3] = np.float32(-1e-7)
    nl[77, 63, :3] = np.float32(-1e-7)
    x = torch.from_numpy(nl).to(cuda)
    pg = htf.Potential.gauss(1.1, 0.05, 1.0)
    for pot in (pg, htf.Potential.lj(), htf.Potential.wca(1.0), htf.Potential.rinv_poly([1.0, -0.5], [12, 1]), htf.Potential.lj_param(1.3, 0.9)):
        f, v = htf.ops.eval_forces(pot, x, virial=True)
        assert torch.isfinite(f).all() and torch.isfinite(v).all(), pot.kind
        assert torch.equal(f, htf.ops.eval_forces(pot, want)), pot.kind
    n = htf.ops.num_partials(nl.shape[0], 64)
    pa, pb = torch.zeros(n, device=cuda), torch.zeros(n, device=cuda)
    fa, fb = htf.ops.eval_forces2(htf.Potential.lj(), pg, x, partials=pa)
    wa, wb = htf.ops.eval_forces2(htf.Potential.lj(), pg, want, partials=pb)
    assert torch.isfinite(fa).all() and torch.isfinite(fb).all() and torch.isfinite(pa).all()
    assert torch.equal(fa, wa) and torch.equal(fb, wb) and torch.equal(pa, pb)


# --------------------------------------------------------------------------- aux kernels
@pytest.mark.parametrize("tdt", [torch.float32, torch.float64])
def test_aux_kernels(htf, cuda, tdt):
    g = torch.Generator(device="cpu").manual_seed(0)
    N, pitch = 37, 40
    src9 = torch.randn(N, 9, generator=g, dtype=tdt)
    dest = torch.randn(6 * pitch, generator=g, dtype=tdt)
    ref = O.receive_virial(dest.numpy().copy(), src9.numpy(), pitch, 0, N)
    out = htf.ops.add_virial(dest.to(cuda), src9.to(cuda), N, pitch)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    a, b = torch.randn(N, 4, generator=g, dtype=tdt), torch.randn(N, 4, generator=g, dtype=tdt)
    np.testing.assert_array_equal(htf.ops.add_scalar4(a.to(cuda), b.to(cuda)).cpu().numpy(), (a + b).numpy())
    xyz = torch.randn(N, 3, generator=g, dtype=tdt)
    types = torch.arange(N) % 5
    p4 = htf.ops.stuff_types(xyz.to(cuda), types.to(cuda), tdt)
    un = htf.ops.copy_positions(p4, offset=3, N=20)
    np.testing.assert_array_equal(un.cpu().numpy()[:, :3], xyz.numpy()[3:23])
    np.testing.assert_array_equal(un.cpu().numpy()[:, 3], types.numpy()[3:23].astype(np.float64))
    un32 = htf.ops.copy_positions(p4, out_dtype=torch.float32)
    np.testing.assert_array_equal(un32.cpu().numpy()[:, 3], types.numpy().astype(np.float32))


# --------------------------------------------------------------------------- context driver
def test_context_lj_batched_period_virial(htf, cuda):
    """computeForces semantics: batching leaves results identical (test_force_overwrite_batched),
    period gating reuses old forces (TensorflowCompute.cc:133), virial folds in with +=."""
    pos, types, L, nn, head, nl = _system(5, 3.0, 0.08, 1, 5.4, np.float64, ntypes=1)
    N = 25
    box = O.make_box(L)
    p4, dnn, dhead, dnl = _to_dev(htf, pos, types, nn, head, nl, np.float64, cuda)
    ref_f, ref_v = O.compute_forces(pos, types, nn, head, nl, box, 5.0, 32,
                                    lambda x: O.lj_model(x.astype(np.float64), virial=True), virial=True,
                                    model_dtype=np.float32)
    outs = []
    for bs in (0, 4, 25, 7):
        ctx = htf.Context(r_cut=5.0, nneighs=32, batch_size=bs, period=3, scalar_dtype=torch.float64,
                          virial=True, max_n=N)
        ctx.set_potential(htf.Potential.lj())
        force = torch.zeros((N, 4), dtype=torch.float64, device=cuda)
        vir = torch.zeros(6 * N, dtype=torch.float64, device=cuda)
        arrays = ctx.make_arrays(p4, N, dnn, dhead, dnl, box, force, vir, N)
        ctx.compute_forces(0, arrays)
        assert_forces_close("ctx_lj_bs%d" % bs, force.cpu().numpy(), ref_f)
        assert_forces_close("ctx_virial_bs%d" % bs, vir.cpu().numpy(), ref_v, atol=2e-5)
        outs.append(force.cpu().numpy().copy())
        # period gate: timestep 1, 2 leave the arrays untouched
        force.fill_(-1.0)
        ctx.compute_forces(1, arrays)
        ctx.compute_forces(2, arrays)
        assert torch.all(force == -1.0)
        ctx.compute_forces(3, arrays)
        np.testing.assert_array_equal(force.cpu().numpy(), outs[-1])
        np.testing.assert_allclose(vir.cpu().numpy(), 2 * ref_v, atol=4e-5)  # accumulated (+=)
        if bs in (0, 25):
            nlb = ctx.nlist_buffer(N).cpu().numpy()
            ref_nl = O.prepare_neighbors(pos, types, nn, head, nl, box, 5.0, 32).astype(np.float32)
            np.testing.assert_array_equal(nlb, ref_nl)
            np.testing.assert_array_equal(ctx.positions_buffer(N).cpu().numpy()[:, :3], pos.astype(np.float32))
    for o in outs[1:]:
        np.testing.assert_array_equal(o, outs[0])


def test_context_errors(htf, cuda):
    """test_overflow (test_tensorflow.py:830-848) and test_skew_fails (:321-333)."""
    pos, L = sq_lattice(8, 4.0)
    rng = np.random.default_rng(1)
    pos[:, :2] += 0.05 * rng.standard_normal((64, 2))
    types = np.zeros(64, np.int32)
    nn, head, nl = brute_nlist(pos, L, 10.0, shuffle_seed=3)
    p4, dnn, dhead, dnl = _to_dev(htf, pos, types, nn, head, nl, np.float64, cuda)
    force = torch.zeros((64, 4), dtype=torch.float64, device=cuda)
    ctx = htf.Context(r_cut=10.0, nneighs=4, scalar_dtype=torch.float64, check_nlist=True, max_n=64)
    ctx.set_potential(htf.Potential.lj())
    with pytest.raises(htf.NlistOverflowError):
        ctx.compute_forces(0, ctx.make_arrays(p4, 64, dnn, dhead, dnl, O.make_box(L), force))
    ctx = htf.Context(r_cut=10.0, nneighs=64, scalar_dtype=torch.float64, check_nlist=True, max_n=64)
    ctx.set_potential(htf.Potential.lj())
    ctx.compute_forces(0, ctx.make_arrays(p4, 64, dnn, dhead, dnl, O.make_box(L), force))
    with pytest.raises(htf.SkewedBoxError):
        ctx.compute_forces(0, ctx.make_arrays(p4, 64, dnn, dhead, dnl, O.make_box(L, tilt=(0.5, 0, 0)), force))
    assert htf.ops.check_nlist(ctx.nlist_buffer(64)) == O.check_nlist_count(
        O.prepare_neighbors(pos, types, nn, head, nl, O.make_box(L), 10.0, 64))


# --------------------------------------------------------------------------- fused gather-evaluate
@pytest.mark.parametrize("hdt", [np.float32, np.float64])
@pytest.mark.parametrize("NN", [8, 32, 128])
def test_fused_matches_two_kernel_path_and_oracle(htf, cuda, hdt, NN):
    """htf_fused_forces == htf_build_pair_vectors -> htf_eval_forces (same semantics incl.
    the NN wrap on overflow: NN=8 overflows on this system) and the oracle."""
    pos, types, L, nn, head, nl = _system(4, 1.6, 0.08, 11, 3.4, hdt, three_d=True)
    box = O.make_box(L, dtype=hdt)
    p4, dnn, dhead, dnl = _to_dev(htf, pos, types, nn, head, nl, hdt, cuda)
    for pot, model in ((htf.Potential.lj(), lambda t: O.lj_model(t, virial=True)),
                       (htf.Potential.wca(1.0), None), (htf.Potential.rinv_poly([1.0], [1]), None)):
        virial = model is not None
        cc = torch.zeros(1, dtype=torch.int32, device=cuda)
        out = htf.ops.fused_forces(pot, p4, dnn, dhead, dnl, box, 3.0, NN, virial=virial, check_count=cc)
        pv = htf.ops.build_pair_vectors(p4, dnn, dhead, dnl, box, 3.0, NN)
        two = htf.ops.eval_forces(pot, pv, virial=virial, out_dtype=p4.dtype)
        f, f2 = (out[0], two[0]) if virial else (out, two)
        scale = max(1.0, float(f2.abs().max()))
        assert float((f - f2).abs().max()) <= 2e-5 * scale
        assert int(cc.item()) == htf.ops.check_nlist(pv)
        # htf_build_eval_forces: the same kernel also writes the tensor -- bit-identical to the
        # build kernel's (zero padding and the overflow wrap included), forces identical to the
        # tensor-less fused call
        pv2 = torch.full_like(pv, 7.0)
        out2 = htf.ops.fused_forces(pot, p4, dnn, dhead, dnl, box, 3.0, NN, virial=virial, pair_vectors=pv2)
        assert torch.equal(pv2, pv)
        assert torch.equal(out2[0] if virial else out2, f)
        if virial:
            assert float((out[1] - two[1]).abs().max()) <= 2e-5 * max(1.0, float(two[1].abs().max()))
            ref_nl = O.prepare_neighbors(pos, types, nn, head, nl, box, 3.0, NN).astype(np.float32).astype(np.float64)
            rf, rv = model(ref_nl)
            cond = _cond_scale(ref_nl, _pair_forces_lj(ref_nl))
            # (jittered fcc with a = 1.6: contacts at r ~ 0.9, pair forces ~100x the row sums)
            assert_forces_close("fused_lj_NN%d_%s" % (NN, hdt.__name__), f.cpu().numpy(), rf, cond, cancelling_rows=CONTACTS)
            assert_forces_close("fused_ljv_NN%d_%s" % (NN, hdt.__name__), out[1].cpu().numpy(), rv, 3 * cond, cancelling_rows=CONTACTS)
    # batches
    full = htf.ops.fused_forces(htf.Potential.lj(), p4, dnn, dhead, dnl, box, 3.0, NN)
    part = htf.ops.fused_forces(htf.Potential.lj(), p4, dnn, dhead, dnl, box, 3.0, NN, offset=7, batch_size=20)
    # (four rows per wave share one trip for their tails when those fit 64 lanes: a row's partial sums then depend
    #  on its neighbours in the group, i.e. on where the batch starts -- equal to rounding, not bit for bit)
    scale = float(full.abs().max())
    assert float((part - full[7:27]).abs().max()) <= 2e-6 * scale
    with pytest.raises(ValueError):
        from hoomd_tf_amd.initializers import mlp_params
        htf.ops.fused_forces(htf.Potential.pair_mlp(mlp_params(), 0, 3), p4, dnn, dhead, dnl, box, 3.0, NN)


@pytest.mark.parametrize("hdt", [np.float32, np.float64])
@pytest.mark.parametrize("NN", [8, 32, 128])
def test_build_eval_forces2_equals_build_then_eval2(htf, cuda, hdt, NN):
    """Config C4's sweep as ONE kernel (htf_build_eval_forces2) against htf_build_pair_vectors ->
    htf_eval_forces2: tensor and compute_rdf histogram identical (the padded slots and, at NN = 8,
    the overflow wrap included), both force sets and the CV sum to summation-order rounding."""
    pos, types, L, nn, head, nl = _system(4, 1.6, 0.08, 11, 3.4, hdt, three_d=True)
    box = O.make_box(L, dtype=hdt)
    p4, dnn, dhead, dnl = _to_dev(htf, pos, types, nn, head, nl, hdt, cuda)
    N = len(pos)
    lj, gauss = htf.Potential.lj(), htf.Potential.gauss(1.1, 0.05, 1.0)
    pv = htf.ops.build_pair_vectors(p4, dnn, dhead, dnl, box, 3.0, NN)
    hist0 = torch.zeros(102, dtype=torch.int32, device=cuda)
    part0 = torch.zeros(htf.ops.num_partials(N, NN), device=cuda)
    fa0, fb0 = htf.ops.eval_forces2(lj, gauss, pv, partials=part0, rdf=(0.0, 3.5, hist0), out_dtype=p4.dtype)
    hist1 = torch.zeros(102, dtype=torch.int32, device=cuda)
    part1 = torch.zeros(htf.ops.num_partials_fused(N), device=cuda)
    pv1 = torch.full_like(pv, 3.0)
    fa1, fb1 = htf.ops.build_eval_forces2(lj, gauss, p4, dnn, dhead, dnl, box, 3.0, NN, partials=part1,
                                          rdf=(0.0, 3.5, hist1), pair_vectors=pv1)
    assert torch.equal(pv1, pv)
    assert torch.equal(hist1, hist0) and int(hist0.sum()) == N * NN
    for a, b in ((fa1, fa0), (fb1, fb0)):
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))
    assert abs(float(part1.sum()) - float(part0.sum())) <= 2e-5 * abs(float(part0.sum()))
    # tensor-less form, and a batch
    fa2, fb2 = htf.ops.build_eval_forces2(lj, gauss, p4, dnn, dhead, dnl, box, 3.0, NN)
    assert torch.equal(fa2, fa1) and torch.equal(fb2, fb1)
    fa3, _ = htf.ops.build_eval_forces2(lj, gauss, p4, dnn, dhead, dnl, box, 3.0, NN, offset=5, batch_size=30)
    # (the rows-per-wave form merges the tails of a wave's rows into one trip: a row's partial sums then depend on which rows
    #  share its wave, i.e. on where the batch starts -- summation-order rounding, as for the LJ step)
    assert float((fa3 - fa1[5:35]).abs().max()) <= 2e-6 * max(1.0, float(fa1.abs().max()))
    with pytest.raises(ValueError):
        htf.ops.build_eval_forces2(lj, lj, p4, dnn, dhead, dnl, box, 3.0, NN)


def test_context_fused_mode(htf, cuda):
    """htf_config.fused: same forces / virial / errors as the reference dataflow."""
    pos, types, L, nn, head, nl = _system(5, 3.0, 0.08, 1, 5.4, np.float64, ntypes=1)
    N = 25
    box = O.make_box(L)
    p4, dnn, dhead, dnl = _to_dev(htf, pos, types, nn, head, nl, np.float64, cuda)
    res = []
    for fused in (0, 1, 2):
        ctx = htf.Context(r_cut=5.0, nneighs=32, batch_size=7, scalar_dtype=torch.float64, virial=True,
                          check_nlist=True, max_n=N, fused=fused)
        ctx.set_potential(htf.Potential.lj())
        force = torch.zeros((N, 4), dtype=torch.float64, device=cuda)
        vir = torch.zeros(6 * N, dtype=torch.float64, device=cuda)
        ctx.compute_forces(0, ctx.make_arrays(p4, N, dnn, dhead, dnl, box, force, vir, N))
        res.append((force.cpu().numpy(), vir.cpu().numpy(), ctx.positions_buffer(4).cpu().numpy(),
                    ctx.nlist_buffer(4, cuda).cpu().numpy()))
    for k in (1, 2):
        np.testing.assert_allclose(res[0][0], res[k][0], rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(res[0][1], res[k][1], rtol=2e-5, atol=1e-7)
        np.testing.assert_array_equal(res[0][2], res[k][2])
    # mode 2 leaves the last batch's pair vectors in the side buffer exactly as the build kernel does
    np.testing.assert_array_equal(res[0][3], res[2][3])
    assert np.abs(res[0][3]).max() > 0
    # test_overflow's system (8x8, r_cut 10, NN 4): the dx > 0 count certainly reaches NN
    pos, L = sq_lattice(8, 4.0)
    pos[:, :2] += 0.05 * np.random.default_rng(1).standard_normal((64, 2))
    nn, head, nl = brute_nlist(pos, L, 10.0, shuffle_seed=3)
    p4, dnn, dhead, dnl = _to_dev(htf, pos, np.zeros(64, np.int32), nn, head, nl, np.float64, cuda)
    ctx = htf.Context(r_cut=10.0, nneighs=4, scalar_dtype=torch.float64, check_nlist=True, max_n=64, fused=True)
    ctx.set_potential(htf.Potential.lj())
    with pytest.raises(htf.NlistOverflowError):
        ctx.compute_forces(0, ctx.make_arrays(p4, 64, dnn, dhead, dnl, O.make_box(L),
                                              torch.zeros((64, 4), dtype=torch.float64, device=cuda)))


# --------------------------------------------------------------------------- full size
def test_full_size_lj_rows_and_properties(htf, cuda):
    """BASELINE size (131072 x 128): rows are independent, so a random sample of rows
    is checked against the oracle exactly as in the small tests; plus properties
    that do not depend on the oracle: slot-permutation invariance and determinism."""
    N, NN = 131072, 128
    g = torch.Generator(device="cuda").manual_seed(3)
    cnt = torch.randint(70, 125, (N, 1), generator=g, device=cuda)
    v = torch.randn(N, NN, 3, generator=g, device=cuda)
    v = v / v.norm(dim=2, keepdim=True)
    r = 0.9 + 2.1 * torch.rand(N, NN, 1, generator=g, device=cuda) ** (1 / 3)
    nl = torch.zeros(N, NN, 4, device=cuda)
    nl[..., :3] = v * r
    nl *= (torch.arange(NN, device=cuda)[None, :, None] < cnt[:, :, None])
    pot = htf.Potential.lj()
    f = htf.ops.eval_forces(pot, nl)
    f_again = htf.ops.eval_forces(pot, nl)
    assert torch.equal(f, f_again)
    rows = torch.randint(0, N, (384,), generator=g, device=cuda)
    sub = nl[rows].cpu().numpy().astype(np.float64)
    ref = O.lj_model(sub)
    assert_forces_close("full_lj_rows", f[rows].cpu().numpy(), ref, _cond_scale(sub, _pair_forces_lj(sub)), cancelling_rows=CONTACTS)
    # permuting the real slots of each row only reorders the sum
    perm = torch.argsort(torch.rand(N, NN, generator=g, device=cuda), dim=1)
    nl_p = torch.gather(nl, 1, perm[:, :, None].expand(-1, -1, 4)).contiguous()
    f_p = htf.ops.eval_forces(pot, nl_p)
    scale = f.abs().max().item()
    assert (f_p - f).abs().max().item() <= 2e-5 * scale
    # total energy: checksum of checksums against a float64 torch reduction of pair terms
    tx = (nl[..., :3].double() + 1e-7)
    rp = tx.norm(dim=2)
    s = torch.where(rp > 3e-6, 1.0 / (rp + 3e-6), torch.zeros_like(rp))
    e_ref = (2.0 * (s ** 12 - s ** 6)).sum().item()
    assert abs(f[:, 3].double().sum().item() - e_ref) <= 2e-6 * abs(e_ref) + 1e-3


@pytest.mark.parametrize("sigma", [0.5, 1.0, 0.8371, 1.7])
def test_wca_mask_at_the_cutoff_is_the_sqrt_mask(htf, cuda, sigma):
    """The WCA mask is evaluated as r2 < t (no square root in the kernel) with t chosen so that it equals
    float32 sqrt(r2) < cut for EVERY r2: sweep +-200 ulps around the cutoff."""
    cut = np.float32(np.float32(sigma) * np.float32(1.2599210498948732))
    xs = [cut]
    for _ in range(200):
        xs.append(np.nextafter(xs[-1], np.float32(10), dtype=np.float32))
    lo = cut
    for _ in range(200):
        lo = np.nextafter(lo, np.float32(0), dtype=np.float32)
        xs.append(lo)
    x = np.array(sorted(xs), dtype=np.float32)
    nl = np.zeros((len(x), 1, 4), dtype=np.float32)
    nl[:, 0, 0] = x
    f = htf.ops.eval_forces(htf.Potential.wca(sigma), torch.from_numpy(nl).to(cuda)).cpu().numpy()
    inside_gpu = f[:, 3] != 0
    inside_ref = np.sqrt((x * x).astype(np.float32)).astype(np.float32) < cut
    np.testing.assert_array_equal(inside_gpu, inside_ref)
    assert inside_ref.any() and not inside_ref.all()


# --------------------------------------------------------------------------- round-2 additions
@pytest.mark.parametrize("precision", ["fp32", "split", "split16"])
@pytest.mark.parametrize("act", ["tanh", "linear"])
def test_pair_mlp_virial(htf, cuda, act, precision):
    """compute_nlist_forces(nlist, energy, virial=True) works for ANY energy upstream (simmodel.py:552-554,
    509-523): the pair-MLP evaluator forms -(|nf| / (2 |x|)) x (x) x per slot where it forms du/dr."""
    from hoomd_tf_amd.initializers import mlp_params
    params = mlp_params(seed=5, K=32, H1=64, H2=64, bias_scale=0.2)
    nl = _nlist_case(11, N=130, NN=72, rmin=0.6)
    nl[3] = 0
    nl64 = nl.astype(np.float64)
    ref, g = O.pair_mlp_model(nl64, params, 0.0, 3.0, act, return_grad=True)
    vref = O.compute_virial(nl64, 2.0 * g)
    pot = htf.Potential.pair_mlp(params, 0.0, 3.0, activation=act, precision=precision)
    f, v = htf.ops.eval_forces(pot, torch.from_numpy(nl).to(cuda), virial=True)
    f0 = htf.ops.eval_forces(pot, torch.from_numpy(nl).to(cuda))
    assert torch.equal(f, f0)  # the virial instantiation leaves the forces bit for bit alone
    cond = np.abs(2 * g).sum(axis=(1, 2))
    assert_forces_close("mlp_virial_f_%s_%s" % (act, precision), f.cpu().numpy(), ref, cond, atol=2e-5, rtol=5e-5, ctol=5e-6)
    vcond = (np.linalg.norm(2 * g, axis=2) * np.linalg.norm(nl64[:, :, :3], axis=2) / 2).sum(axis=1)
    assert_forces_close("mlp_virial_v_%s_%s" % (act, precision), v.cpu().numpy(), vref, vcond, atol=2e-5, rtol=5e-5, ctol=5e-6)
    assert np.all(v.cpu().numpy()[3] == 0)
    v_np = v.cpu().numpy()
    np.testing.assert_array_equal(v_np, np.transpose(v_np, (0, 2, 1)))  # symmetric by construction
    # and through the context (SimModel(virial=True) + htf_add_virial's 6-component pick)
    f64, v64 = htf.ops.eval_forces(pot, torch.from_numpy(nl64).to(cuda), virial=True)
    assert v64.dtype == torch.float64
    assert_forces_close("mlp_virial_v64_%s_%s" % (act, precision), v64.cpu().numpy(), vref, vcond, atol=2e-5, rtol=5e-5, ctol=5e-6)


@pytest.mark.parametrize("tdt", [torch.float32, torch.float64])
def test_positions_forces_radial(htf, cuda, tdt):
    """a15 on the device: compute_positions_forces of BenchmarkNonlistModel's energy
    (build_examples.py:59-64: divide_no_nan(1., tf.norm(positions, axis=1)), the type column included)."""
    rng = np.random.default_rng(4)
    p = rng.uniform(-30, 30, size=(1000, 4))
    p[:, 3] = rng.integers(0, 3, size=1000)
    p[7] = 0.0
    p[8] = [3.0, 0.0, 4.0, 0.0]
    p[9] = [1.0, 2.0, 2.0, 4.0]
    pt = torch.tensor(p, dtype=tdt, device=cuda)
    got = htf.ops.positions_forces_radial(pt).cpu().numpy()
    ref = O.positions_radial_model(p.astype(np.float32 if tdt == torch.float32 else np.float64).astype(np.float64))
    assert_forces_close("positions_radial_%s" % str(tdt)[-7:], got, ref)
    np.testing.assert_allclose(got[8], [3 / 125, 0.0, 4 / 125, 0.2], rtol=1e-6)   # by hand
    np.testing.assert_allclose(got[9], [1 / 125, 2 / 125, 2 / 125, 0.2], rtol=1e-6)
    assert np.all(got[7] == 0.0)
    # other powers / three columns against the same closed form
    for power, ncomp, coef in ((-2, 3, 0.5), (2, 4, 0.25), (1, 3, -1.5)):
        got = htf.ops.positions_forces_radial(pt, coef=coef, power=power, ncomp=ncomp).cpu().numpy()
        assert_forces_close("positions_radial_p%d_c%d" % (power, ncomp), got,
                            O.positions_radial_model(p, coef=coef, power=power, ncomp=ncomp), atol=2e-5, rtol=2e-5)
    with pytest.raises(ValueError):
        htf.ops.positions_forces_radial(pt, power=0)


def test_scalar_total_energy_is_tiled(htf, cuda):
    """_add_energy (simmodel.py:558-578): a rank-0 energy lands in EVERY particle's energy column; the
    forces are those of the per-particle form (tf.gradients sums the energy either way)."""
    nl = _nlist_case(3, N=50, NN=16, rmin=0.9)
    x = htf.Nlist(torch.from_numpy(nl).to(cuda))
    rinv = htf.nlist_rinv(x)
    e_pair = 2.0 * (rinv ** 12 - rinv ** 6)
    f_row = htf.compute_nlist_forces(x, htf.reduce_sum(e_pair, axis=1)).cpu().numpy()
    f_tot = htf.compute_nlist_forces(x, htf.reduce_sum(e_pair)).cpu().numpy()
    np.testing.assert_array_equal(f_tot[:, :3], f_row[:, :3])
    np.testing.assert_allclose(f_tot[:, 3], np.full(50, f_row[:, 3].astype(np.float64).sum()), rtol=1e-5)
    ref = O.lj_model(nl.astype(np.float64))
    assert_forces_close("lj_total_energy", f_tot[:, :3], ref[:, :3])


@pytest.mark.parametrize("act", [None, "tanh"])
@pytest.mark.parametrize("NN,K,H", [(64, 8, 16), (128, 8, 32), (24, 6, 8), (200, 16, 64)])
def test_topk_mlp_example08(htf, cuda, act, NN, K, H):
    """Example 08's model as ONE kernel (htf_potential_kind TOPK_MLP): wave arg-max top-k of 1/r, the per-particle
    Dense stack forward and backward, the gradient routed back to the slots the sorted values came from."""
    rng = np.random.default_rng(NN + K)
    nl, _ = random_nlist(rng, 203, NN, fill=0.6, rmin=0.8, rmax=3.0, dtype=np.float32)
    nl[0] = 0
    nl[1, 3:] = 0          # fewer real neighbors than K
    nl[2, 1, :3] = nl[2, 0, :3][::-1]   # two slots at exactly the same distance (a lattice tie)
    params = {"W1": rng.normal(0, 0.4, (K, H)), "b1": rng.normal(0, 0.1, H), "W2": rng.normal(0, 0.3, (H, H)),
              "b2": rng.normal(0, 0.1, H), "W3": rng.normal(0, 0.3, (H, 1)), "b3": rng.normal(0, 0.1, 1)}
    params = {k: v.astype(np.float32) for k, v in params.items()}
    nl64 = nl.astype(np.float64)
    ref, g = O.topk_mlp_model(nl64, {k: v.astype(np.float64) for k, v in params.items()}, act=act, return_grad=True)
    pot = htf.Potential.topk_mlp(params, activation=act)
    x = torch.from_numpy(nl).to(cuda)
    f, v = htf.ops.eval_forces(pot, x, virial=True)
    cond = np.abs(2 * g).sum(axis=(1, 2))
    assert_forces_close("topk_mlp_%s_NN%d" % (act, NN), f.cpu().numpy(), ref, cond, atol=2e-5, rtol=5e-5)
    assert_forces_close("topk_mlp_v_%s_NN%d" % (act, NN), v.cpu().numpy(), O.compute_virial(nl64, 2.0 * g), atol=2e-5, rtol=5e-5)
    # (the virial instantiation is compiled separately: fma contraction may differ in the last bit)
    np.testing.assert_allclose(htf.ops.eval_forces(pot, x).cpu().numpy(), f.cpu().numpy(), rtol=2e-5, atol=1e-5)
    f64 = htf.ops.eval_forces(pot, torch.from_numpy(nl64).to(cuda))
    assert f64.dtype == torch.float64
    assert_forces_close("topk_mlp64_%s_NN%d" % (act, NN), f64.cpu().numpy(), ref, cond, atol=2e-5, rtol=5e-5)
    # the sorted features themselves: tf.math.top_k semantics (values descending, ties by index)
    s = htf.ops.nlist_rinv(x)
    vals, idx = htf.ops.topk_desc(s, K)
    s_np = s.cpu().numpy()
    order = np.argsort(-s_np, axis=1, kind="stable")[:, :K]
    np.testing.assert_array_equal(idx.cpu().numpy(), order)
    np.testing.assert_array_equal(vals.cpu().numpy(), np.take_along_axis(s_np, order, axis=1))


def test_topk_mlp_through_the_model_surface(htf, cuda):
    """build_examples.NlistNN written as the reference writes it (sort -> Dense -> Dense -> Dense ->
    compute_nlist_forces) lowers to the fused kernel and equals the oracle on the same pair vectors."""
    import build_examples
    rng = np.random.default_rng(5)
    nl, _ = random_nlist(rng, 64, 32, fill=0.7, rmin=0.8, rmax=3.0, dtype=np.float32)
    model = build_examples.NlistNN(32, dim=16, top_neighs=8)
    x = htf.Nlist(torch.from_numpy(nl).to(cuda))
    (f,) = model([x, None, None])
    d1, d2, d3 = model.dense1, model.dense2, model.last
    assert d1.kernel.shape == (8, 16) and d3.kernel.shape == (16, 1) and not d1.bias.any()   # built on first call
    params = {"W1": d1.kernel, "b1": d1.bias, "W2": d2.kernel, "b2": d2.bias, "W3": d3.kernel, "b3": d3.bias}
    ref = O.topk_mlp_model(nl.astype(np.float64), {k: v.astype(np.float64) for k, v in params.items()})
    assert_forces_close("nlistnn_model", f.cpu().numpy(), ref, atol=2e-5, rtol=5e-5)
    # new weights reach the kernel
    d1.set_weights([2.0 * d1.kernel, d1.bias + 0.1])
    (f2,) = model([x, None, None])
    params["W1"], params["b1"] = d1.kernel, d1.bias
    ref2 = O.topk_mlp_model(nl.astype(np.float64), {k: v.astype(np.float64) for k, v in params.items()})
    assert_forces_close("nlistnn_model_reweighted", f2.cpu().numpy(), ref2, atol=2e-5, rtol=5e-5)


def _liquid(htf, cuda, cells=8, steps=300, seed=9, dtype=torch.float32):
    """An equilibrated LJ liquid (rho 0.8442, kT ~ 1) produced by the stand-in MD itself: the kind of
    configuration the benchmark evaluates, with no overlapping pairs."""
    from hoomd_tf_amd import standin
    pos, L, a = standin.fcc_positions(cells, 0.8442)
    rng = np.random.default_rng(seed)
    pos = pos + 0.05 * a * rng.standard_normal(pos.shape)
    pos -= np.round(pos / L) * L
    sysm = standin.System(pos, L, dtype=dtype, device=cuda)
    sysm.randomize_velocities(kT=1.0, seed=seed)
    nl = standin.CellNlist(sysm, r_cut=3.0, r_buff=0.4, check_period=2)
    ctx = htf.Context(r_cut=3.0, nneighs=128, max_n=sysm.N, fused=2, scalar_dtype=dtype)
    ctx.set_potential(htf.Potential.lj())
    nve = standin.NVE(sysm, 0.005)
    for ts in range(steps):
        nl.compute(ts)
        ctx.compute_forces(ts, ctx.make_arrays(sysm.pos, sysm.N, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, sysm.force))
        f3 = sysm.force[:, :3]
        f3.mul_(torch.clamp(200.0 / f3.norm(dim=1, keepdim=True).clamp_min(1e-12), max=1.0))
        nve.step()
        v3 = sysm.vel[:, :3]
        v3.mul_(torch.sqrt(1.0 / ((v3 * v3).sum() / (3.0 * sysm.N))))
    nl.build()
    return sysm, nl, L


LIQUID = ("dense LJ liquid at kT = 1: a dozen first-shell neighbors push with |f_ij| = 20-90 each (sum_j |f_ij| ~ 300 per row) "
          "against net forces of O(10): ANY fp32 row sum, TensorFlow's included, sits ~1e-4 from the fp64 value")


def test_liquid_configuration_bounds(htf, cuda):
    """The benchmark's own kind of input, an equilibrated liquid, through every closed-form route: build + evaluate
    (two kernels), gather-evaluate in registers, the one-kernel step that also writes the tensor, the context, WCA,
    the virial.  The energy column meets SURVEY 8(c)'s bound as stated.  The force components cannot, for ANY fp32
    implementation: the test shows it with a second, independent fp32 evaluation (numpy's pairwise row sums of the
    oracle's fp32 restatement), which misses the fp64 forces by as much as the kernels do -- so the forces carry the
    named condition term, and the kernels must stay within 3x of that reference fp32 error."""
    sysm, nl, L = _liquid(htf, cuda)
    N, NN = sysm.N, 128
    pv = htf.ops.build_pair_vectors(sysm.pos, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, 3.0, NN)
    pv64 = pv.cpu().numpy().astype(np.float64)
    r = np.sqrt((pv64[:, :, :3] ** 2).sum(axis=2))
    assert int((r > 0).sum(axis=1).max()) < NN and r[r > 0].min() > 0.8   # a liquid: nothing overlaps
    ref_f, ref_v = O.lj_model(pv64, virial=True)
    cond = _cond_scale(pv64, _pair_forces_lj(pv64))
    fp32_err = np.abs(O.lj_model(pv.cpu().numpy()).astype(np.float64) - ref_f)[:, :3].max()
    strict = 1e-5 + 2e-5 * np.abs(ref_f)
    _record("liquid_fp32_restatement_vs_fp64", max_abs_err=fp32_err, median_cond=float(np.median(cond)),
            rows_outside_strict_bound=float((np.abs(O.lj_model(pv.cpu().numpy()) - ref_f) > strict).any(axis=1).mean()))
    assert fp32_err > 1e-5, "the fp32 restatement itself meets the strict bound here: drop the condition term"

    def check(name, f):
        f = f.cpu().numpy()
        assert_forces_close(name + "_energy", f[:, 3], ref_f[:, 3])                            # as stated
        assert_forces_close(name, f[:, :3], ref_f[:, :3], cond, cancelling_rows=LIQUID)
        assert np.abs(f[:, :3] - ref_f[:, :3]).max() <= 3.0 * fp32_err

    f, v = htf.ops.eval_forces(htf.Potential.lj(), pv, virial=True)
    check("liquid_lj_two_kernel", f)
    assert_forces_close("liquid_lj_virial", v.cpu().numpy(), ref_v, 3.0 * cond, atol=2e-5, cancelling_rows=LIQUID)
    fr = htf.ops.fused_forces(htf.Potential.lj(), sysm.pos, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, 3.0, NN)
    check("liquid_lj_registers", fr)
    pv2 = torch.empty_like(pv)
    fs = htf.ops.fused_forces(htf.Potential.lj(), sysm.pos, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, 3.0, NN, pair_vectors=pv2)
    assert torch.equal(pv2, pv)
    check("liquid_lj_one_kernel", fs)
    ctx = htf.Context(r_cut=3.0, nneighs=NN, max_n=N, fused=2)
    ctx.set_potential(htf.Potential.lj())
    force = torch.zeros((N, 4), dtype=torch.float32, device=cuda)
    ctx.compute_forces(0, ctx.make_arrays(sysm.pos, N, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, force))
    check("liquid_lj_context", force)
    # batches and row ranges of the liquid (N = 2 048 < 16 384 rows: the plain two-row form runs here; the four-rows-per-wave form
    # with merged tails is asserted against the oracle at its own size in
    # test_full_size_pair_vectors_every_row_bit_exact and test_liquid_at_headline_size): the tensor is bit-identical
    # however the step is cut, forces agree to rounding
    part = htf.ops.fused_forces(htf.Potential.lj(), sysm.pos, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, 3.0, NN, offset=5, batch_size=1001)
    assert float((part - fr[5:1006]).abs().max()) <= 2e-7 * float(cond.max())
    pv3 = torch.empty((1001, NN, 4), dtype=torch.float32, device=cuda)
    htf.ops.fused_forces(htf.Potential.lj(), sysm.pos, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, 3.0, NN, offset=5, batch_size=1001, pair_vectors=pv3)
    assert torch.equal(pv3, pv[5:1006])
    fw = htf.ops.eval_forces(htf.Potential.wca(1.0), pv).cpu().numpy()
    rw = O.wca_model(pv64, 1.0)
    s, t, rp, cnd = O._rinv_and_grad_factor(pv64)
    assert_forces_close("liquid_wca", fw, rw, _cond_scale(pv64, 2 * O._grad_from_dEds(6 * s ** 5, s, t, rp, cnd)), cancelling_rows=LIQUID)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["float32", "float64"])
def test_liquid_at_headline_size(htf, cuda, dtype):
    """The equilibrated liquid at 55 296 rows (fcc 24^3), a batch that takes the kernel the bench times
    (fused_forces_tails_kernel<LJ, STORE, 4, float>: four rows per wave, merged tails; launch_fused's threshold is 49 152
    rows; ``float64`` = HOOMD built in double precision, TensorflowCompute.h:117-124, the <LJ, STORE, 4, double> form of
    ``bench.py --f64``): 512 sampled rows against the oracle -- energy as stated, forces with the named condition term and
    within 3x of an independent fp32 evaluation's own error -- with and without the tensor, and from the context.  The
    fp64 wire's tensor is the fp64 reference rounded once (simmodel.py:226-227's tf.cast), bit for bit."""
    sysm, nl, L = _liquid(htf, cuda, cells=24, steps=200, seed=10, dtype=dtype)
    N, NN = sysm.N, 128
    assert N == 55296 and sysm.pos.dtype == dtype
    pv = htf.ops.build_pair_vectors(sysm.pos, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, 3.0, NN, out_dtype=torch.float32)
    rows = np.random.default_rng(6).choice(N, 512, replace=False)
    if dtype == torch.float64:
        from oracle import c_oracle
        ref64 = c_oracle.prepare_neighbors(c_oracle.load(), sysm.pos.cpu().numpy(), nl.n_neigh.cpu().numpy().view(np.uint32),
                                           nl.head_list.cpu().numpy().view(np.uint32), nl.nlist.cpu().numpy().view(np.uint32),
                                           O.make_box(L, dtype=np.float64), 3.0, NN)
        np.testing.assert_array_equal(pv.cpu().numpy(), ref64.astype(np.float32))     # every row
        del ref64
    pv32 = pv.cpu().numpy()[rows]
    pv64 = pv32.astype(np.float64)
    r = np.sqrt((pv64[:, :, :3] ** 2).sum(axis=2))
    assert int((r > 0).sum(axis=1).max()) < NN and r[r > 0].min() > 0.8
    ref_f = O.lj_model(pv64)
    cond = _cond_scale(pv64, _pair_forces_lj(pv64))
    fp32_err = np.abs(O.lj_model(pv32).astype(np.float64) - ref_f)[:, :3].max()
    got = {}
    got["registers"] = htf.ops.fused_forces(htf.Potential.lj(), sysm.pos, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, 3.0, NN)
    pv2 = torch.empty_like(pv)
    got["one_kernel"] = htf.ops.fused_forces(htf.Potential.lj(), sysm.pos, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, 3.0, NN, pair_vectors=pv2)
    assert torch.equal(pv2, pv)
    ctx = htf.Context(r_cut=3.0, nneighs=NN, max_n=N, fused=2, scalar_dtype=dtype)
    ctx.set_potential(htf.Potential.lj())
    force = torch.zeros((N, 4), dtype=dtype, device=cuda)
    ctx.compute_forces(0, ctx.make_arrays(sysm.pos, N, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, force))
    assert torch.equal(ctx.nlist_buffer(N, cuda), pv)
    got["context"] = force
    tag = "liquid16k_lj" if dtype == torch.float32 else "liquid16k_f64wire_lj"
    for name, f in got.items():
        assert f.dtype == dtype
        f = f.cpu().numpy()[rows]
        assert_forces_close("%s_%s_energy" % (tag, name), f[:, 3], ref_f[:, 3])
        assert_forces_close("%s_%s" % (tag, name), f[:, :3], ref_f[:, :3], cond, cancelling_rows=LIQUID)
        assert np.abs(f[:, :3] - ref_f[:, :3]).max() <= 3.0 * fp32_err


def test_profile_read_survives_an_error_return(htf, cuda):
    """A batch that returns early (nlist overflow) leaves its closing event unrecorded; htf_profile_read must
    skip that scope, keep working, and count only the batches that completed."""
    pos, types, L, nn, head, nl = _system(4, 1.6, 0.08, 11, 3.4, np.float32, three_d=True)
    box = O.make_box(L, dtype=np.float32)
    p4, dnn, dhead, dnl = _to_dev(htf, pos, types, nn, head, nl, np.float32, cuda)
    N = len(pos)
    force = torch.zeros((N, 4), dtype=torch.float32, device=cuda)
    for fused in (0, 2):
        ctx = htf.Context(r_cut=3.0, nneighs=8, max_n=N, check_nlist=True, fused=fused)   # NN = 8 overflows on this system
        ctx.set_potential(htf.Potential.lj())
        ctx.profile_enable(True)
        arr = ctx.make_arrays(p4, N, dnn, dhead, dnl, box, force)
        with pytest.raises(htf.NlistOverflowError):
            ctx.compute_forces(0, arr)
        ok = htf.Context(r_cut=3.0, nneighs=128, max_n=N, fused=fused)
        ok.set_potential(htf.Potential.lj())
        # the failed context still reads (0 or 1 completed scopes, never an error) ...
        b, e, n = ctx.profile_read()
        assert n <= 1 and b >= 0 and e >= 0
        # ... and profiles normally afterwards
        ctx2 = htf.Context(r_cut=3.0, nneighs=128, max_n=N, fused=fused)
        ctx2.set_potential(htf.Potential.lj())
        ctx2.profile_enable(True)
        arr2 = ctx2.make_arrays(p4, N, dnn, dhead, dnl, box, force)
        for ts in range(3):
            ctx2.compute_forces(ts, arr2)
        b, e, n = ctx2.profile_read()
        assert n == 3 and e > 0


def test_nlist_buffer_written_by_the_caller_needs_a_reset(htf, cuda):
    """The context re-zeroes only the slots a row lost since the previous step (htf_amd.h, htf_reset_nlist_buffer):
    a caller that scribbles into the zero tail must say so, and then gets a clean tensor again."""
    pos, types, L, nn, head, nl = _system(4, 1.6, 0.08, 11, 3.4, np.float32, three_d=True)
    box = O.make_box(L, dtype=np.float32)
    p4, dnn, dhead, dnl = _to_dev(htf, pos, types, nn, head, nl, np.float32, cuda)
    N, NN = len(pos), 128
    ref = O.prepare_neighbors(pos, types, nn, head, nl, box, 3.0, NN)
    force = torch.zeros((N, 4), dtype=torch.float32, device=cuda)
    for fused in (0, 2):
        ctx = htf.Context(r_cut=3.0, nneighs=NN, max_n=N, fused=fused)
        ctx.set_potential(htf.Potential.lj())
        arr = ctx.make_arrays(p4, N, dnn, dhead, dnl, box, force)
        ctx.compute_forces(0, arr)
        view = ctx.nlist_buffer(N, cuda)
        np.testing.assert_array_equal(view.cpu().numpy(), ref)
        view[:, NN - 1, :] = 5.0                      # a model writes into the tensor (allowed upstream)
        ctx.reset_nlist_buffer()
        ctx.compute_forces(1, arr)
        np.testing.assert_array_equal(ctx.nlist_buffer(N, cuda).cpu().numpy(), ref)


def _jittered(standin, cuda, lattice, cells, seed, dtype=torch.float32):
    pos, L, a = (standin.sc_positions if lattice == "sc" else standin.fcc_positions)(cells, 0.8442)
    rng = np.random.default_rng(seed)
    pos = pos + 0.05 * a * rng.standard_normal(pos.shape)
    pos -= np.round(pos / L) * L
    sysm = standin.System(pos, L, dtype=dtype, device=cuda)
    nl = standin.CellNlist(sysm, r_cut=3.0, r_buff=0.4)
    nl.build()
    return sysm, nl, L


def test_full_size_c2_wca(htf, cuda):
    """BASELINE configs[1] at its own size -- WCARepulsion, 32 768 particles (sc 32^3), NN 128 -- from HOOMD-layout
    arrays through the context: a row sample against the oracle, and size-independent properties."""
    from hoomd_tf_amd import standin
    sysm, nl, L = _jittered(standin, cuda, "sc", 32, 2)
    N, NN = sysm.N, 128
    assert N == 32768
    pot = htf.Potential.wca(1.0)
    forces = {}
    for fused in (0, 1, 2):
        ctx = htf.Context(r_cut=3.0, nneighs=NN, max_n=N, fused=fused)
        ctx.set_potential(pot)
        f = torch.zeros((N, 4), dtype=torch.float32, device=cuda)
        ctx.compute_forces(0, ctx.make_arrays(sysm.pos, N, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, f))
        forces[fused] = f
        if fused != 1:
            pv = ctx.nlist_buffer(N, cuda).clone()
    rows = np.random.default_rng(0).choice(N, 512, replace=False)
    p32 = sysm.pos.cpu().numpy()[:, :3]
    ref_pv = O.prepare_neighbors(p32, np.zeros(N, np.int32), nl.n_neigh.cpu().numpy().view(np.uint32),
                                 nl.head_list.cpu().numpy().view(np.uint32), nl.nlist.cpu().numpy().view(np.uint32),
                                 O.make_box(L, dtype=np.float32), 3.0, NN)[rows]
    np.testing.assert_array_equal(pv.cpu().numpy()[rows], ref_pv)                   # bit-exact pair vectors
    sub = ref_pv.astype(np.float64)
    ref = O.wca_model(sub, 1.0)
    s, t, rp, cnd = O._rinv_and_grad_factor(sub)
    cond = _cond_scale(sub, 2 * O._grad_from_dEds(6 * s ** 5, s, t, rp, cnd))
    for fused, f in forces.items():
        assert_forces_close("c2_wca_fused%d" % fused, f.cpu().numpy()[rows], ref, cond, cancelling_rows=CONTACTS)
    # properties: Newton's third law over the whole periodic system (every pair appears in both rows), a
    # repulsive energy, determinism
    # (not exactly zero even in exact arithmetic: safe_norm adds 1e-7 to every component, so F_ij + F_ji is
    #  2e-7 * |f_ij| / r, the reference's own asymmetry)
    tot = forces[2][:, :3].double().sum(dim=0).abs().max().item()
    assert tot <= 2e-6 * forces[2][:, :3].abs().double().sum().item()
    assert float(forces[2][:, 3].min()) >= 0.0 and float(forces[2][:, 3].max()) <= 10.0 * NN
    f_again = torch.zeros_like(forces[2])
    ctx.compute_forces(0, ctx.make_arrays(sysm.pos, N, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, f_again))
    assert torch.equal(f_again, forces[2])


def test_full_size_c4_eds_sweep(htf, cuda):
    """BASELINE configs[3] at its own size -- 262 144 particles (sc 64^3), NN 128, LJ + Gaussian CV channel + RDF
    histogram in ONE kernel: row samples of both force sets against the oracle, the CV against an fp64 sum, the
    histogram against its total-count identity and against the stand-alone histogram kernel."""
    from hoomd_tf_amd import standin
    sysm, nl, L = _jittered(standin, cuda, "sc", 64, 4)
    N, NN = sysm.N, 128
    assert N == 262144
    lj, gauss = htf.Potential.lj(), htf.Potential.gauss(1.1, 0.05, 1.0)
    pv = torch.zeros((N, NN, 4), dtype=torch.float32, device=cuda)
    npart = htf.ops.num_partials_fused(N)
    partials = torch.empty(npart, dtype=torch.float32, device=cuda)
    hist = torch.zeros(102, dtype=torch.int32, device=cuda)
    fa, fb = htf.ops.build_eval_forces2(lj, gauss, sysm.pos, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, 3.0, NN, partials=partials,
                                        rdf=(0.0, 3.5, hist), pair_vectors=pv)
    cv = torch.zeros(1, dtype=torch.float32, device=cuda)
    htf.ops.reduce_partials(partials, npart, 1.0 / N, cv)
    rows = np.random.default_rng(1).choice(N, 384, replace=False)
    sub = pv.cpu().numpy()[rows].astype(np.float64)
    assert_forces_close("c4_lj_rows", fa.cpu().numpy()[rows], O.lj_model(sub), _cond_scale(sub, _pair_forces_lj(sub)), cancelling_rows=CONTACTS)
    _, g = O.gauss_pair_terms(sub, 1.1, 0.05)
    assert_forces_close("c4_gauss_rows", fb.cpu().numpy()[rows], O.gauss_model(sub, 1.1, 0.05, 1.0), np.abs(2 * g).sum(axis=(1, 2)),
                        ctol=4e-6, cancelling_rows=CONTACTS)
    # the tensor of the one-kernel sweep == the build kernel's
    ref_pv = htf.ops.build_pair_vectors(sysm.pos, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, 3.0, NN)
    assert torch.equal(pv, ref_pv)
    # CV = (1/N) sum of the Gaussian energy column, in fp64
    np.testing.assert_allclose(float(cv), fb[:, 3].double().sum().item() / N, rtol=2e-6)
    # histogram: every slot of the tensor lands in exactly one bin; equal to the stand-alone kernel's
    assert int(hist.sum()) == N * NN
    h2 = torch.zeros(102, dtype=torch.int32, device=cuda)
    import ctypes as C
    from hoomd_tf_amd._lib import lib, check
    check(lib.htf_rdf_histogram(pv.data_ptr(), 0, N, NN, 0.0, 3.5, 102, None, 0, -1, -1, h2.data_ptr(),
                                C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    assert torch.equal(hist, h2)
    # ... and to the oracle's tf.histogram_fixed_width over ALL 33.5 M slots, bin for bin: pairs within an ulp of a bin
    # edge are decided by the roundings of tf.norm and of (v - lo) / (hi - lo) (uncontracted products and sums,
    # correctly rounded sqrt and division) -- the first version of this sweep multiplied by nbins / (hi - lo) and let
    # the compiler fuse x*x + y*y + z*z, and disagreed on 4 pairs at this size while passing every fixture-size test
    x3 = pv.cpu().numpy()[:, :, :3]
    r_all = np.sqrt((x3 * x3).sum(axis=2, dtype=np.float32)).astype(np.float32)
    np.testing.assert_array_equal(hist.cpu().numpy(), O.histogram_fixed_width(r_all, np.array([0.0, 3.5], np.float32), 102))


@pytest.mark.parametrize("lattice,cells,dtype", [("sc", 32, torch.float32), ("fcc", 32, torch.float32), ("fcc", 32, torch.float64)],
                         ids=["sc-32", "fcc-32", "fcc-32-float64"])
def test_full_size_pair_vectors_every_row_bit_exact(htf, cuda, lattice, cells, dtype):
    """prepareNeighbors at C2 (32 768) and C3 (131 072) size, EVERY row bit for bit: the build kernel, the one-kernel
    step's tensor and the four-rows-per-wave form against the C restatement (itself bit-exact against the numpy
    oracle, tests/test_oracle_c.py).  ``float64``: HOOMD built in double precision (TensorflowCompute.h:117-124) -- the
    tensor is the fp64 restatement rounded once to fp32 (simmodel.py:226-227), and the forces of
    fused_forces_tails_kernel<LJ, STORE, 4, double> (``bench.py --f64``) meet the oracle at the kernel's own size."""
    from hoomd_tf_amd import standin
    from oracle import c_oracle
    sysm, nl, L = _jittered(standin, cuda, lattice, cells, 7, dtype=dtype)
    N, NN = sysm.N, 128
    f64 = dtype == torch.float64
    ref = c_oracle.prepare_neighbors(c_oracle.load(), sysm.pos.cpu().numpy(), nl.n_neigh.cpu().numpy().view(np.uint32),
                                     nl.head_list.cpu().numpy().view(np.uint32), nl.nlist.cpu().numpy().view(np.uint32),
                                     O.make_box(L, dtype=np.float64 if f64 else np.float32), 3.0, NN)
    if f64:
        assert ref.dtype == np.float64
        ref = ref.astype(np.float32)
    pv = htf.ops.build_pair_vectors(sysm.pos, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, 3.0, NN, out_dtype=torch.float32)
    np.testing.assert_array_equal(pv.cpu().numpy(), ref)
    pv2 = torch.full_like(pv, 3.0)
    forces = {}
    forces["one_kernel"] = htf.ops.fused_forces(htf.Potential.lj(), sysm.pos, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, 3.0, NN, pair_vectors=pv2)
    assert torch.equal(pv2, pv)
    forces["registers"] = htf.ops.fused_forces(htf.Potential.lj(), sysm.pos, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, 3.0, NN)
    ctx = htf.Context(r_cut=3.0, nneighs=NN, max_n=N, fused=2, scalar_dtype=dtype)
    ctx.set_potential(htf.Potential.lj())
    f = torch.zeros((N, 4), dtype=dtype, device=cuda)
    for ts in range(2):  # second call: the delta zero-fill path (rows keep their live counts)
        f.zero_()
        ctx.compute_forces(ts, ctx.make_arrays(sysm.pos, N, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, f))
        assert torch.equal(ctx.nlist_buffer(N, cuda), pv)
        forces["context_fused2_call%d" % ts] = f.clone()
    ctx1 = htf.Context(r_cut=3.0, nneighs=NN, max_n=N, fused=1, scalar_dtype=dtype)
    ctx1.set_potential(htf.Potential.lj())
    f1 = torch.zeros((N, 4), dtype=dtype, device=cuda)
    ctx1.compute_forces(0, ctx1.make_arrays(sysm.pos, N, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, f1))
    forces["context_fused1"] = f1
    # The LJ FORCES of the kernel the bench times (N >= 16 384 rows: fused_forces_tails_kernel<LJ, STORE, 4, float / double>,
    # four rows per wave with merged tails; with and without the tensor, from the stateless entry point and from the context
    # at fused = 2 / 1) against the oracle on 512 sampled rows of the bit-exact tensor; energy column as stated.
    rows = np.random.default_rng(5).choice(N, 512, replace=False)
    sub = ref[rows].astype(np.float64)
    ref_f = O.lj_model(sub)
    cond = _cond_scale(sub, _pair_forces_lj(sub))
    s_, _, _, _ = O._rinv_and_grad_factor(sub)
    e_cond = (2.0 * (s_ ** 12 + s_ ** 6)).sum(axis=1)   # sum_j |e_ij| scale of the energy column's row sum (jitter leaves contacts at r ~ 0.8)
    for name, ff in forces.items():
        assert ff.dtype == dtype
        got = ff.cpu().numpy()[rows]
        tag = "full_%s%d%s_lj_%s" % (lattice, cells, "_f64wire" if f64 else "", name)
        assert_forces_close(tag + "_energy", got[:, 3], ref_f[:, 3], e_cond, cancelling_rows=CONTACTS)
        assert_forces_close(tag, got[:, :3], ref_f[:, :3], cond, cancelling_rows=CONTACTS)
    # every row, not only the sample: the whole-system energy against the C restatement's fp64 sum over the tensor
    e_ref = float(O.lj_model(ref[::64].astype(np.float64))[:, 3].sum())
    np.testing.assert_allclose(forces["context_fused2_call1"][::64, 3].double().sum().item(), e_ref, rtol=1e-5)


@pytest.mark.parametrize("size", ["two-row form (500 rows)", "two rows, merged tails (16384 rows)", "four rows, merged tails (55296 rows)"])
def test_dropped_candidates_contribute_exact_zeros(htf, cuda, size):
    """The fused kernels evaluate every candidate of a trip, dropped ones too -- at a far point where the potential vanishes
    identically, or with their results selected away (pair_eval_if).  A list whose every candidate lies beyond r_cut must
    therefore leave forces, energies and the tensor EXACTLY zero, for every closed-form potential, in every fused form."""
    from hoomd_tf_amd import standin
    cells = {"two-row form (500 rows)": 5, "two rows, merged tails (16384 rows)": 16, "four rows, merged tails (55296 rows)": 24}[size]
    pos, L, a = standin.fcc_positions(cells, 0.8442)
    rng = np.random.default_rng(3)
    pos = pos + 0.03 * a * rng.standard_normal(pos.shape)
    pos -= np.round(pos / L) * L
    sysm = standin.System(pos, L, dtype=torch.float32, device=cuda)
    nl = standin.CellNlist(sysm, r_cut=2.5, r_buff=0.3)
    nl.build()
    N, NN = sysm.N, 64
    assert int(nl.n_neigh.min()) > 20  # every row has candidates ...
    r_cut = 0.5                          # ... and none of them is within the cutoff (nearest neighbors sit at ~1.1)
    pots = [htf.Potential.lj(), htf.Potential.wca(1.0), htf.Potential.lj_param(1.3, 0.9),
            htf.Potential.rinv_poly([1.0, -0.5], [12, 1]), htf.Potential.simple()]
    for pot in pots:
        for with_tensor in (True, False):
            pv = torch.full((N, NN, 4), 7.0, dtype=torch.float32, device=cuda) if with_tensor else None
            f = htf.ops.fused_forces(pot, sysm.pos, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, r_cut, NN, pair_vectors=pv)
            assert torch.count_nonzero(f) == 0, (pot.kind, with_tensor, float(f.abs().max()))
            if with_tensor:
                assert torch.count_nonzero(pv) == 0
