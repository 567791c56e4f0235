"""HOOMD stand-in kernels on the GPU: the binned neighbor search must produce the same
FULL neighbor sets as the O(N^2) brute force, and the leapfrog step must match numpy."""
import numpy as np
import pytest
import torch

from helpers import brute_nlist, sq_lattice

pytestmark = pytest.mark.gpu


def _rows(nn, head, nl):
    return [np.sort(nl[int(head[i]): int(head[i]) + int(nn[i])]) for i in range(len(nn))]


# 5 lattice cells: 5 search cells along an axis, the near image of a candidate needs rint(); 7 and 10: >= 7 search cells,
# the image follows from the candidate's cell (build_nlist_kernel<.., SHIFT>)
@pytest.mark.parametrize("cells", [5, 7, 10])
@pytest.mark.parametrize("tdt", [torch.float32, torch.float64])
def test_cell_nlist_matches_brute_force(htf, cuda, tdt, cells):
    from hoomd_tf_amd import standin
    pos, L, a = standin.fcc_positions(cells, 0.8442)
    rng = np.random.default_rng(0)
    pos = pos + 0.08 * a * rng.standard_normal(pos.shape)
    pos -= np.round(pos / L) * L
    sysm = standin.System(pos, L, dtype=tdt, device=cuda)
    nl = standin.CellNlist(sysm, r_cut=2.5, r_buff=0.4)
    nl.build()
    p = sysm.pos.cpu().numpy()[:, :3].astype(np.float64)
    bn, bh, bl = brute_nlist(p, L, 2.9)
    got = _rows(nl.n_neigh.cpu().numpy(), nl.head_list.cpu().numpy(), nl.nlist.cpu().numpy())
    ref = _rows(bn, bh, bl)
    mism = 0
    for g, r in zip(got, ref):
        if len(g) != len(r) or np.any(g != r):
            # pairs within 1e-5 of r_list may flip with fp32 rounding; anything else is a bug
            sym = set(g.tolist()) ^ set(r.tolist())
            mism += len(sym)
    assert mism <= (4 if tdt == torch.float32 else 0)
    assert nl.pitch >= int(bn.max())


def test_cell_nlist_2d_and_small_pitch_regrow(htf, cuda):
    from hoomd_tf_amd import standin
    pos, L = sq_lattice(16, 2.0)
    rng = np.random.default_rng(1)
    pos[:, :2] += 0.1 * rng.standard_normal((256, 2))
    sysm = standin.System(pos, L, dtype=torch.float64, device=cuda)
    nl = standin.CellNlist(sysm, r_cut=3.0, r_buff=0.4, pitch=8)  # too small on purpose
    nl.build()
    bn, bh, bl = brute_nlist(sysm.pos.cpu().numpy()[:, :3], L, 3.4)
    assert nl.pitch >= bn.max()
    got = _rows(nl.n_neigh.cpu().numpy(), nl.head_list.cpu().numpy(), nl.nlist.cpu().numpy())
    for g, r in zip(got, _rows(bn, bh, bl)):
        np.testing.assert_array_equal(g, r)


def test_nve_step_and_displacement(htf, cuda):
    from hoomd_tf_amd import standin
    pos, L, a = standin.fcc_positions(3, 0.8)
    sysm = standin.System(pos, L, dtype=torch.float64, device=cuda, types=np.arange(len(pos)) % 4)
    sysm.randomize_velocities(1.0, 5)
    g = torch.Generator().manual_seed(0)
    sysm.force[:, :3] = torch.randn(sysm.N, 3, generator=g, dtype=torch.float64).to(cuda)
    p0, v0, f0 = sysm.pos.cpu().numpy().copy(), sysm.vel.cpu().numpy().copy(), sysm.force.cpu().numpy().copy()
    nl = standin.CellNlist(sysm, r_cut=2.0, r_buff=0.4)
    nl.build()
    nve = standin.NVE(sysm, 0.05)
    nve.step()
    v1 = v0[:, :3] + 0.05 * f0[:, :3]
    x1 = p0[:, :3] + 0.05 * v1
    x1 = x1 - np.floor((x1 + L / 2) / L) * L
    np.testing.assert_allclose(sysm.vel.cpu().numpy()[:, :3], v1, rtol=1e-14, atol=1e-14)
    np.testing.assert_allclose(sysm.pos.cpu().numpy()[:, :3], x1, rtol=1e-13, atol=1e-13)
    np.testing.assert_array_equal(sysm.types_numpy(), np.arange(sysm.N) % 4)  # w (type bits) untouched
    # distance check trips once something moved more than r_buff/2
    assert not nl.needs_update() or np.max(np.linalg.norm(0.05 * v1, axis=1)) > 0.2
    for _ in range(20):
        nve.step()
    assert nl.needs_update()


@pytest.mark.parametrize("mode", ["plain", "virial", "fused", "fused-store", "mlp"])
def test_compute_forces_in_row_ranges_equals_whole(htf, cuda, mode):
    """htf_compute_forces_rows: a step computed as [0, n1) then [n1, N) (interior rows while the
    halo is in flight, boundary rows after it) leaves forces, virial and the context's
    pair-vector / positions buffers exactly as one htf_compute_forces call does."""
    from hoomd_tf_amd import standin
    pos, L, a = standin.fcc_positions(6, 0.8442)
    rng = np.random.default_rng(2)
    pos = pos + 0.06 * a * rng.standard_normal(pos.shape)
    pos -= np.round(pos / L) * L
    sysm = standin.System(pos, L, dtype=torch.float32, device=cuda)
    nl = standin.CellNlist(sysm, r_cut=2.5, r_buff=0.3)
    nl.build()
    N, NN = sysm.N, 80
    pot = htf.Potential.pair_mlp(__import__("hoomd_tf_amd").initializers.mlp_params(seed=3), 0.0, 3.0, activation="tanh") \
        if mode == "mlp" else htf.Potential.lj()
    virial = mode == "virial"

    def run(split):
        ctx = htf.Context(r_cut=2.5, nneighs=NN, scalar_dtype=torch.float32, max_n=N, virial=virial,
                          fused={"fused": 1, "fused-store": 2}.get(mode, 0))
        ctx.set_potential(pot)
        force = torch.zeros((N, 4), device=cuda)
        vir = torch.zeros(6 * N, device=cuda) if virial else None
        arr = ctx.make_arrays(sysm.pos, N, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, force, vir, N)
        for ts in range(2):  # second pass exercises the delta zero-fill bookkeeping per row
            if virial:
                vir.zero_()
            if split:
                n1 = 317
                ctx.compute_forces(ts, arr, rows=(0, n1))
                ctx.compute_forces(ts, arr, rows=(n1, N - n1))
            else:
                ctx.compute_forces(ts, arr)
        torch.cuda.synchronize()
        bufs = None if mode == "fused" else (ctx.nlist_buffer(N, cuda).clone(), ctx.positions_buffer(N, cuda).clone())
        return force.clone(), (vir.clone() if virial else None), bufs

    f0, v0, b0 = run(False)
    f1, v1, b1 = run(True)
    if mode == "fused-store":  # the one-kernel mode keeps the side buffers of the two-kernel mode
        mode = "plain"
        _, _, b_plain = run(False)
        assert torch.equal(b_plain[0], b0[0]) and torch.equal(b_plain[1], b0[1])
    assert torch.equal(f0, f1) and float(f0.abs().sum()) > 0
    if virial:
        assert torch.equal(v0, v1) and float(v0.abs().sum()) > 0
    if b0 is not None:
        assert torch.equal(b0[0], b1[0]) and torch.equal(b0[1], b1[1])
    with pytest.raises(ValueError):
        ctx = htf.Context(r_cut=2.5, nneighs=NN, scalar_dtype=torch.float32, max_n=N)
        ctx.set_potential(pot)
        arr = ctx.make_arrays(sysm.pos, N, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, torch.zeros((N, 4), device=cuda))
        ctx.compute_forces(0, arr, rows=(N - 3, 10))


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [0, 2])
def test_nve_energy_is_conserved(htf, cuda, fused):
    """End-to-end sanity of the whole step (neighbor search with rebuilds, pair-vector build, LJModel
    forces, leapfrog): 3000 NVE steps of a 4000-particle liquid.  LJModel's energy is cut (not shifted)
    at r_cut, so the total energy random-walks by the jumps at the cutoff; a wrong force, a missed
    neighbor or a stale list would show as drift orders of magnitude larger."""
    from hoomd_tf_amd import standin
    pos, L, a = standin.fcc_positions(10, 0.8442)
    rng = np.random.default_rng(5)
    pos = pos + 0.03 * a * rng.standard_normal(pos.shape)
    pos -= np.round(pos / L) * L
    s = standin.System(pos, L, dtype=torch.float32, device=cuda)
    s.randomize_velocities(kT=0.8, seed=5)
    nl = standin.CellNlist(s, r_cut=3.0, r_buff=0.4, check_period=1)
    nl.build()
    ctx = htf.Context(r_cut=3.0, nneighs=128, scalar_dtype=torch.float32, max_n=s.N, fused=fused)
    ctx.set_potential(htf.Potential.lj())
    nve = standin.NVE(s, 0.002)
    arr = {"a": ctx.make_arrays(s.pos, s.N, nl.n_neigh, nl.head_list, nl.nlist, s.box, s.force), "b": nl.n_builds}

    def step(ts):
        nl.compute(ts)
        if nl.n_builds != arr["b"]:
            arr["a"] = ctx.make_arrays(s.pos, s.N, nl.n_neigh, nl.head_list, nl.nlist, s.box, s.force)
            arr["b"] = nl.n_builds
        ctx.compute_forces(ts, arr["a"])
        nve.step()

    def energy():
        # leapfrog: velocities live at half steps; the kinetic energy at the force's time needs v(t) --
        # average of the neighbouring half-step energies is accurate to O(dt^2)
        pe = float(s.force[:, 3].double().sum())
        ke = 0.5 * float((s.vel[:, :3].double() ** 2).sum())
        return pe, ke

    for ts in range(300):  # settle
        step(ts)
    tot = []
    for ts in range(300, 3300):
        step(ts)
        if ts % 50 == 0:
            pe, ke = energy()
            tot.append((pe + ke) / s.N)
    tot = np.array(tot)
    assert nl.n_builds > 20
    assert np.all(np.isfinite(tot))
    drift = abs(np.polyfit(np.arange(len(tot)), tot, 1)[0]) * len(tot)
    print("energy per particle: mean %.5f drift %.2e std %.2e rebuilds %d" % (tot.mean(), drift, tot.std(), nl.n_builds))
    assert drift < 2e-3 and tot.std() < 2e-3, (drift, tot.std(), tot[:5], tot[-5:])  # measured 1.5e-4 / 3e-4


def test_device_decided_rebuild_equals_host_decided(htf, cuda):
    """The distance check's verdict taken by the gated kernels themselves (CellNlist(device_decision=True):
    no read-back in the step loop) must reproduce the host-decided run bit for bit: same rebuild steps,
    same neighbor rows, same trajectory."""
    from hoomd_tf_amd import standin

    def run(device_decision):
        pos, L, a = standin.fcc_positions(8, 0.8442)
        rng = np.random.default_rng(5)
        pos = pos + 0.05 * a * rng.standard_normal(pos.shape)
        pos -= np.round(pos / L) * L
        sysm = standin.System(pos, L, dtype=torch.float32, device=cuda)
        sysm.randomize_velocities(kT=1.2, seed=5)
        nl = standin.CellNlist(sysm, r_cut=2.5, r_buff=0.4, check_period=2, device_decision=device_decision)
        ctx = htf.Context(r_cut=2.5, nneighs=96, max_n=sysm.N, fused=2)
        ctx.set_potential(htf.Potential.lj())
        nve = standin.NVE(sysm, 0.004)
        snaps = []
        for ts in range(120):
            nl.compute(ts)
            arr = ctx.make_arrays(sysm.pos, sysm.N, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, sysm.force)
            ctx.compute_forces(ts, arr)
            f3 = sysm.force[:, :3]
            f3.mul_(torch.clamp(100.0 / f3.norm(dim=1, keepdim=True).clamp_min(1e-12), max=1.0))
            nve.step()
            if ts % 20 == 19:
                snaps.append((sysm.pos.clone(), nl.n_neigh.clone(), nl.nlist.clone()))
        torch.cuda.synchronize()
        nl._poll_overflow()
        builds = nl.n_builds + nl.device_builds()
        return snaps, builds, nl

    host, hb, _ = run(False)
    dev, db, nl = run(True)
    assert hb == db and hb >= 4, (hb, db)
    assert nl.n_builds == 1  # only the first build went through the host path
    for (p0, n0, l0), (p1, n1, l1) in zip(host, dev):
        assert torch.equal(p0, p1) and torch.equal(n0, n1)
        pitch = nl.pitch
        # rows are compared over their live entries (the tail of a row is stale scratch)
        live = torch.arange(pitch, device=cuda)[None, :] < n0[:, None]
        assert torch.equal(l0.view(-1, pitch)[live], l1.view(-1, pitch)[live])


def test_device_decided_rebuild_reports_row_overflow_late(htf, cuda):
    from hoomd_tf_amd import standin
    pos, L, a = standin.fcc_positions(6, 0.8442)
    sysm = standin.System(pos, L, dtype=torch.float32, device=cuda)
    nl = standin.CellNlist(sysm, r_cut=2.5, r_buff=0.4, check_period=1, device_decision=True)
    nl.compute(0)                                   # host-path first build sizes the pitch
    nl.pitch = 8                                    # pretend the rows were sized far too small
    nl.n_neigh = torch.empty(sysm.N, dtype=torch.int32, device=cuda)
    nl.head_list = torch.empty(sysm.N, dtype=torch.int32, device=cuda)
    nl.nlist = torch.empty(sysm.N * 8, dtype=torch.int32, device=cuda)
    sysm.pos[:, :3] += 0.3                          # everything moved: the gate opens
    nl.compute(1)
    with pytest.raises(RuntimeError, match="row overflow"):
        nl.compute(2)


def test_graphed_run_equals_stepwise(htf, cuda):
    """Simulation.run(n, graph=True): whole check periods replayed from a hipGraph (one launch per cycle) must give
    the trajectory of the step-by-step loop bit for bit -- the captured launch sequence IS the step, every decision in
    it (distance check, gated rebuild) is taken on the device."""
    from hoomd_tf_amd import standin

    class LJModel(htf.SimModel):
        def compute(self, nlist, positions, box):
            rinv = htf.nlist_rinv(nlist)
            inv_r6 = rinv**6
            energy = htf.reduce_sum(4.0 / 2.0 * (inv_r6 * inv_r6 - inv_r6), axis=1)
            return htf.compute_nlist_forces(nlist, energy)

    def run(graph):
        pos, L, a = standin.fcc_positions(6, 0.8442)
        rng = np.random.default_rng(11)
        pos = pos + 0.03 * a * rng.standard_normal(pos.shape)
        pos -= np.round(pos / L) * L
        sysm = standin.System(pos, L, dtype=torch.float32, device=cuda)
        sysm.randomize_velocities(kT=1.0, seed=11)
        sim = standin.Simulation(sysm)
        sim.integrate_nve(0.004)
        tfc = htf.tfcompute(LJModel(96))
        cell = sim.nlist_cell(r_buff=0.4, check_period=3, pitch=160)
        tfc.attach(cell, r_cut=2.5)
        sim.run(7)  # traces, installs the plan, first device-side check
        sim.run(200, graph=graph)
        sim.run(5)
        torch.cuda.synchronize()
        cell._poll_overflow()
        return sysm.pos.clone(), sysm.vel.clone(), tfc.force.clone(), cell.device_builds(), getattr(sim, "_graph", None), sysm.timestep

    p0, v0, f0, b0, g0, t0 = run(False)
    p1, v1, f1, b1, g1, t1 = run(True)
    assert g0 is None and g1 is not None, "the graphed run did not capture"
    assert t0 == t1 == 212 and b0 == b1 and b0 >= 5, (t0, t1, b0, b1)
    assert torch.equal(p0, p1) and torch.equal(v0, v1) and torch.equal(f0, f1)


def test_default_run_picks_the_replay_and_recaptures_on_a_new_timestep(htf, cuda):
    """Simulation.run(n) with graph=None: a run of >= 256 qualifying steps times its own first steps both ways (stepwise, then
    replayed from a hipGraph) and keeps the faster -- at 864 particles the step is bound by the host's enqueue and the replay
    wins by ~2x (``sim.graph_choice`` records both figures); shorter runs step; a captured launch carries dt by value, so a
    change of it must re-capture -- the trajectory through the measurement and the dt change equals the stepwise one bit for bit."""
    from hoomd_tf_amd import standin

    class LJModel(htf.SimModel):
        def compute(self, nlist, positions, box):
            rinv = htf.nlist_rinv(nlist)
            inv_r6 = rinv**6
            energy = htf.reduce_sum(4.0 / 2.0 * (inv_r6 * inv_r6 - inv_r6), axis=1)
            return htf.compute_nlist_forces(nlist, energy)

    def run(graph):
        pos, L, a = standin.fcc_positions(6, 0.8442)
        rng = np.random.default_rng(12)
        pos = pos + 0.03 * a * rng.standard_normal(pos.shape)
        pos -= np.round(pos / L) * L
        sysm = standin.System(pos, L, dtype=torch.float32, device=cuda)
        sysm.randomize_velocities(kT=1.0, seed=12)
        sim = standin.Simulation(sysm)
        nve = sim.integrate_nve(0.004)
        tfc = htf.tfcompute(LJModel(96))
        cell = sim.nlist_cell(r_buff=0.4, check_period=2, pitch=160)
        tfc.attach(cell, r_cut=2.5)
        sim.run(6)
        assert getattr(sim, "_graph", None) is None  # a short run steps
        sim.run(256, graph=graph)
        first = getattr(sim, "_graph", None)
        if graph is None:
            c = sim.graph_choice
            assert c["use_graph"] and c["graph_us"] < c["stepwise_us"], c
        nve.dt = 0.002
        sim.run(256, graph=graph)
        torch.cuda.synchronize()
        cell._poll_overflow()
        return sysm.pos.clone(), sysm.vel.clone(), first, getattr(sim, "_graph", None), sysm.timestep

    p0, v0, a0, b0, t0 = run(False)
    p1, v1, a1, b1, t1 = run(None)
    assert a0 is None and b0 is None and a1 is not None and b1 is not None and b1 is not a1, "no capture, or no re-capture after dt changed"
    assert t0 == t1 == 518
    assert torch.equal(p0, p1) and torch.equal(v0, v1)


@pytest.mark.parametrize("seed", range(8))
def test_cell_nlist_random_boxes_match_brute_force(htf, cuda, seed):
    """The binned search on random systems: anisotropic boxes (so that some periodic axes have fewer than 7 search
    cells -- rint() images -- and others more -- images from the candidate's cell), random density and list radius,
    both precisions, with and without the type split of a mapped neighbor list.  fp64: the exact brute-force sets;
    fp32: pairs within rounding of r_list may differ, nothing else."""
    from hoomd_tf_amd import standin
    rng = np.random.default_rng(4000 + seed)
    tdt = torch.float64 if seed % 2 else torch.float32
    L = rng.uniform(6.0, 22.0, size=3)
    rho = float(rng.uniform(0.3, 0.9))
    N = int(min(6000, max(64, rho * np.prod(L))))
    pos = (rng.random((N, 3)) - 0.5) * L
    r_cut, r_buff = float(rng.uniform(1.2, 2.6)), float(rng.uniform(0.1, 0.5))
    types = (rng.random(N) < 0.3).astype(np.int32) if seed % 4 >= 2 else None
    sysm = standin.System(pos, L, types=types, dtype=tdt, device=cuda)
    nl = standin.CellNlist(sysm, r_cut=r_cut, r_buff=r_buff)
    nl.type_split = 1 if types is not None else -1
    nl.build()
    p = sysm.pos.cpu().numpy()[:, :3].astype(np.float64)
    bn, bh, bl = brute_nlist(p, L, r_cut + r_buff)
    ref = _rows(bn, bh, bl)
    if types is not None:  # pairs whose types lie on different sides of the split are left out
        ref = [r[types[r] == types[i]] for i, r in enumerate(ref)]
    got = _rows(nl.n_neigh.cpu().numpy(), nl.head_list.cpu().numpy(), nl.nlist.cpu().numpy())
    mism = sum(len(set(g.tolist()) ^ set(r.tolist())) for g, r in zip(got, ref))
    assert mism <= (6 if tdt == torch.float32 else 0), (seed, mism, L, r_cut + r_buff)
    for g in got[:: max(1, N // 50)]:  # rows hold no duplicates and never the particle itself
        assert len(set(g.tolist())) == len(g)
    assert all(i not in set(g.tolist()) for i, g in enumerate(got[:200]))


@pytest.mark.parametrize("kernel", ["HTFS_NLIST_PER_CELL", "HTFS_NLIST_PER_PARTICLE"])
def test_random_boxes_through_either_search_kernel(kernel):
    """The stand-in picks its search kernel by grid size (one wave per cell from 1 024 cells, the per-particle walk below): the
    random boxes above -- both precisions, type splits, axes with fewer than seven cells -- and the lattice cases through EACH
    kernel, forced in a child process on the variants build (the switch exists there only)."""
    import os
    import subprocess
    import sys
    from helpers import variants_env
    env = variants_env(**{kernel: "1"})
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "random_boxes_match_brute_force or cell_nlist_matches_brute_force or small_pitch_regrow"],
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_slab_plan_kernels_match_torch(htf, cuda):
    """The migration / ghost plan of a decomposed rebuild on the device (csrc/standin.hip: slab_classify_kernel,
    htfs_key_sort16, segment_copy_kernel) against plain torch ops on the same data, at a size where one key holds tens of
    thousands of members (the C3 box over two ranks: 65 536 particles per rank, most of them "stay, interior")."""
    import ctypes as C
    from hoomd_tf_amd._lib import lib, check
    g = torch.Generator(device="cuda").manual_seed(4)
    N, world, rank, r_ghost = 70001, 4, 1, 3.4
    L = 53.75
    bounds = np.linspace(-L / 2, L / 2, world + 1)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for dtype, code in ((torch.float32, 0), (torch.float64, 1)):
        pos = torch.zeros((N, 4), dtype=dtype, device=cuda)
        # mostly inside slab 1, a few hundred just across either face, a handful far away (destination 3)
        x = bounds[1] + (bounds[2] - bounds[1]) * torch.rand(N, generator=g, device=cuda, dtype=torch.float64)
        x[:300] = bounds[1] - 0.3 * torch.rand(300, generator=g, device=cuda, dtype=torch.float64)
        x[300:700] = bounds[2] + 0.3 * torch.rand(400, generator=g, device=cuda, dtype=torch.float64)
        x[700:705] = bounds[3] + 1.0
        x = x[torch.randperm(N, generator=g, device=cuda)]
        pos[:, 0] = x.to(dtype)
        key = torch.empty(N, dtype=torch.int32, device=cuda)
        bnd = torch.as_tensor(bounds, dtype=dtype, device=cuda)
        check(lib.htfs_slab_classify(pos.data_ptr(), code, N, bnd.data_ptr(), world, rank, r_ghost, key.data_ptr(), stream))
        xs = pos[:, 0].contiguous()
        owner = torch.bucketize(xs, bnd[1:-1].contiguous(), right=True)
        dest = (owner == rank - 1).long() + 2 * (owner == rank + 1).long() + 3 * ((owner != rank) & (owner != rank - 1) & (owner != rank + 1)).long()
        near_l, near_r = xs < bnd[owner] + r_ghost, xs >= bnd[owner + 1] - r_ghost
        cls = torch.where(near_l, torch.where(near_r, 2, 1), torch.where(near_r, 3, 0))
        ref_key = dest * 4 + cls
        assert torch.equal(key.long(), ref_key)
        assert int((ref_key // 4 == 3).sum()) == 5 and int((ref_key == 0).sum()) > 30000
        scratch = torch.zeros(16 * ((N + 4095) // 4096), dtype=torch.int32, device=cuda)
        start = torch.empty(17, dtype=torch.int32, device=cuda)
        order = torch.full((N,), -1, dtype=torch.int32, device=cuda)
        check(lib.htfs_key_sort16(key.data_ptr(), N, scratch.data_ptr(), start.data_ptr(), order.data_ptr(), stream))
        assert torch.equal(order.long(), torch.sort(ref_key, stable=True)[1])
        cnt = torch.zeros(16, dtype=torch.int64, device=cuda).index_add_(0, ref_key, torch.ones_like(ref_key))
        assert torch.equal((start[1:] - start[:-1]).long(), cnt) and int(start[0]) == 0
    # segment copy: three class-sorted segments merged by class
    rng = np.random.default_rng(2)
    segs = [rng.integers(0, 900, size=4) for _ in range(3)]
    segs[1][2] = 0
    n = int(sum(int(v.sum()) for v in segs))
    src = torch.arange(n * 8, dtype=torch.float32, device=cuda).reshape(n, 8)
    cls_all = torch.repeat_interleave(torch.arange(4, device=cuda).repeat(3), torch.as_tensor(np.concatenate(segs), device=cuda))
    want = src.index_select(0, torch.sort(cls_all, stable=True)[1])
    from hoomd_tf_amd.domain import SlabDomain
    got = SlabDomain._merge_segments_device(None, src, segs[0], segs[1], segs[2])
    assert torch.equal(got, want)


def test_rebuild_on_a_non_periodic_grid_takes_the_nearest_image(htf, cuda):
    """htfs_rebuild_nlist_ghosts' image_L: on a cell grid that is NOT periodic along x while the coordinates are (a decomposed system's
    brick + ghost layer), rows whose x is a whole period away -- wrapped to the far side of the logical box by the integrator after
    they left through a face on its boundary, or the ghosts of such rows -- are binned and searched as their image next to the
    grid: the neighbor rows of EVERY particle equal those of the same configuration with nothing shifted."""
    from hoomd_tf_amd import standin
    rng = np.random.default_rng(9)
    L = np.array([14.0, 12.0, 12.0])
    pos, _, a = standin.fcc_positions(7, 0.8442)
    pos = pos[(np.abs(pos) < L / 2 - 0.05).all(axis=1)]
    pos = pos + 0.03 * a * rng.standard_normal(pos.shape)
    period = 40.0                                      # the logical box along x; the grid spans 14 of it

    class Domain:                                      # what CellNlist asks a fixed-capacity domain for
        fixed_capacity, local_grid, world, replica, n_global = True, True, 1, False, len(pos)
        def rebuild(self): pass
        def nlist_box(self): return np.array([-L / 2, L / 2, [0.0, 0.0, 0.0]]), (0, 1, 1)
        def image_lengths(self): return (period, 0.0, 0.0)
        def attach_n_neigh(self, t): pass

    sysm = standin.System(pos, np.array([period, L[1], L[2]]), dtype=torch.float32, device=cuda)
    nl = standin.CellNlist(sysm, r_cut=2.5, r_buff=0.4, pitch=128)
    nl.domain = Domain()
    nl.build()                                         # (the sizing build: separate calls, everything where it belongs)
    nl.build()                                         # the fixed-capacity fast path on the same positions
    torch.cuda.synchronize()

    def rows():
        n = nl.n_neigh.cpu().numpy()
        t = nl.nlist.cpu().numpy().reshape(len(n), -1)
        return [sorted(t[i, :n[i]].tolist()) for i in range(len(n))]

    want = rows()
    assert sum(len(r) for r in want) > 30 * len(pos)
    moved = rng.random(len(pos)) < 0.3
    sysm.pos[torch.from_numpy(moved).to(cuda), 0] += torch.from_numpy(np.where(rng.random(moved.sum()) < 0.5, period, -period)).float().to(cuda)
    nl.build()
    torch.cuda.synchronize()
    assert rows() == want
    # and without the period the same build loses them (the shifted rows clamp into the edge cells): the parameter is what does it
    Domain.image_lengths = lambda self: (0.0, 0.0, 0.0)
    nl.build()
    torch.cuda.synchronize()
    assert rows() != want


@pytest.mark.parametrize("pot", ["lj", "wca"])
@pytest.mark.parametrize("cells,period", [(10, 4), (18, 5), (24, 5)])
@pytest.mark.parametrize("tdt", [torch.float32, torch.float64])
def test_fused_step_equals_force_kernel_plus_integrator(htf, cuda, tdt, cells, period, pot):
    """Round 6: the stand-in integrator as the EPILOGUE of the one-kernel force step (standin.FusedStep, include/htf_standin.h
    htfs_step_epilogue) -- a finished row's lanes go on with v += f dt and x(t + dt) = wrap(x + v dt) into the OTHER position
    array -- against the force launch + htfs_nve_step launch it replaces: positions, velocities and forces bit for bit over 80
    steps with device-decided rebuilds, 4 000 (the plain two-row form), 23 328 (two rows with merged tails) and 55 296 rows (four
    rows), an even check period (every step fused) and an odd one (the period's last step classic: it must end on its home array)."""
    from hoomd_tf_amd import standin
    pos, L, a = standin.fcc_positions(cells, 0.8442)
    rng = np.random.default_rng(cells)
    pos = pos + 0.03 * a * rng.standard_normal(pos.shape)
    pos -= np.round(pos / L) * L
    out = {}
    for mode in ("classic", "fused"):
        sysm = standin.System(pos, L, dtype=tdt, device=cuda)
        sysm.randomize_velocities(kT=1.2, seed=4)
        nl = standin.CellNlist(sysm, r_cut=2.5, r_buff=0.4, check_period=period, device_decision=True)
        nl.build()
        ctx = htf.Context(r_cut=2.5, nneighs=80, scalar_dtype=tdt, max_n=sysm.N, fused=2)
        ctx.set_potential(htf.Potential.lj() if pot == "lj" else htf.Potential.wca(1.0))
        nve = standin.NVE(sysm, 0.004)
        fs = standin.FusedStep(sysm, nl, ctx, nve)
        # (fp64 positions -- a HOOMD DOUBLE build -- keep the integrator's own launch: the epilogue measured slower there, twice -- 9.7 k
        #  against 11.1 k steps/s at C3 in its first form, 10.6 k against 10.9 k in the one-copy form -- and is not compiled;
        #  FusedStep then IS the classic pair of launches)
        assert fs.available == (tdt == torch.float32)
        home = sysm.pos.data_ptr()
        swaps = 0
        for ts in range(80):
            if mode == "fused":
                before = sysm.pos.data_ptr()
                fs.step(ts)
                swaps += int(sysm.pos.data_ptr() != before)
                if ts % period == period - 1 and fs.available:
                    assert sysm.pos.data_ptr() == home
            else:
                nl.compute(ts)
                ctx.compute_forces(ts, fs.arrays())
                nve.step()
        torch.cuda.synchronize()
        assert nl.device_builds() >= 2
        if mode == "fused":
            assert swaps == ((80 if period % 2 == 0 else 80 - 80 // period) if fs.available else 0)
        out[mode] = (sysm.pos.clone(), sysm.vel.clone(), sysm.force.clone(), nl.device_builds())
    assert out["classic"][3] == out["fused"][3]
    for k in range(3):
        assert torch.equal(out["classic"][k], out["fused"][k]), k


@pytest.mark.parametrize("lattice,cells,NN,rbuff", [("sc", 27, 40, 0.4), ("sc", 37, 40, 0.4), ("sc", 37, 80, 0.4), ("fcc", 16, 128, 1.5),
                                                    ("fcc", 24, 128, 1.5)])
def test_fused_step_on_the_rows_the_fast_path_hands_back(htf, cuda, lattice, cells, NN, rbuff):
    """The step epilogue is ONE piece of code at the end of a row group; rows the straight-line path does not take hand their sums
    back to it: rows that OVERFLOW NN (NN = 40 at r_cut 2.5: every row of the liquid, redone by the generic routine), lists longer
    than 192 entries (r_buff = 1.5: the whole group falls back), a last group with fewer rows than the form takes (19 683 and 50 653
    particles are odd numbers).  Two-row and four-row merged-tails forms; positions, velocities, forces against force launch +
    htfs_nve_step bit for bit over 40 steps with rebuilds."""
    from hoomd_tf_amd import standin
    pos, L, a = (standin.sc_positions if lattice == "sc" else standin.fcc_positions)(cells, 0.8442)
    rng = np.random.default_rng(cells)
    pos = pos + 0.03 * a * rng.standard_normal(pos.shape)
    pos -= np.round(pos / L) * L
    out = {}
    for mode in ("classic", "fused"):
        sysm = standin.System(pos, L, dtype=torch.float32, device=cuda)
        sysm.randomize_velocities(kT=1.0, seed=4)
        nl = standin.CellNlist(sysm, r_cut=2.5, r_buff=rbuff, check_period=4, device_decision=True)
        nl.build()
        if rbuff > 1.0:
            assert int(nl.n_neigh.max()) > 192
        ctx = htf.Context(r_cut=2.5, nneighs=NN, scalar_dtype=torch.float32, max_n=sysm.N, fused=2)
        ctx.set_potential(htf.Potential.lj())
        nve = standin.NVE(sysm, 0.002)
        fs = standin.FusedStep(sysm, nl, ctx, nve)
        assert fs.available
        for ts in range(40):
            if mode == "fused":
                fs.step(ts)
            else:
                nl.compute(ts)
                ctx.compute_forces(ts, fs.arrays())
                nve.step()
        torch.cuda.synchronize()
        if NN == 40:     # every row holds NN pairs: the overflow replay ran for all of them
            pv = ctx.nlist_buffer(sysm.N, cuda)
            assert int((pv[:, :, :3] != 0).any(dim=2).sum(dim=1).min()) == NN
        out[mode] = (sysm.pos.clone(), sysm.vel.clone(), sysm.force.clone())
    assert bool(torch.isfinite(out["fused"][2]).all())
    for k in range(3):
        assert torch.equal(out["classic"][k], out["fused"][k]), k


def test_fused_step_is_refused_where_the_kernel_cannot_carry_it(htf, cuda):
    """A virial request (the one-row kernel), a pair-MLP (its own evaluator), a generated kernel, batching: FusedStep.available is
    False and step() is the classic pair of launches."""
    from hoomd_tf_amd import standin
    from hoomd_tf_amd.initializers import mlp_params
    pos, L, a = standin.fcc_positions(6, 0.8442)
    for kw, potf in ((dict(virial=True), lambda: htf.Potential.lj()), (dict(batch_size=256), lambda: htf.Potential.lj()),
                     (dict(fused=0), lambda: htf.Potential.lj()),
                     (dict(), lambda: htf.Potential.pair_mlp(mlp_params(seed=3), 0.0, 3.0, activation="tanh", precision="split16"))):
        sysm = standin.System(pos, L, dtype=torch.float32, device=cuda)
        sysm.randomize_velocities(kT=1.0, seed=1)
        nl = standin.CellNlist(sysm, r_cut=2.5, r_buff=0.4)
        nl.build()
        ctx = htf.Context(r_cut=2.5, nneighs=80, max_n=sysm.N, **dict(dict(fused=2), **kw))
        ctx.set_potential(potf())
        fs = standin.FusedStep(sysm, nl, ctx, standin.NVE(sysm, 0.004))
        assert not fs.available, kw
        p0 = sysm.pos.clone()
        fs.step(0)
        torch.cuda.synchronize()
        assert not torch.equal(p0, sysm.pos) and bool(torch.isfinite(sysm.pos).all())
