"""Domain-decomposed MD on the GPU kernels: 2 ranks share the one GPU of the test box (gloo
transport with host bounce buffers -- RCCL refuses two ranks on one device), run LJ MD
with migration, ghost halo and the interior/boundary split of htf_compute_forces_rows, and
every rank's forces must equal the single-domain forces of the gathered configuration (the
reference's MPI assertion, test_mpi_tensorflow.py:57-79, on the HIP path)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, per_slab=6):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        sys.path.insert(0, ROOT)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import hoomd_tf_amd as htf
        from hoomd_tf_amd import standin
        from hoomd_tf_amd.domain import SlabDomain

        dev = torch.device("cuda:0")
        rcut, rbuf, NN = 2.5, 0.4, 80
        # slabs of 6 fcc cells (10 sigma >= 2 r_ghost) side by side; per_slab = 2: slabs 3.36 wide, r_ghost 2.9 --
        # thinner than 2 r_ghost, every particle a ghost somewhere (the 131072-particle box over 8 ranks)
        cells = (per_slab * world, 6, 6)
        a = (4.0 / 0.8442) ** (1.0 / 3.0)
        base = np.array([[0, 0, 0], [.5, .5, 0], [.5, 0, .5], [0, .5, .5]])
        grid = np.stack(np.meshgrid(*[np.arange(c) for c in cells], indexing="ij"), -1).reshape(-1, 3)
        pos = ((grid[:, None, :] + base[None]) * a).reshape(-1, 3)
        L = np.array(cells, dtype=np.float64) * a
        pos = pos - L / 2
        rng = np.random.default_rng(5)
        pos += 0.05 * a * rng.standard_normal(pos.shape)
        pos -= np.round(pos / L) * L
        Ng = len(pos)
        ids = np.arange(Ng)
        vel = np.zeros((Ng, 4))
        vel[:, :3] = 1.5 * rng.standard_normal((Ng, 3))
        vel[:, 3] = 1.0
        bounds = -L[0] / 2 + np.linspace(0, 1, world + 1) * L[0]
        mine = (pos[:, 0] >= bounds[rank]) & (pos[:, 0] < bounds[rank + 1])
        sysm = standin.System(pos[mine], L, types=ids[mine], dtype=torch.float32, device=dev)
        sysm.vel = torch.from_numpy(vel[mine]).to(torch.float32).to(dev)
        nl = standin.CellNlist(sysm, r_cut=rcut, r_buff=rbuf, check_period=1)
        nl.domain = SlabDomain(sysm, rank, world, r_ghost=rcut + rbuf)
        nl.build()
        ctx = htf.Context(r_cut=rcut, nneighs=NN, scalar_dtype=torch.float32, max_n=sysm.N)
        pot = htf.Potential.lj()
        ctx.set_potential(pot)
        nve = standin.NVE(sysm, 0.004)
        builds, arr = -1, None
        overlapped = 0
        for ts in range(80):
            nl.compute(ts)
            if nl.n_builds != builds:
                arr = ctx.make_arrays(sysm.pos, sysm.N, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, sysm.force)
                builds = nl.n_builds
            overlapped += int(nl.domain.pending)
            ctx.compute_forces_overlapped(ts, arr, nl.domain)
            if ts < 79:
                nve.step()
        torch.cuda.synchronize()
        assert nl.n_builds >= 2 and overlapped >= 20, (nl.n_builds, overlapped)
        assert (nl.domain.n_interior == 0) == (per_slab == 2)
        # gather the configuration by particle id
        N = sysm.N
        my_ids = sysm.types_numpy()
        loc = torch.zeros((Ng, 3), dtype=torch.float64)
        loc[my_ids] = sysm.pos[:N, :3].double().cpu()
        owned = torch.zeros(Ng, dtype=torch.float64)
        owned[my_ids] = 1
        dist.all_reduce(loc)
        dist.all_reduce(owned)
        assert bool((owned == 1).all()), "particles lost or duplicated"
        # single-domain forces of that configuration on the same kernels
        ref_sys = standin.System(loc.numpy(), L, dtype=torch.float32, device=dev)
        ref_nl = standin.CellNlist(ref_sys, r_cut=rcut, r_buff=rbuf)
        ref_nl.build()
        ref_ctx = htf.Context(r_cut=rcut, nneighs=NN, scalar_dtype=torch.float32, max_n=Ng)
        ref_ctx.set_potential(pot)
        ref_arr = ref_ctx.make_arrays(ref_sys.pos, Ng, ref_nl.n_neigh, ref_nl.head_list, ref_nl.nlist, ref_sys.box, ref_sys.force)
        ref_ctx.compute_forces(0, ref_arr)
        torch.cuda.synchronize()
        want = ref_sys.force.cpu().numpy()[my_ids]
        got = sysm.force[:N].cpu().numpy()
        # same pair set, different summation order (neighbor order differs between the two lists)
        scale = np.abs(want).max()
        assert np.abs(got - want).max() < 2e-5 * scale, (np.abs(got - want).max(), scale)
        moved = torch.tensor([nl.domain.n_migrated])
        dist.all_reduce(moved)
        assert int(moved) > 0, "test must exercise migration"
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


def _train_worker(rank, world, port, q, kind="slab"):
    """tfcompute-driven online force matching under slabs: LJ drives the MD (traced one-kernel
    path, halo overlapped), a pair-MLP is trained on it every step; the [loss, gradient, count]
    all-reduce must keep every rank's weights bit-identical and equal to single-domain training."""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        import hoomd_tf_amd as htf
        from hoomd_tf_amd import standin
        from hoomd_tf_amd.domain import SlabDomain
        import build_examples

        dev = torch.device("cuda:0")
        rcut, NN = 2.5, 64
        cells = (12, 5, 5)
        a = (4.0 / 0.8442) ** (1.0 / 3.0)
        base = np.array([[0, 0, 0], [.5, .5, 0], [.5, 0, .5], [0, .5, .5]])
        grid = np.stack(np.meshgrid(*[np.arange(c) for c in cells], indexing="ij"), -1).reshape(-1, 3)
        pos = ((grid[:, None, :] + base[None]) * a).reshape(-1, 3)
        L = np.array(cells, dtype=np.float64) * a
        pos = pos - L / 2
        rng = np.random.default_rng(6)
        pos += 0.04 * a * rng.standard_normal(pos.shape)
        pos -= np.round(pos / L) * L
        Ng = len(pos)
        vel = np.zeros((Ng, 4))
        vel[:, :3] = 0.8 * rng.standard_normal((Ng, 3))
        vel[:, 3] = 1.0
        bounds = -L[0] / 2 + np.linspace(0, 1, world + 1) * L[0]
        mine = (pos[:, 0] >= bounds[rank]) & (pos[:, 0] < bounds[rank + 1])
        system = standin.System(pos[mine], L, types=np.arange(Ng)[mine], dtype=torch.float32, device=dev)
        system.vel = torch.from_numpy(vel[mine]).to(torch.float32).to(dev)
        sim = standin.Simulation(system)
        sim.integrate_nve(0.002)
        nlist = sim.nlist_cell(check_period=1)
        if world > 1 and kind == "brick":
            from hoomd_tf_amd.brick import BrickDomain
            nlist.domain = BrickDomain(system, rank, (world, 1, 1), r_ghost=rcut + nlist.r_buff, r_buff=nlist.r_buff, n_global=Ng)
        elif world > 1:
            nlist.domain = SlabDomain(system, rank, world, r_ghost=rcut + nlist.r_buff)
        lj = htf.tfcompute(build_examples.LJModel(NN))
        lj.attach(nlist, r_cut=rcut)
        model = build_examples.PairMLPModel(NN, output_forces=False, activation='tanh', seed=9)
        model.compile(htf.optimizers.Adam(0.003), loss='MeanSquaredError')
        tfc = htf.tfcompute(model)
        tfc.attach(nlist, train=True, r_cut=rcut)
        tfc.set_reference_forces(lj)
        sim.run(12)
        torch.cuda.synchronize()
        w = np.concatenate([x.ravel() for x in model.mlp.get_weights()])
        loss = float(tfc._opt_state[20])
        assert np.all(np.isfinite(w)) and np.isfinite(loss)
        if world > 1:
            both = [torch.zeros(len(w), dtype=torch.float64) for _ in range(world)]
            dist.all_gather(both, torch.from_numpy(w.astype(np.float64)))
            for o in both[1:]:
                assert torch.equal(o, both[0]), "ranks hold different weights after training"
            n_all = torch.tensor([nlist.domain.n_local if kind == "brick" else system.N])
            dist.all_reduce(n_all)
            assert int(n_all) == Ng
            dist.barrier()
            dist.destroy_process_group()
        q.put((rank, world, w, loss))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, world, None, traceback.format_exc()))


@pytest.mark.parametrize("kind", ["slab", "brick"])
def test_training_under_slabs_matches_single_domain(htf, cuda, kind):
    """Online pair-MLP force matching under two slabs -- SlabDomain, and BrickDomain's fixed-capacity arrays (inert rows: zero
    prediction against a zero label, no gradient; the global batch is the conserved particle count) -- == single-domain training."""
    ctx = mp.get_context("spawn")
    out = {}
    for world in (1, 2):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_train_worker, args=(r, world, port, q, kind)) for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=600) for _ in procs]
        for p in procs:
            p.join(timeout=60)
        for rank, wd, w, loss in res:
            assert w is not None, "rank %d of %d failed:\n%s" % (rank, wd, loss)
        out[world] = res[0]
    w1, w2 = out[1][2], out[2][2]
    assert np.abs(w1).max() > 0 and np.abs(w2 - w1).max() < 2e-4 * np.abs(w1).max(), np.abs(w2 - w1).max()
    assert abs(out[2][3] - out[1][3]) < 2e-3 * abs(out[1][3])


def _headline_worker(rank, world, port, q, steps=40):
    """The strong-scaling geometry of the metric's own box: the C3 fcc box (131 072 particles, L = 53.75, r_cut 3.0 + r_buff 0.4,
    NN 128) cut into ``world`` slabs -- 65 536 rows + ~16.6 k ghost rows per rank at world 2, the ranks sharing the one GPU -- run
    for ``steps`` MD steps (migration across both faces, >= 1 rebuild), then forces of 512 sampled rows per rank against
    O.compute_forces of the UNDIVIDED box (fp64 oracle on the gathered fp32 positions; the row's neighbors from a brute-force
    list over all 131 072 particles, independent of the stand-in's cell list)."""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import hoomd_tf_amd as htf
        from hoomd_tf_amd import standin
        from hoomd_tf_amd.domain import SlabDomain
        from oracle import htf_oracle as O
        from test_gpu_parity import LIQUID, _cond_scale, _pair_forces_lj, assert_forces_close

        dev = torch.device("cuda:0")
        rcut, rbuf, NN = 3.0, 0.4, 128
        # the bench's own kind of configuration: rank 0 relaxes the jittered lattice into a liquid at kT = 1 (force cap + velocity
        # rescale, as bench.py's preparation; a jittered 131 072-particle lattice holds pairs at r ~ 0.6 that would blow up plain
        # NVE) and hands every rank the same positions and velocities
        pos, L, a = standin.fcc_positions(32, 0.8442)
        Ng = len(pos)
        assert Ng == 131072
        state = torch.zeros((Ng, 6), dtype=torch.float64)
        if rank == 0:
            from test_gpu_parity import _liquid
            lsys, _, _ = _liquid(htf, dev, cells=32, steps=150, seed=3)
            state[:, :3] = lsys.pos[:Ng, :3].double().cpu()
            state[:, 3:] = lsys.vel[:Ng, :3].double().cpu()
            del lsys
        dist.broadcast(state, src=0)
        pos = state[:, :3].numpy().copy()
        ids = np.arange(Ng)
        vel = np.zeros((Ng, 4))
        vel[:, :3] = state[:, 3:].numpy()
        vel[:, 3] = 1.0
        Lv = np.array([L, L, L], dtype=np.float64) if np.ndim(L) == 0 else np.asarray(L, dtype=np.float64)
        bounds = -Lv[0] / 2 + np.linspace(0, 1, world + 1) * Lv[0]
        mine = (pos[:, 0] >= bounds[rank]) & (pos[:, 0] < bounds[rank + 1])
        sysm = standin.System(pos[mine], L, types=ids[mine], dtype=torch.float32, device=dev)
        sysm.vel = torch.from_numpy(vel[mine]).to(torch.float32).to(dev)
        nl = standin.CellNlist(sysm, r_cut=rcut, r_buff=rbuf, check_period=1)
        nl.domain = SlabDomain(sysm, rank, world, r_ghost=rcut + rbuf)
        nl.build()
        if world == 2:
            assert 60000 < sysm.N < 71000 and 15000 < sysm.n_ghost < 18500, (sysm.N, sysm.n_ghost)
            assert nl.domain.n_interior > 0.6 * sysm.N                      # slabs 26.9 thick: 0.75 of the rows see no ghost
        ctx = htf.Context(r_cut=rcut, nneighs=NN, scalar_dtype=torch.float32, max_n=sysm.N)
        ctx.set_potential(htf.Potential.lj())
        nve = standin.NVE(sysm, 0.005)
        builds, arr, overlapped = -1, None, 0
        for ts in range(steps):
            nl.compute(ts)
            if nl.n_builds != builds:
                arr = ctx.make_arrays(sysm.pos, sysm.N, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, sysm.force)
                builds = nl.n_builds
            overlapped += int(nl.domain.pending)
            ctx.compute_forces_overlapped(ts, arr, nl.domain)
            if ts < steps - 1:
                nve.step()
        torch.cuda.synchronize()
        assert nl.n_builds >= 2 and overlapped >= steps // 2, (nl.n_builds, overlapped)
        N = sysm.N
        my_ids = sysm.types_numpy()
        loc = torch.zeros((Ng, 3), dtype=torch.float64)
        loc[my_ids] = sysm.pos[:N, :3].double().cpu()
        owned = torch.zeros(Ng, dtype=torch.float64)
        owned[my_ids] = 1
        dist.all_reduce(loc)
        dist.all_reduce(owned)
        assert bool((owned == 1).all()), "particles lost or duplicated"
        moved = torch.tensor([nl.domain.n_migrated])
        dist.all_reduce(moved)
        assert int(moved) > 0, "test must exercise migration"
        # the oracle on the undivided box, for 512 of this rank's rows
        allpos = loc.numpy().astype(np.float32)
        pick = np.random.default_rng(100 + rank).choice(N, 512, replace=False)
        gid = my_ids[pick]
        rows = []
        for c in range(0, 512, 64):                                        # brute force over all particles, 64 rows at a time
            d = allpos[None, :, :].astype(np.float64) - allpos[gid[c:c + 64], None, :].astype(np.float64)
            d -= np.round(d / Lv) * Lv
            m = (d * d).sum(axis=2) <= (rcut + 0.05) ** 2
            m[np.arange(len(gid[c:c + 64])), gid[c:c + 64]] = False
            rows += [np.nonzero(r)[0] for r in m]
        n_neigh = np.array([len(r) for r in rows], dtype=np.uint32)
        head = np.concatenate([[0], np.cumsum(n_neigh)[:-1]]).astype(np.uint32)
        flat = np.concatenate(rows)
        # O.compute_forces evaluates rows [0, n_local): put the sampled particles first
        order = np.concatenate([gid, np.setdiff1d(np.arange(Ng), gid)])
        new_of_old = np.empty(Ng, dtype=np.int64)
        new_of_old[order] = np.arange(Ng)
        ppos = allpos[order]
        captured = {}

        def model(x):
            captured["pv"] = x.astype(np.float64)
            return O.lj_model(captured["pv"])

        ref, _ = O.compute_forces(ppos, np.zeros(Ng, np.int32), n_neigh, head, new_of_old[flat].astype(np.uint32),
                                  O.make_box(L, dtype=np.float32), rcut, NN, model, model_dtype=np.float32, n_local=512)
        pv64 = captured["pv"]
        assert int((np.abs(pv64[:, :, :3]).sum(axis=2) > 0).sum(axis=1).max()) < NN
        got = sysm.force[:N].cpu().numpy()[pick]
        cond = _cond_scale(pv64, _pair_forces_lj(pv64))
        tag = "slabs%d_c3_geometry_rank%d" % (world, rank)
        assert_forces_close(tag + "_energy", got[:, 3], ref[:, 3])
        assert_forces_close(tag, got[:, :3], ref[:, :3], cond, cancelling_rows=LIQUID)
        import json
        from test_gpu_parity import STATS
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok", {k: v for k, v in STATS.items() if k.startswith(tag)}))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc(), {}))


def test_two_slabs_at_the_headline_geometry(htf, cuda):
    """the decomposed step at the geometry `bench.py --gpus 2` runs (test_mpi_tensorflow.py:57-79 at the
    metric's own size) -- 2 x 65 536 rows of the C3 box + 16.6 k ghosts per rank -- against the oracle of the undivided box."""
    from test_gpu_parity import _record
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_headline_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg, stats in results:
        assert msg == "ok", "rank %d failed:\n%s" % (rank, msg)
        for k, v in stats.items():
            _record(k, **v)


@pytest.mark.parametrize("world,per_slab", [(2, 6), (3, 6), (5, 2)])
def test_slabs_on_one_gpu(htf, cuda, world, per_slab):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, per_slab)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", "rank %d failed:\n%s" % (rank, msg)


def _deferred_worker(rank, world, port, q, per_slab):
    """Two runs of the same decomposed LJ MD: the rebuild decision (standin.DeferredRebuildRule) fed WITHOUT a host
    synchronisation -- device all-reduce, pinned copy, read one check later -- and fed from a blocking read at every check
    (the host-decided twin).  Same rule, same one-check lag: the two must rebuild at the same steps and end in bit-identical
    positions on every rank; no build may be dangerous; forces of the final configuration are whole (nobody lost)."""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        sys.path.insert(0, ROOT)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import hoomd_tf_amd as htf
        from hoomd_tf_amd import standin
        from hoomd_tf_amd.domain import SlabDomain

        dev = torch.device("cuda:0")
        rcut, rbuf, NN = 2.5, 0.4, 80
        cells = (per_slab * world, 5, 5)
        a = (4.0 / 0.8442) ** (1.0 / 3.0)
        base = np.array([[0, 0, 0], [.5, .5, 0], [.5, 0, .5], [0, .5, .5]])
        grid = np.stack(np.meshgrid(*[np.arange(c) for c in cells], indexing="ij"), -1).reshape(-1, 3)
        pos0 = ((grid[:, None, :] + base[None]) * a).reshape(-1, 3)
        L = np.array(cells, dtype=np.float64) * a
        pos0 = pos0 - L / 2
        rng = np.random.default_rng(11)
        pos0 += 0.04 * a * rng.standard_normal(pos0.shape)
        pos0 -= np.round(pos0 / L) * L
        Ng = len(pos0)
        vel0 = np.zeros((Ng, 4))
        vel0[:, :3] = 1.0 * rng.standard_normal((Ng, 3))
        bounds = -L[0] / 2 + np.linspace(0, 1, world + 1) * L[0]
        mine = (pos0[:, 0] >= bounds[rank]) & (pos0[:, 0] < bounds[rank + 1])
        out = {}
        for mode in ("device", "host", "torch-plan"):
            # "torch-plan": the migration / ghost plan of a rebuild through the torch restatement (HTF_DOMAIN_TORCH=1: the
            # path the CPU gloo tests exercise) instead of the classification + counting-sort + segment-copy kernels
            os.environ["HTF_DOMAIN_TORCH"] = "1" if mode == "torch-plan" else "0"
            sysm = standin.System(pos0[mine], L, types=np.arange(Ng)[mine], dtype=torch.float32, device=dev)
            sysm.vel = torch.from_numpy(vel0[mine]).to(torch.float32).to(dev)
            nl = standin.CellNlist(sysm, r_cut=rcut, r_buff=rbuf, check_period=3, device_decision=(mode == "device"),
                                   deferred_reference=(mode != "device"))
            nl.domain = SlabDomain(sysm, rank, world, r_ghost=rcut + rbuf)
            nl.build()
            ctx = htf.Context(r_cut=rcut, nneighs=NN, scalar_dtype=torch.float32, max_n=sysm.N)
            ctx.set_potential(htf.Potential.lj())
            nve = standin.NVE(sysm, 0.004)
            builds, arr, steps_built = -1, None, []
            for ts in range(90):
                nl.compute(ts)
                if nl.n_builds != builds:
                    arr = ctx.make_arrays(sysm.pos, sysm.N, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, sysm.force)
                    builds = nl.n_builds
                    steps_built.append(ts)
                ctx.compute_forces_overlapped(ts, arr, nl.domain)
                nve.step()
            torch.cuda.synchronize()
            assert nl._rule is not None and nl.dangerous_builds == 0, nl.dangerous_builds
            assert len(steps_built) >= 4, steps_built
            out[mode] = (steps_built, sysm.types_numpy().copy(), sysm.pos[:sysm.N].cpu().numpy().copy(),
                         sysm.force[:sysm.N].cpu().numpy().copy())
            tot = torch.tensor([sysm.N])
            dist.all_reduce(tot)
            assert int(tot) == Ng
        for other in ("host", "torch-plan"):
            assert out["device"][0] == out[other][0], (other, out["device"][0], out[other][0])  # rebuilt at the same steps
            np.testing.assert_array_equal(out["device"][1], out[other][1])                      # same particles, same order
            np.testing.assert_array_equal(out["device"][2], out[other][2])                      # bit-identical trajectories
            np.testing.assert_array_equal(out["device"][3], out[other][3])
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("world,per_slab", [(3, 4), (8, 2)])
def test_deferred_device_decision_equals_host_decided(htf, cuda, world, per_slab):
    """under decomposition the step loop has no read-back -- the distance check is all-reduced on the
    device and read one check late -- and the trajectory is the one the host-decided twin produces, at world 3 and at
    world 8 with slabs thinner than 2 r_ghost (the ranks share the one GPU of the test box over gloo)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_deferred_worker, args=(r, world, port, q, per_slab)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", "rank %d failed:\n%s" % (rank, msg)


@pytest.mark.parametrize("args", [["--gpus", "2"], ["--gpus", "3"], ["--gpus", "2", "--scaling", "weak"], ["--gpus", "8", "--grid", "4x2x1"],
                                  ["--gpus", "4", "--grid", "2x2x1"]])
def test_bench_starts_its_own_ranks(htf, cuda, args):
    """`python bench.py --gpus N` with no launcher around it (how the driver calls it): the parent starts the N rank
    processes itself and relays ONE JSON line.  Rehearsed here with the ranks sharing the one GPU (gloo)."""
    import json
    import subprocess
    env = dict(os.environ, HTF_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    # (the ranks share ONE GPU here: up to eight processes time-sliced, the library-free transport's bounded waits in flight -- a
    #  rehearsal now and then runs one of them out, and the guarded section then reports a named skip, as designed; the asserts
    #  below want to see the transport run, so such a line gets ONE more attempt)
    env.setdefault("HTF_PEER_SPIN", str(1 << 25))
    env.setdefault("HTF_BRICK_WAIT_S", "120")
    for attempt in range(2):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args + ["--cells", "12", "--steps", "10", "--warmup", "3", "--equil", "40",
                                                                                      "--no-cpu-baseline", "--no-fused", "--windows", "1"],
                           capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(lines) == 1, r.stdout
        d = json.loads(lines[0])
        skipped = " ".join(str(d.get(k, {}).get("skipped", "")) for k in ("peer_selftest", "graph_variant_peer")) + str(d.get("guarded_section_ended_by", ""))
        if attempt == 0 and any(w in skipped for w in ("did not arrive", "never reached cycle", "watchdog", "timed out", "another rank failed")):
            continue
        break
    n = int(args[1])
    weak = "weak" in args
    assert d["n_gpus"] == n and d["scaling"] == ("weak" if weak else "strong")
    assert d["config"]["global_particles"] == 4 * 12 ** 3 * (n if weak else 1)
    assert d["value"] > 0 and abs(d["value"] * d["config"]["global_particles"] - d["particle_steps_per_s"]) < 1e-6 * d["particle_steps_per_s"]
    assert -7.0 < d["energy_per_particle"] < -4.0 and 0.5 < d["kT_final"] < 1.5   # still the same liquid
    assert d["config"]["halo"]["ghosts_rank0"] > 0 and "BrickDomain" in d["config"]["halo"]["domain"]
    assert d["config"]["parallelism"] == ("dd" + args[3] if "--grid" in args else "dd%sx1x1" % args[1])
    # who ran it, the MLP box beside the LJ box, and the guarded section's fields (a record or a NAMED skip)
    assert len(d["ranks"]) == n and [r["rank"] for r in d["ranks"]] == list(range(n))
    assert all(r["device_uuid"] and r["process_group"]["world_size"] == n for r in d["ranks"])
    if not weak:
        m = d["mlp"]
        assert m["n_gpus"] == n and m["value"] > 0 and m["roofline"]["bound"] == "mfma" and 0.5 < m["kT_final"] < 1.5
        assert m["config"]["parallelism"] == d["config"]["parallelism"] and m["config"]["halo"]["ghosts_rank0"] > 0
    for key in ("native_selftest", "graph_variant", "peer_selftest", "graph_variant_peer"):
        assert key in d and ("value" in d[key] or "exchanges" in d[key] or d[key].get("skipped")), (key, d.get(key))
    # ranks sharing one GPU: RCCL refuses (a named skip); the library-free transport runs for real, self-test and replay
    assert "share" in d["native_selftest"]["skipped"] and d["graph_variant"]["skipped"]
    assert d["peer_selftest"]["bit_equal_to_torch_transport"] and "fine-grained" in d["peer_selftest"]["inbox_memory"]
    # `value` is the fastest path that ran to the end and passed its checks; the eager figure stays on the line
    assert "value_path" in d
    if "eager" in d:
        assert d["value"] >= d["eager"]["value"] and d["value"] in (d["graph_variant"].get("value"), d["graph_variant_peer"].get("value"))
    g = d["graph_variant_peer"]
    assert g["value"] > 0 and g["halo"]["transport"] == "peer" and g["particles"] == d["config"]["global_particles"]
    assert -7.0 < g["energy_per_particle"] < -4.0 and 0.5 < g["kT"] < 1.5


def _self_exchange():
    """Body of test_native_halo_single_rank_self_exchange (run in a child process: a transport that hung would
    otherwise take the whole test session with it)."""
    import ctypes as C
    sys.path.insert(0, ROOT)
    import hoomd_tf_amd  # noqa: F401
    from hoomd_tf_amd._lib import lib, check
    cuda = torch.device("cuda:0")
    if not lib.htf_halo_available():
        print("SKIP librccl not loadable")
        return
    ident = (C.c_char * 128)()
    check(lib.htf_halo_unique_id(ident))
    h = C.c_void_p()
    check(lib.htf_halo_create(ident, 0, 1, C.byref(h)))
    for tdt, code in ((torch.float32, 0), (torch.float64, 1)):
        N, nl_, nr_ = 1000, 37, 53
        pos = torch.zeros((N + nl_ + nr_, 4), dtype=tdt, device=cuda)
        pos[:N] = torch.arange(N * 4, dtype=tdt, device=cuda).reshape(N, 4)
        send_left, send_right = (N - nl_ - nr_, N - nr_), (N - nr_, N)
        # ghosts: [from left (= my right-going message) | from right (= my left-going message)]
        recv_left, recv_right = (N, N + nr_), (N + nr_, N + nr_ + nl_)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for rep in range(3):   # more than once: pending flag, event reuse
            pos[N:] = -1.0
            check(lib.htf_halo_exchange_begin(h, pos.data_ptr(), code, 0, 0, send_left[0], nl_, send_right[0], nr_,
                                              recv_left[0], nr_, recv_right[0], nl_, stream))
            try:  # a second begin before the end is refused
                check(lib.htf_halo_exchange_begin(h, pos.data_ptr(), code, 0, 0, 0, 1, 0, 1, N, 1, N + 1, 1, stream))
                raise AssertionError("second begin accepted")
            except ValueError:
                pass
            check(lib.htf_halo_exchange_end(h, stream))
            torch.cuda.synchronize()
            assert torch.equal(pos[recv_right[0]:recv_right[1]], pos[send_left[0]:send_left[1]])
            assert torch.equal(pos[recv_left[0]:recv_left[1]], pos[send_right[0]:send_right[1]])
    lib.htf_halo_destroy(h)
    print("SELF-EXCHANGE OK")


def test_native_halo_single_rank_self_exchange(htf, cuda):
    """csrc/halo.hip on the real GPU with the one rank a 1-GPU box offers: communicator bring-up from a unique
    id, the grouped ncclSend x2 / ncclRecv x2 (to and from itself: both neighbors of a lone slab are the slab),
    the halo stream and its two events.  The slab's left-going message must land in the 'from right' ghosts and
    the right-going one in the 'from left' ghosts, as with real neighbors."""
    import subprocess
    code = "import sys; sys.path.insert(0, %r); import test_gpu_domain as t; t._self_exchange()" % os.path.join(ROOT, "tests")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240, cwd=ROOT)
    if "SKIP" in r.stdout:
        pytest.skip(r.stdout.strip())
    assert r.returncode == 0 and "SELF-EXCHANGE OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def _rccl_worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        sys.path.insert(0, ROOT)
        torch.cuda.set_device(rank)
        dev = torch.device("cuda", rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        from hoomd_tf_amd import standin
        from hoomd_tf_amd.domain import SlabDomain
        pos, L, a = standin.fcc_positions(12, 0.8442)
        rng = np.random.default_rng(3)
        pos = pos + 0.05 * a * rng.standard_normal(pos.shape)
        pos -= np.round(pos / L) * L
        bounds = -L[0] / 2 + np.linspace(0, 1, world + 1) * L[0]
        mine = (pos[:, 0] >= bounds[rank]) & (pos[:, 0] < bounds[rank + 1])
        out = {}
        for transport in ("torch", "native"):
            sysm = standin.System(pos[mine], L, types=np.arange(len(pos))[mine], dtype=torch.float32, device=dev)
            dom = SlabDomain(sysm, rank, world, r_ghost=3.4, transport=transport)
            dom.rebuild()
            assert dom.transport == transport
            sysm.pos[:sysm.N, :3] += 0.01   # move, then the per-step halo
            dom.exchange_begin()
            dom.exchange_end()
            torch.cuda.synchronize()
            out[transport] = sysm.pos.clone()
        assert torch.equal(out["torch"], out["native"])
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


def test_rccl_halo_two_gpus(htf, cuda):
    """The RCCL transports themselves (torch.distributed's and the native one) between two real devices:
    both must deliver the same ghosts.  Needs a multi-GPU node; skipped on the 1-GPU test box."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL wants one device per rank)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", "rank %d failed:\n%s" % (rank, msg)
