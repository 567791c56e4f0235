"""BrickDomain on CPU (the torch restatement of csrc/brick.hip; gloo ranks): fixed-capacity arrays with inert rows, slabs and
px x py bricks.  Per grid: ownership, nobody lost, ghost == owner position every step, migration across both axes and the
periodic boundary, the [interior | boundary by class] layout, no interior row with a ghost neighbor, per-rank oracle forces over
local + ghost rows == single-domain forces (the reference's MPI assertion, test_mpi_tensorflow.py:57-79, with
``comm.decomposition(nx=..., ny=...)``); for slabs additionally: the same particles in the same order as SlabDomain, whose
trajectories a BrickDomain run therefore reproduces bit for bit (asserted on the GPU in tests/test_gpu_brick.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT, brute_nlist, sq_lattice


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ids(w):
    return (w.contiguous().view(torch.int64) & 0xFFFFFFFF).numpy()


def _worker(rank, world, port, grid, fractions, q, n, expect_interior, local_grid=False):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from hoomd_tf_amd import _lib, standin
        from hoomd_tf_amd.brick import BrickDomain
        from hoomd_tf_amd.domain import SlabDomain
        from oracle import htf_oracle as O

        a, rcut, rbuf = 1.3, 2.5, 0.4
        pos, L = sq_lattice(n, a)
        rng = np.random.default_rng(7)
        pos[:, :2] += 0.08 * rng.standard_normal((n * n, 2))
        pos = pos - np.floor((pos + L / 2) / L) * L
        vel = np.zeros((n * n, 4))
        vel[:, :2] = 0.6 * rng.standard_normal((n * n, 2))
        vel[:, 3] = 1.0
        ids = np.arange(n * n)

        def make(kind):
            if kind == "brick":
                probe = standin.System(pos[:1], L, dtype=torch.float64, device="cpu")
                d0 = BrickDomain(probe, rank, grid, r_ghost=rcut + rbuf, fractions=fractions, n_global=n * n)
                mine = np.ones(n * n, dtype=bool)
                for d in range(2):
                    mine &= (pos[:, d] >= d0.lo[d]) & (pos[:, d] < d0.hi[d])
            else:
                cuts = np.linspace(0, 1, world + 1) if fractions is None else np.concatenate([[0], fractions[0], [1]])
                b = -L[0] / 2 + cuts * L[0]
                mine = (pos[:, 0] >= b[rank]) & (pos[:, 0] < b[rank + 1])
            system = standin.System(pos[mine], L, types=ids[mine], dtype=torch.float64, device="cpu")
            system.vel = torch.from_numpy(vel[mine]).clone()
            if kind == "brick":
                return system, BrickDomain(system, rank, grid, r_ghost=rcut + rbuf, r_buff=rbuf, fractions=fractions, n_global=n * n,
                                           local_grid=local_grid)
            return system, SlabDomain(system, rank, world, r_ghost=rcut + rbuf, fractions=None if fractions is None else fractions[0])

        system, dom = make("brick")
        slab_sys, slab = make("slab") if grid[1] == 1 and not local_grid else (None, None)

        def ghosts_are_their_owners(g_xyz, gid, global_pos, fresh=False):
            if not local_grid:
                np.testing.assert_array_equal(g_xyz, global_pos[gid])
                return
            # a local cell grid: a ghost sits NEXT TO the brick -- its owner's position, moved by whole box vectors where its
            # message crossed the periodic boundary -- inside the brick + ghost layer
            d = g_xyz - global_pos[gid]
            k = np.round(d / L)
            np.testing.assert_allclose(d, k * L, rtol=0, atol=1e-12)
            assert np.all(np.abs(k) <= 1) and np.all(k[:, 2] == 0)
            b3, per = dom.nlist_box()
            for ax in dom.axes:   # (right after a rebuild, which is when the list is binned)
                assert per[ax] == 0 and (not fresh or np.all((g_xyz[:, ax] >= b3[0][ax]) & (g_xyz[:, ax] < b3[1][ax])))
        cap = dom.cap
        assert system.N == cap and system.n_ghost == dom.n_ghost_cap and system.pos.shape[0] == cap + dom.n_ghost_cap

        def check(global_pos):
            dom.rebuild()
            p_all = system.pos.numpy()
            live = ~np.isnan(p_all[:cap, 0])
            c = dom.counts_host()
            n_int, n_bnd = int(c[_lib.BC_N_INT]), int(c[_lib.BC_N_BND])
            # layout: particles first in each segment, inert rows behind them
            assert np.array_equal(np.nonzero(live)[0], np.concatenate([np.arange(n_int), dom.cap_int + np.arange(n_bnd)]))
            my_ids = _ids(system.pos[:cap, 3])[live]
            # (1) ownership, and nobody is lost
            for d in dom.axes:
                assert np.all((p_all[:cap][live, d] >= dom.lo[d]) & (p_all[:cap][live, d] < dom.hi[d]))
            cnt = torch.tensor([int(live.sum())])
            dist.all_reduce(cnt)
            assert int(cnt) == n * n
            assert dom.n_local == int(live.sum())
            # (2) halo: ghosts carry their owner's CURRENT position; every particle within r_ghost of my brick (through the
            # periodic images) that is not mine is among them
            g = p_all[cap:]
            glive = ~np.isnan(g[:, 0])
            gid = _ids(system.pos[cap:, 3])[glive]
            ghosts_are_their_owners(g[glive, :3], gid, global_pos, fresh=True)
            np.testing.assert_array_equal(p_all[:cap][live, :3], global_pos[my_ids])
            # (3) forces over local + ghost rows == single-domain forces (inert rows: no neighbors, zero force)
            with np.errstate(invalid="ignore"):
                nn, head, nl = brute_nlist(p_all[:, :3], L, rcut + rbuf, n_local=cap)
            assert np.all(nn[~live] == 0)
            safe = np.where(np.isnan(p_all[:, :3]), 0.0, p_all[:, :3])   # (never read through a neighbor list)
            f, _ = O.compute_forces(safe, np.zeros(len(p_all), np.int32), nn, head, nl, O.make_box(L), rcut, 64,
                                    lambda t: O.lj_model(t.astype(np.float64)), model_dtype=np.float64, n_local=cap)
            gn, gh, gl = brute_nlist(global_pos, L, rcut + rbuf)
            fg, _ = O.compute_forces(global_pos.copy(), np.zeros(n * n, np.int32), gn, gh, gl, O.make_box(L), rcut, 64,
                                     lambda t: O.lj_model(t.astype(np.float64)), model_dtype=np.float64)
            np.testing.assert_allclose(f[live], fg[my_ids], atol=1e-5)
            assert np.all(f[~live] == 0)
            # (4) rows [0, n_interior) have no ghost neighbor; the boundary segment is ordered by class
            assert dom.n_interior == dom.cap_int
            for i in range(n_int):
                assert np.all(nl[head[i]:head[i] + nn[i]] < cap), "interior row %d has a ghost neighbor" % i
            x = p_all[:cap]
            key = np.zeros(cap, dtype=np.int64)
            for k, d in enumerate(dom.axes):
                lo_, hi_ = x[:, d] < dom.lo[d] + dom.r_ghost, x[:, d] >= dom.hi[d] - dom.r_ghost
                key += np.where(lo_, np.where(hi_, 2, 1), np.where(hi_, 3, 0)) << (2 * k)
            assert np.all(key[:n_int] == 0) and np.all(key[dom.cap_int:dom.cap_int + n_bnd] > 0)
            assert np.all(np.diff(key[dom.cap_int:dom.cap_int + n_bnd]) >= 0)
            assert (n_int > 0) == expect_interior, n_int
            return my_ids, live

        def slab_same():
            """BrickDomain(p, 1, 1) holds SlabDomain's particles in SlabDomain's order, ghosts included."""
            if slab is None:
                return
            slab.rebuild()
            N = slab_sys.N
            b = system.pos.numpy()
            live = ~np.isnan(b[:cap, 0])
            np.testing.assert_array_equal(b[:cap][live], slab_sys.pos[:N].numpy())
            np.testing.assert_array_equal(system.vel.numpy()[live], slab_sys.vel.numpy())
            gl = ~np.isnan(b[cap:, 0])
            np.testing.assert_array_equal(b[cap:][gl], slab_sys.pos[N:N + slab_sys.n_ghost].numpy())
            assert slab.n_interior == int(dom.counts_host()[_lib.BC_N_INT])

        gpos = pos.copy()
        check(gpos)
        slab_same()
        for step in range(3):
            gpos = gpos + 0.4 * vel[:, :3]
            gpos = gpos - np.floor((gpos + L / 2) / L) * L
            live_rows = dom.live_rows()
            my_ids = _ids(system.pos[:cap, 3])[live_rows.numpy()]
            system.pos[live_rows, :3] = torch.from_numpy(gpos[my_ids])
            if slab is not None:
                slab_sys.pos[:slab_sys.N, :3] = torch.from_numpy(gpos[slab_sys.types_numpy()])
                slab.exchange()
            if step == 1:
                dom.exchange_begin()
                assert dom.pending
                dom.exchange_end()
                assert not dom.pending
            else:
                dom.exchange()
            g = system.pos[cap:]
            glive = ~torch.isnan(g[:, 0])
            ghosts_are_their_owners(g[glive, :3].numpy(), _ids(g[:, 3])[glive.numpy()], gpos)
        ids_after, live = check(gpos)
        slab_same()
        moved = torch.tensor([dom.n_migrated])
        dist.all_reduce(moved)
        assert int(moved) > 0, "test must exercise migration"
        np.testing.assert_array_equal(system.vel.numpy()[live], vel[ids_after])           # velocities travelled with their particles
        assert np.all(system.vel.numpy()[~live] == np.array([0, 0, 0, 1.0]))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


def test_brick_geometry_is_checked():
    from hoomd_tf_amd import standin
    from hoomd_tf_amd.brick import BrickDomain
    pos, L = sq_lattice(8, 1.3)
    system = standin.System(pos, L, dtype=torch.float64, device="cpu")
    with pytest.raises(ValueError, match="same peer"):
        BrickDomain(system, 0, (2, 1, 1), r_ghost=2.9, n_global=64)          # bricks 5.2 < 2 * 2.9
    with pytest.raises(ValueError, match="beyond the adjacent brick"):
        BrickDomain(system, 0, (4, 1, 1), r_ghost=2.9, n_global=64)          # bricks 2.6 < 2.9
    with pytest.raises(ValueError, match="one or two axes"):
        BrickDomain(system, 0, (1, 1, 1), r_ghost=2.9, n_global=64)
    d = BrickDomain(standin.System(pos, L, dtype=torch.float64, device="cpu"), 0, (3, 1, 1), r_ghost=2.9, n_global=64, replica=True)
    assert d.world == 1 and d.neighbors == [0, 0] and d.coords == (1, 0, 0) and d.shift[0, 0] > 0 > d.shift[1, 0]


# (grid, fractions, lattice side, interior rows expected)
#   8 x 1, n 24: L = 31.2, slabs 3.9 wide, r_ghost 2.9 < 3.9 < 5.8 -- the geometry of the 131 072-particle box over 8 slabs: no interior;
#   4 x 2, n 32: L = 41.6, bricks 10.4 x 20.8 (> 2 r_ghost both ways) -- the same 8 ranks keep interior rows;
#   2 x 2 and 3 x 2 with uneven cuts: both faces of an axis leading to one peer, edge messages across the periodic corner
CASES = [((2, 1, 1), None, 32, True), ((2, 1, 1), {0: [0.33], 1: None, 2: None}, 32, True), ((3, 1, 1), None, 32, True),
         ((8, 1, 1), None, 24, False), ((4, 2, 1), None, 32, True), ((2, 2, 1), None, 32, True),
         ((3, 2, 1), {0: [0.3, 0.62], 1: [0.45], 2: None}, 40, True)]


@pytest.mark.parametrize("grid,fractions,n,interior,local_grid", [c + (False,) for c in CASES] + [((4, 2, 1), None, 32, True, True),
                                                                                                   ((3, 1, 1), None, 32, True, True),
                                                                                                   ((2, 2, 1), None, 32, True, True)])
def test_brick_domain_gloo(grid, fractions, n, interior, local_grid):
    world = grid[0] * grid[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, grid, fractions, q, n, interior, local_grid)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", "rank %d failed:\n%s" % (rank, msg)


def _replan_worker(rank, world, port, grid, n, k, q):
    """BrickDomain(replan_every=k) between real ranks: particles drift by just under r_buff / 2 per period; after every rebuild --
    the k - 1 of k that leave the plan alone included -- every pair within the LIST radius of a local row has its partner among
    this rank's local + ghost rows (the neighbor count of every row equals the undivided box's), and one more period later every
    pair within r_cut still does (forces == the undivided box's): what the thicker ghost layer is for."""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from hoomd_tf_amd import _lib, standin
        from hoomd_tf_amd.brick import BrickDomain
        from oracle import htf_oracle as O
        a, rcut, rbuf = 1.3, 2.5, 0.4
        pos, L = sq_lattice(n, a)
        rng = np.random.default_rng(11)                       # (the same stream on every rank: one global trajectory)
        pos[:, :2] += 0.08 * rng.standard_normal((n * n, 2))
        pos = pos - np.floor((pos + L / 2) / L) * L
        ids = np.arange(n * n)
        probe = standin.System(pos[:1], L, dtype=torch.float64, device="cpu")
        d0 = BrickDomain(probe, rank, grid, r_ghost=rcut + rbuf, r_buff=rbuf, n_global=n * n, replan_every=k)
        mine = np.ones(n * n, dtype=bool)
        for d in range(2):
            mine &= (pos[:, d] >= d0.lo[d]) & (pos[:, d] < d0.hi[d])
        system = standin.System(pos[mine], L, types=ids[mine], dtype=torch.float64, device="cpu")
        dom = BrickDomain(system, rank, grid, r_ghost=rcut + rbuf, r_buff=rbuf, n_global=n * n, replan_every=k, local_grid=True)
        assert abs(dom.r_ghost - (rcut + rbuf + (k - 1) * rbuf)) < 1e-12
        cap = dom.cap
        gpos = pos.copy()

        def move():
            nonlocal gpos
            step = rng.standard_normal((n * n, 3))
            step[:, 2] = 0.0
            step *= (0.19 * rng.random((n * n, 1)) ** 0.5) / np.linalg.norm(step, axis=1, keepdims=True)     # |step| < r_buff / 2
            gpos = gpos + step
            gpos = gpos - np.floor((gpos + L / 2) / L) * L
            live = dom.live_rows()
            my = _ids(system.pos[:cap, 3])[live.numpy()]
            system.pos[live, :3] = torch.from_numpy(gpos[my])

        def complete(radius, what):
            p_all = system.pos.numpy()
            live = ~np.isnan(p_all[:cap, 0])
            my = _ids(system.pos[:cap, 3])[live]
            with np.errstate(invalid="ignore"):
                nn, _, _ = brute_nlist(p_all[:, :3], L, radius, n_local=cap)
            gn, _, _ = brute_nlist(gpos, L, radius)
            assert np.array_equal(nn[live], gn[my]), "%s: %d rows miss a partner" % (what, int((nn[live] != gn[my]).sum()))

        n_light = 0
        for period in range(2 * k + 1):
            dom.rebuild()                                     # index 0, k, 2k: migrate + re-plan; the others: the halo alone
            c = dom.counts_host()
            assert int(c[_lib.BC_FLAGS]) == 0
            if period % k:
                n_light += 1
            else:
                live = dom.live_rows().numpy()
                p = system.pos.numpy()[:cap][live]
                for d in dom.axes:
                    assert np.all((p[:, d] >= dom.lo[d]) & (p[:, d] < dom.hi[d]))      # a full rebuild leaves everybody at home
            assert dom.n_light == n_light
            complete(rcut + rbuf, "list radius at rebuild %d" % period)
            move()
            dom.exchange()
            complete(rcut, "r_cut one period after rebuild %d" % period)
        assert n_light == 2 * (k - 1)
        cnt = torch.tensor([int(len(dom.live_rows()))])
        dist.all_reduce(cnt)
        assert int(cnt) == n * n
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("grid,n,k", [((2, 1, 1), 32, 2), ((4, 2, 1), 32, 2), ((3, 1, 1), 32, 3)])
def test_fewer_replans_keep_every_pair(grid, n, k):
    world = grid[0] * grid[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_replan_worker, args=(r, world, port, grid, n, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", "rank %d failed:\n%s" % (rank, msg)
