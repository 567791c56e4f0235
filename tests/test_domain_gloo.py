"""Multi-rank path on CPU: 2 (and 3) gloo ranks run the slab decomposition -- migration,
ghost plan, per-step halo -- and the per-rank oracle forces over local+ghost particles must
equal the single-domain forces (the reference's own MPI assertion,
test_mpi_tensorflow.py:57-79, for the even and the uneven 0.33 split)."""
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


def _worker(rank, world, port, fractions, q, n=32, thin=False):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from hoomd_tf_amd import standin
        from hoomd_tf_amd.domain import SlabDomain
        from oracle import htf_oracle as O

        # global system: 32 x 32 square lattice (N = 1024) as in test_mpi_tensorflow.py, a = 1.3
        a, rcut, rbuf = 1.3, 2.5, 0.4
        pos, L = sq_lattice(n, a)
        rng = np.random.default_rng(7)
        pos[:, :2] += 0.08 * rng.standard_normal((n * n, 2))
        pos = pos - np.floor((pos + L / 2) / L) * L
        vel = np.zeros((n * n, 4))
        vel[:, :2] = 0.6 * rng.standard_normal((n * n, 2))
        ids = np.arange(n * n)
        dom_tmp_bounds = None
        cuts = np.linspace(0, 1, world + 1) if fractions is None else np.concatenate([[0], fractions, [1]])
        bounds = -L[0] / 2 + cuts * L[0]
        mine = (pos[:, 0] >= bounds[rank]) & (pos[:, 0] < bounds[rank + 1])
        # particle identity rides in the type slot so the test can track who is where
        system = standin.System(pos[mine], L, types=ids[mine], dtype=torch.float64, device="cpu")
        system.vel = torch.from_numpy(vel[mine]).clone()
        dom = SlabDomain(system, rank, world, r_ghost=rcut + rbuf, fractions=fractions)

        def check(global_pos):
            dom.rebuild()
            N = system.N
            p_all = system.pos.numpy()
            my_ids = system.types_numpy()
            # (1) ownership: every local particle lies in my slab, and nobody is lost
            assert np.all((p_all[:N, 0] >= dom.xlo) & (p_all[:N, 0] < dom.xhi))
            cnt = torch.tensor([N])
            dist.all_reduce(cnt)
            assert int(cnt) == n * n
            # (2) halo: ghosts carry their owner's CURRENT position
            gid = (system.pos[N:, 3].contiguous().view(torch.int64) & 0xFFFFFFFF).numpy()
            np.testing.assert_array_equal(p_all[N:, :3], global_pos[gid])
            np.testing.assert_array_equal(p_all[:N, :3], global_pos[my_ids])
            # (3) forces over local+ghost == single-domain forces (reference MPI assertion)
            nn, head, nl = brute_nlist(p_all[:, :3], L, rcut + rbuf, n_local=N)
            f, _ = O.compute_forces(p_all[:, :3].copy(), np.zeros(len(p_all), np.int32), nn, head, nl,
                                    O.make_box(L), rcut, 64, lambda t: O.lj_model(t.astype(np.float64)),
                                    model_dtype=np.float64, n_local=N)
            gn, gh, gl = brute_nlist(global_pos, L, rcut + rbuf)
            fg, _ = O.compute_forces(global_pos.copy(), np.zeros(n * n, np.int32), gn, gh, gl, O.make_box(L), rcut, 64,
                                     lambda t: O.lj_model(t.astype(np.float64)), model_dtype=np.float64)
            np.testing.assert_allclose(f, fg[my_ids], atol=1e-5)
            # (4) layout [interior | left only | both | right only]: the halo messages are two
            # (possibly overlapping) tail slices, and no interior row has a ghost in its neighbor
            # list -- what lets the force compute run rows [0, n_interior) while the halo is in flight
            ni = dom.n_interior
            assert dom.send_left[0] == ni and dom.send_right[1] == N and dom.send_right[0] <= dom.send_left[1]
            near_l, near_r = x_near = (p_all[:N, 0] < dom.xlo + dom.r_ghost), (p_all[:N, 0] >= dom.xhi - dom.r_ghost)
            assert np.array_equal(np.nonzero(near_l)[0], np.arange(*dom.send_left))
            assert np.array_equal(np.nonzero(near_r)[0], np.arange(*dom.send_right))
            x = p_all[:N, 0]
            assert np.all(x[dom.send_left[0]:dom.send_left[1]] < dom.xlo + dom.r_ghost)
            assert np.all(x[dom.send_right[0]:dom.send_right[1]] >= dom.xhi - dom.r_ghost)
            assert np.all((x[:ni] >= dom.xlo + dom.r_ghost) & (x[:ni] < dom.xhi - dom.r_ghost))
            for i in range(ni):
                assert np.all(nl[head[i]:head[i] + nn[i]] < N), "interior row %d has a ghost neighbor" % i
            c0, c1, c2, c3 = dom.class_counts
            assert (c0, c0 + c1 + c2 + c3) == (ni, N)
            if thin:
                # slab < 2 r_ghost: nothing is interior, some particles are ghosts on BOTH sides and
                # the two send slices overlap on exactly those rows
                assert ni == 0 and c2 > 0 and dom.send_left[1] - dom.send_right[0] == c2
            else:
                assert 0 < ni < N and c2 == 0
            return my_ids

        gpos = pos.copy()
        check(gpos)
        # move everything (some particles cross slab faces and the periodic boundary), then
        # per-step halo refresh and a second migration
        for step in range(3):
            gpos = gpos + 0.4 * vel[:, :3]
            gpos = gpos - np.floor((gpos + L / 2) / L) * L
            N = system.N
            my_ids = system.types_numpy()
            system.pos[:N, :3] = torch.from_numpy(gpos[my_ids])
            if step == 1:  # split form: interior work would go between the two calls
                dom.exchange_begin()
                assert dom.pending
                dom.exchange_end()
                assert not dom.pending
            else:
                dom.exchange()
            gid = (system.pos[N:, 3].contiguous().view(torch.int64) & 0xFFFFFFFF).numpy()
            np.testing.assert_array_equal(system.pos[N:, :3].numpy(), gpos[gid])
        ids_after = check(gpos)
        moved = torch.tensor([dom.n_migrated])
        dist.all_reduce(moved)
        assert int(moved) > 0, "test must exercise migration"
        # velocities travelled with their particles
        np.testing.assert_array_equal(system.vel.numpy(), vel[ids_after])
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


def test_two_thin_slabs_are_refused():
    from hoomd_tf_amd import standin
    from hoomd_tf_amd.domain import SlabDomain
    pos, L = sq_lattice(8, 1.3)
    system = standin.System(pos, L, dtype=torch.float64, device="cpu")
    with pytest.raises(ValueError, match="same peer"):
        SlabDomain(system, 0, 2, r_ghost=2.9)          # slabs 5.2 < 2 * 2.9
    with pytest.raises(ValueError, match="beyond the adjacent slab"):
        SlabDomain(system, 0, 4, r_ghost=2.9)          # slabs 2.6 < 2.9
    SlabDomain(system, 0, 3, r_ghost=2.9)              # slabs 3.47: thin but legal


# world 8, n 24: L = 31.2, slabs 3.9 wide, r_ghost 2.9 < 3.9 < 5.8 -- the geometry of the 131 072-particle
# box over 8 ranks (slab 6.72 against r_ghost 3.4), which the round-1 layout refused
@pytest.mark.parametrize("world,fractions,n,thin", [(2, None, 32, False), (2, [0.33], 32, False), (3, None, 32, False),
                                                     (8, None, 24, True)])
def test_slab_domain_gloo(world, fractions, n, thin):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fractions, q, n, thin)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", "rank %d failed:\n%s" % (rank, msg)
