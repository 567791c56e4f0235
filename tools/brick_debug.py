"""Replica-mode forces against the replicated single-domain box at bench geometry.  python tools/brick_debug.py 4x2x1 [cells]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hoomd_tf_amd as htf  # noqa: E402
from hoomd_tf_amd import _lib, standin  # noqa: E402
from hoomd_tf_amd.brick import BrickDomain  # noqa: E402

grid = tuple(int(v) for v in sys.argv[1].split("x"))
ncells = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
cells = np.array([ncells // g for g in grid])
a = (4.0 / 0.8442) ** (1.0 / 3.0)
base = np.array([[0.25, 0.25, 0.25], [0.75, 0.75, 0.25], [0.75, 0.25, 0.75], [0.25, 0.75, 0.75]])
ijk = np.stack(np.meshgrid(*[np.arange(c) for c in cells], indexing="ij"), -1).reshape(-1, 3)
Lb = cells * a
Lg = Lb * np.array(grid)
lo = -Lg / 2 + (np.array(grid) // 2) * Lb
rng = np.random.default_rng(3)
pos = ((ijk[:, None, :] + base[None]) * a).reshape(-1, 3) + 0.03 * a * rng.standard_normal((len(ijk) * 4, 3))
pos = pos - np.floor(pos / Lb) * Lb + lo
rcut, rbuf, NN = 3.0, 0.4, 128
sysm = standin.System(pos, Lg, types=np.arange(len(pos)), dtype=torch.float32, device=dev)
sysm.randomize_velocities(kT=1.0, seed=3)
nl = standin.CellNlist(sysm, r_cut=rcut, r_buff=rbuf, check_period=1)
dom = nl.domain = BrickDomain(sysm, 0, grid, r_ghost=rcut + rbuf, r_buff=rbuf, replica=True, transport="local")
nl.build()
ctx = htf.Context(r_cut=rcut, nneighs=NN, scalar_dtype=torch.float32, max_n=sysm.N)
ctx.set_potential(htf.Potential.lj())
nve = standin.NVE(sysm, 0.005)
arr = ctx.make_arrays(sysm.pos, sysm.N, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, sysm.force)


def compare(tag):
    torch.cuda.synchronize()
    live = dom.live_rows()
    p = sysm.pos[live, :3].double().cpu().numpy()
    got = sysm.force[live].cpu().numpy()
    reps = np.stack(np.meshgrid(*[np.arange(g) for g in grid], indexing="ij"), -1).reshape(-1, 3)
    allp = np.concatenate([p - lo + r * Lb - Lg / 2 for r in reps])
    allp -= np.floor((allp + Lg / 2) / Lg) * Lg
    rs = standin.System(allp, Lg, dtype=torch.float32, device=dev)
    rn = standin.CellNlist(rs, r_cut=rcut, r_buff=rbuf)
    rn.build()
    rc = htf.Context(r_cut=rcut, nneighs=NN, scalar_dtype=torch.float32, max_n=rs.N)
    rc.set_potential(htf.Potential.lj())
    rc.compute_forces(0, rc.make_arrays(rs.pos, rs.N, rn.n_neigh, rn.head_list, rn.nlist, rs.box, rs.force))
    torch.cuda.synchronize()
    mine = int(np.nonzero((reps == np.asarray(grid) // 2).all(axis=1))[0][0])
    want = rs.force.cpu().numpy()[mine * len(p):(mine + 1) * len(p)]
    err = np.abs(got - want).max(axis=1)
    bad = np.nonzero(err > 1e-3 * np.abs(want).max())[0]
    inside = [(p[:, d] >= dom.lo[d] - 0.3).all() and (p[:, d] < dom.hi[d] + 0.3).all() for d in range(3)]
    c = dom.counts_host()
    nn_got = nl.n_neigh[live].cpu().numpy()
    nn_ref = rn.n_neigh.cpu().numpy()[mine * len(p):(mine + 1) * len(p)]
    print(tag, "live", len(live), "max err", err.max(), "scale", np.abs(want).max(), "bad rows", len(bad), "inside", inside,
          "n_int", int(c[_lib.BC_N_INT]), "msgs", c[_lib.BC_MSG:_lib.BC_MSG + dom.n_msg].tolist(),
          "neighbor count diffs", int((nn_got != nn_ref).sum()), flush=True)
    if len(bad):
        b = bad[:8]
        print("   bad rows at", np.round(p[b] - lo, 2).tolist(), "nn got/ref", nn_got[b].tolist(), nn_ref[b].tolist(), flush=True)


ctx.compute_forces_overlapped(0, arr, dom)
compare("t=0")
for ts in range(1, 61):
    f3 = sysm.force[:, :3]
    f3.mul_(torch.clamp(200.0 / f3.norm(dim=1, keepdim=True).clamp_min(1e-12), max=1.0))
    nve.step()
    nl.compute(ts)
    ctx.compute_forces_overlapped(ts, arr, dom)
    if ts in (1, 5, 20, 60):
        compare("t=%d builds %d migrated %d" % (ts, nl.n_builds, dom.n_migrated))
