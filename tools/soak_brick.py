#!/usr/bin/env python3
"""Soak of the REPLAYED decomposed step: one rank's brick of the 131 072-particle LJ box at the 8-rank geometry (replica mode, as
bench.py --workload dd-self), --steps MD steps from the two hipGraphs per rank (standin.BrickRun.run(n, graph=True)), the rebuild
decided one check late from the pinned word the check kernel writes.  Every --every steps: total energy per particle (kinetic +
half the pair energy of the live rows), kT, rebuilds, dangerous builds, migrants, the decomposition's flags -- what a long run of
this path must keep: energy conserved to the integrator's fluctuation, no drift, no overflow flag, every particle accounted for.
    python tools/soak_brick.py [--grid 8x1x1] [--transport local|peer|native] [--steps 20000] > profiles/r05_soak_brick_<grid>_<transport>.json"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hoomd_tf_amd as htf  # noqa: E402
from hoomd_tf_amd import _lib, standin  # noqa: E402
from hoomd_tf_amd.brick import BrickDomain  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--grid", default="8x1x1")
ap.add_argument("--transport", default="local")
ap.add_argument("--steps", type=int, default=20000)
ap.add_argument("--every", type=int, default=1000)
ap.add_argument("--cells", type=int, default=32)
ap.add_argument("--replan-every", type=int, default=1)
a = ap.parse_args()

dev = torch.device("cuda:0")
grid = tuple(int(v) for v in a.grid.split("x"))
cells = np.array([a.cells // g for g in grid])
lat = (4.0 / 0.8442) ** (1.0 / 3.0)
base = np.array([[0.25, 0.25, 0.25], [0.75, 0.75, 0.25], [0.75, 0.25, 0.75], [0.25, 0.75, 0.75]])
ijk = np.stack(np.meshgrid(*[np.arange(c) for c in cells], indexing="ij"), -1).reshape(-1, 3)
Lb = cells * lat
Lg = Lb * np.array(grid)
lo = -Lg / 2 + (np.array(grid) // 2) * Lb
rng = np.random.default_rng(3)
pos = ((ijk[:, None, :] + base[None]) * lat).reshape(-1, 3)
pos = pos + 0.05 * lat * rng.standard_normal(pos.shape)
pos = pos - np.floor(pos / Lb) * Lb + lo
n_rank = len(pos)
rcut, rbuff, NN, P, dt = 3.0, 0.4, 128, 5, 0.005

sysm = standin.System(pos, Lg, dtype=torch.float32, device=dev)
sysm.randomize_velocities(kT=1.0, seed=3)
nl = standin.CellNlist(sysm, r_cut=rcut, r_buff=rbuff, check_period=P, device_decision=True)
dom = nl.domain = BrickDomain(sysm, 0, grid, r_ghost=rcut + rbuff, r_buff=rbuff, replica=True, transport=a.transport, replan_every=a.replan_every)
nl.build()
ctx = htf.Context(r_cut=rcut, nneighs=NN, scalar_dtype=torch.float32, max_n=sysm.N, check_nlist=False, fused=2)
ctx.set_potential(htf.Potential.lj())
nve = standin.NVE(sysm, dt)
run = standin.BrickRun(sysm, nl, ctx, nve)
run._arr = run._arrays()
for _ in range(300):      # relaxation: force cap + velocity rescale, as bench.py does
    ts = sysm.timestep
    b = nl.n_builds
    nl.compute(ts)
    if nl.n_builds != b:
        run._arr = run._arrays()
    ctx.compute_forces_overlapped(ts, run._arr, dom)
    f3 = sysm.force[:, :3]
    f3.mul_(torch.clamp(200.0 / f3.norm(dim=1, keepdim=True).clamp_min(1e-12), max=1.0))
    nve.step()
    v3 = sysm.vel[:, :3]
    v3.mul_(torch.sqrt(1.0 / ((v3 * v3).sum() / (3.0 * n_rank))))
    sysm.timestep += 1
run.run(200 + (-(sysm.timestep + 200)) % P)
run.run(20 * P, graph=True)


def sample():
    torch.cuda.synchronize()
    c = dom.counts_host()
    live = dom.live_rows()
    # the force array holds F(x(t)) of the LAST step's evaluation while the positions have moved on by one update: energies are
    # taken half a step apart, which is the leapfrog's own bookkeeping error (constant in time), not a drift
    pe = float(sysm.force[live, 3].double().sum()) / n_rank
    v3 = sysm.vel[live, :3].double()
    ke = 0.5 * float((v3 * v3).sum()) / n_rank
    return {"step": int(sysm.timestep), "particles": int(len(live)), "E_per_particle": pe + ke, "PE": pe, "kT": 2.0 * ke / 3.0,
            "list_rebuilds": int(nl.n_builds), "rebuild_cycles": int(run.n_rebuild_cycles), "cycles": int(run.n_cycles),
            "dangerous_builds": int(run.dangerous_builds), "rebuilds_without_a_replan": int(dom.n_light), "migrated_total": int(dom.n_migrated), "flags": int(c[_lib.BC_FLAGS]),
            "ghosts": int(dom.n_ghosts)}


samples = [sample()]
t0 = time.perf_counter()
for _ in range(a.steps // a.every):
    run.run(a.every, graph=True)
    samples.append(sample())
wall = time.perf_counter() - t0
E = np.array([s["E_per_particle"] for s in samples])
st = np.array([s["step"] for s in samples], dtype=np.float64)
slope = float(np.polyfit(st, E, 1)[0])
out = {
    "what": "standin.BrickRun.run(n, graph=True): brick %s of fcc %d^3 x 4 in replica mode, transport %s, %d particles + %d ghosts, r_cut %.1f, "
            "r_buff %.1f, NN %d, check_period %d, dt %g" % (a.grid, a.cells, a.transport, n_rank, samples[-1]["ghosts"], rcut, rbuff, NN, P, dt),
    "steps": a.steps, "us_per_step_including_the_samples": wall / a.steps * 1e6,
    "E_mean": float(E.mean()), "E_std": float(E.std()), "E_drift_per_step": slope, "E_drift_over_run_relative": abs(slope) * a.steps / abs(float(E.mean())),
    "kT_mean": float(np.mean([s["kT"] for s in samples])), "particles_conserved": all(s["particles"] == n_rank for s in samples),
    "flags_ever_set": int(np.bitwise_or.reduce([s["flags"] for s in samples])), "dangerous_builds": samples[-1]["dangerous_builds"],
    "steps_per_rebuild": a.steps / max(1, samples[-1]["rebuild_cycles"] - samples[0]["rebuild_cycles"]),
    "samples": samples,
}
assert out["particles_conserved"] and out["flags_ever_set"] == 0
print(json.dumps(out, indent=1))
