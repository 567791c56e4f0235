"""`--workload dd-self`: one rank's share of the decomposed step at an 8-rank geometry, on one GPU (BrickDomain in replica mode)."""
import json
import os
import sys
import time

import numpy as np
import torch

from .common import HBM_PEAK_GBS, PROF_EVERY, ROOT, algorithmic_bytes, gpu_state, make_potential  # noqa: F401


def run_dd_self(args, htf, standin, dev):
    """One rank's share of the decomposed step at an 8-rank geometry, on the one GPU of this box: BrickDomain in REPLICA mode --
    the rank is its own neighbor in every direction, its brick repeated px x py times IS the C3 box (fcc 32^3 x 4 = 131 072
    particles: 16 384 rows per rank + the ghosts of that cut) -- so rows, ghost rows, messages, launches and the rebuild are those
    of rank k of N.  Timed per transport: ``local`` (the pack kernel writes the ghosts: no communication library, the floor) and
    ``native`` (grouped ncclSend / ncclRecv of csrc/halo.hip, this rank sending to itself); eagerly (Python issues every launch)
    and replayed from two hipGraphs per check period (standin.BrickRun).  What crosses xGMI between real ranks is NOT measured."""
    from hoomd_tf_amd import _lib
    from hoomd_tf_amd.brick import BrickDomain
    args.grid = args.grid or "8x1x1"
    grid = tuple(int(v) for v in args.grid.lower().split("x"))
    grid = grid + (1,) * (3 - len(grid))
    cells = np.array([args.cells // grid[0], args.cells // grid[1], args.cells // grid[2]])
    assert np.all(cells * np.array(grid) == args.cells), "--cells must be divisible by the grid"
    a = (4.0 / 0.8442) ** (1.0 / 3.0)
    base = np.array([[0.25, 0.25, 0.25], [0.75, 0.75, 0.25], [0.75, 0.25, 0.75], [0.25, 0.75, 0.75]])
    ijk = np.stack(np.meshgrid(*[np.arange(c) for c in cells], indexing="ij"), -1).reshape(-1, 3)
    Lb = cells * a
    Lg = Lb * np.array(grid)
    coords = np.array(grid) // 2
    lo = -Lg / 2 + coords * Lb
    rng = np.random.default_rng(3)
    pos = ((ijk[:, None, :] + base[None]) * a).reshape(-1, 3)
    pos = pos + 0.05 * a * rng.standard_normal(pos.shape)
    pos = pos - np.floor(pos / Lb) * Lb + lo
    n_rank = len(pos)
    transports = ["local", "peer", "native"] if args.transport == "all" else [args.transport]
    if not _lib.lib.htf_halo_available():
        transports = [t for t in transports if t != "native"]
    P = args.check_period
    results = {}
    for transport in transports:
        sysm = standin.System(pos, Lg, dtype=torch.float32, device=dev)
        sysm.randomize_velocities(kT=1.0, seed=3)
        nl = standin.CellNlist(sysm, r_cut=args.rcut, r_buff=args.rbuff, check_period=P, device_decision=True)
        dom = nl.domain = BrickDomain(sysm, 0, grid, r_ghost=args.rcut + args.rbuff, r_buff=args.rbuff, replica=True, transport=transport,
                                      replan_every=args.replan_every or 1)
        nl.build()
        ctx = htf.Context(r_cut=args.rcut, nneighs=args.nn, scalar_dtype=torch.float32, max_n=sysm.N, check_nlist=False, fused=2)
        ctx.set_potential(htf.Potential.lj())
        nve = standin.NVE(sysm, args.dt)
        run = standin.BrickRun(sysm, nl, ctx, nve)
        run._arr = run._arrays()
        # relaxation: force cap + velocity rescale (the jittered lattice holds close pairs), then plain NVE
        for _ in range(args.equil):
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
        run.run(args.settle + (-(sysm.timestep + args.settle)) % P)          # plain NVE, ends on a check step
        rec = {}
        for mode in ("eager", "graph"):
            run.run(max(args.warmup, 4 * P) // P * P, graph=(mode == "graph"))
            wins = []
            steps = max(args.steps, P) // P * P
            b0, m0 = nl.n_builds, None
            for _ in range(args.windows or 5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run.run(steps, graph=(mode == "graph"))
                torch.cuda.synchronize()
                wins.append((time.perf_counter() - t0) / steps * 1e6)
            rec[mode] = {"us_per_step": float(np.median(wins)), "windows_us_per_step": wins, "steps": steps,
                         "rebuilds_per_window": (nl.n_builds - b0) / float(len(wins))}
        c = dom.counts_host()
        live = dom.live_rows()
        e = float(sysm.force[live, 3].double().sum()) / n_rank
        v3 = sysm.vel[live, :3].double()
        rec.update({"energy_per_particle": e, "kT": float((v3 * v3).sum() / (3.0 * n_rank)), "dangerous_builds": run.dangerous_builds,
                    "particles": int(len(live)), "interior_particles": int(c[_lib.BC_N_INT]), "ghosts": dom.n_ghosts,
                    "rows": sysm.N, "interior_rows": dom.cap_int, "ghost_rows": sysm.n_ghost, "messages_per_halo": dom.n_msg,
                    "halo_bytes_per_step": dom.n_ghost_cap * 16, "migrated": dom.n_migrated,
                    "replan_every": dom.replan_every, "rebuilds_without_a_replan": dom.n_light})
        if os.environ.get("HTF_DD_PHASES") == "1" and run._graphs is not None:
            # where a replayed cycle's time goes, without a profiler in the way: each of the two graphs replayed alone, back to
            # back, nothing read in between (the trajectory is garbage afterwards: this is the last thing done with the system)
            ph = {}
            gA, gB = run._graphs[False], run._graphs[True]
            for name, seq, n in (("rebuild_then_plain_cycle", (gB, gA), 100), ("plain_cycle", (gA,), 200)):
                for _ in range(4):
                    for g in seq:
                        g.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    for g in seq:
                        g.replay()
                t_issue = time.perf_counter() - t0
                torch.cuda.synchronize()
                ph[name] = {"us": (time.perf_counter() - t0) / n * 1e6, "host_us_to_launch": t_issue / n * 1e6}
            ph["steps_per_cycle"] = P
            rec["phases"] = ph
        results[transport] = rec
        del run, ctx, nl, dom, sysm
    best = min(results[t]["graph"]["us_per_step"] for t in results)
    line = {
        "metric": "MD steps/sec of ONE rank's decomposed step at the %s geometry of the 131072-particle box (replica mode: this GPU is "
                  "its own neighbor; no byte crosses xGMI)" % args.grid,
        "value": 1e6 / best, "unit": "steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": best / 1000.0,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "dd-self: brick %s of fcc %d^3 x 4 (%d particles per rank), rho 0.8442, r_cut %.1f, r_buff %.1f, NN %d, "
                               "check_period %d" % (args.grid, args.cells, n_rank, args.rcut, args.rbuff, args.nn, P),
                   "value_is": "the fastest transport's replayed (hipGraph) step"},
        "transports": results,
    }
    print(json.dumps(line))
