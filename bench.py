#!/usr/bin/env python3
"""Headline benchmark: MD steps/s + achieved HBM GB/s, 131 072 particles, NN = 128.

    python bench.py --gpus N --steps K --warmup W
        N > 1 without WORLD_SIZE in the environment: this process starts N rank processes itself
        (fresh children, before anything touches the GPU) and relays rank 0's JSON line
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over the whole particle batch, exactly what
HOOMD's run loop does around TensorflowCompute::computeForces (TensorflowCompute.cc:
129-216): neighbor-list distance check (rebuild when tripped), pair-vector build,
force/energy evaluation written into the HOOMD force array, and the integrator update
that moves the particles so the next step sees new input.  All inputs are resident in
HBM before the timed region starts.

Workload (SURVEY 8(d) "C3-LJ", BASELINE.json metric): fcc 4*32^3 = 131 072 particles,
rho = 0.8442, Gaussian jitter 0.05 a (seed 3), r_cut = 3.0, r_buff = 0.4, NN = 128,
LJModel, fp32, dt = 0.005, Maxwell velocities kT = 1.0.
N > 1, --scaling strong (default): THAT box decomposed into N slabs along x (BASELINE metric:
"131k particles ... at 1/2/4/8 GPUs"); --scaling weak: every rank owns one such block of an
N-block box (config 5: 8 x 131 072 = 1.05 M particles).  `value` is always the MD steps/s of the
GLOBAL system; `particle_steps_per_s` = global particles x value.

Prints ONE JSON line (rank 0).  roofline.achieved = ALGORITHMIC bytes per launch /
average launch duration measured with hipEvents on the launch stream inside the timed
region (htf_profile_*); see DESIGN.md "Measurement".
"""
import argparse
import json
import os
import sys
import time

# the host driver only supports dmabuf IPC: RCCL's cross-process buffer sharing fails without this
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md "Chip-level parameters")
PROF_EVERY = 7         # bracket every 7th force batch with hipEvents inside the timed region (see main; 14 samples per 100 steps)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="lj", choices=["lj", "wca", "mlp", "mlp-fp32", "mlp-split", "mlp-bf16", "mlp-train", "eds", "ref-lj256", "c1", "ex01", "generic-lj", "dd-self"])
    ap.add_argument("--grid", default=None, help="rank grid PXxPYx1: --gpus N > 1: how the box is cut (default: slabs along x; e.g. 4x2x1 "
                                                 "for 8 ranks); dd-self: the grid whose one brick this GPU runs (default 8x1x1)")
    ap.add_argument("--transport", default="all", help="dd-self: local | peer | native | all")
    ap.add_argument("--replan-every", type=int, default=0,
                    help="BrickDomain(replan_every=k): only every k-th neighbor-list rebuild migrates and re-plans (a ghost layer (k - 1) r_buff "
                         "thicker).  0 = the default: 1 for --workload dd-self, 2 for --gpus N > 1 (there a re-plan is a grouped exchange)")
    ap.add_argument("--train-period", type=int, default=100, help="mlp-train (C5b): force-matching step every this many MD steps")
    ap.add_argument("--cells", type=int, default=32, help="fcc cells per side (N = 4 cells^3 per rank)")
    ap.add_argument("--lattice", default="fcc", choices=["fcc", "sc"], help="fcc: N = 4 cells^3 (C3, C5); sc: N = cells^3 (C2 = sc 32^3 = 32768)")
    ap.add_argument("--nn", type=int, default=128)
    ap.add_argument("--rcut", type=float, default=3.0)
    ap.add_argument("--rbuff", type=float, default=0.4)
    ap.add_argument("--dt", type=float, default=0.005)
    ap.add_argument("--check-period", type=int, default=5, help="nlist distance-check period (HOOMD check_period)")
    ap.add_argument("--equil", type=int, default=300, help="untimed relaxation steps (force cap + velocity rescale)")
    ap.add_argument("--settle", type=int, default=100, help="untimed plain NVE steps between the relaxation and the warmup")
    ap.add_argument("--sort", action="store_true", help="enable the stand-in's particle sorter (HOOMD SFCPack analogue; measured: no kernel gain)")
    ap.add_argument("--no-fused", action="store_true", help="skip the extra variants (two-kernel dataflow, tensor-less fused mode)")
    ap.add_argument("--one-kernel", action="store_true",
                    help="eds workload: the whole C4 sweep as one kernel (htf_build_eval_forces2) -- the default since the "
                         "tensor is written with streaming stores; --two-kernel selects build + eval2")
    ap.add_argument("--sync-train", action="store_true", help="mlp-train: run the training step on the MD stream (no overlap)")
    ap.add_argument("--two-kernel", action="store_true",
                    help="headline run with separate build and evaluator kernels (htf_config.fused = 0)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = the --cells box itself cut into N slabs (default; the BASELINE metric), "
                         "weak = N such boxes side by side, one per rank (config 5)")
    ap.add_argument("--windows", type=int, default=0,
                    help="timed windows of --steps steps each; `value` is their median (0 = 5 windows when --steps <= 50, else 1)")
    ap.add_argument("--host-nlist-decision", action="store_true",
                    help="read the neighbor-list distance check back to the host (round-1 behaviour); default: the "
                         "rebuild is gated on the device, the step loop never synchronises")
    ap.add_argument("--no-mlp", action="store_true", help="default N = 1 run: skip the pair-MLP sub-records")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--f64", action="store_true",
                    help="HOOMD built in double precision: fp64 positions / velocities / forces on the wire, fp32 pair vectors "
                         "and model arithmetic (the reference casts the fp64 buffer to the model dtype, simmodel.py:226-238)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def gpu_state(index=0):
    """Clocks and power cap of the GPU as sysfs shows them right now (a 10 % spread of an MFMA-bound kernel
    between boxes should be explained by a number).  Best effort: every field is optional."""
    import glob
    out = {}
    cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
    if not cards:
        return None
    dev = os.path.dirname(cards[min(index, len(cards) - 1)])

    def read(path):
        try:
            with open(path) as f:
                return f.read().strip()
        except OSError:
            return None

    for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk"):
        txt = read(os.path.join(dev, name))
        if txt:
            levels = [l.strip() for l in txt.splitlines()]
            cur = [l for l in levels if l.endswith("*")]
            out[name[7:] + "_now"] = cur[0].rstrip(" *").split(":")[-1].strip() if cur else None
            out[name[7:] + "_max"] = levels[-1].rstrip(" *").split(":")[-1].strip()
    for hw in glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
        for key, fname, scale in (("power_cap_W", "power1_cap", 1e-6), ("power_cap_max_W", "power1_cap_max", 1e-6),
                                  ("power_now_W", "power1_average", 1e-6), ("power_now_W", "power1_input", 1e-6),
                                  ("temp_edge_C", "temp1_input", 1e-3), ("sclk_hwmon_MHz", "freq1_input", 1e-6)):
            txt = read(os.path.join(hw, fname))
            if txt and key not in out:
                try:
                    out[key] = round(float(txt) * scale, 1)
                except ValueError:
                    pass
    out["perf_level"] = read(os.path.join(dev, "power_dpm_force_performance_level"))
    return out or None


def make_potential(htf, workload):
    if workload == "lj":
        return htf.Potential.lj()
    if workload == "wca":
        return htf.Potential.wca(1.0)
    from hoomd_tf_amd.initializers import mlp_params
    if workload == "mlp-train":
        # C5b = online force matching (example 06, FORCE_MODE::hoomd2tf): the reference LJ force
        # drives the MD; the pair-MLP is the model being trained, it does not push particles
        make_potential.layer = htf.PairMLP(32, 64, 64, 0.0, 3.0, activation="tanh", seed=3)
        return htf.Potential.lj()
    # "mlp": the default precision of PairMLP, fp32 operands as hi + lo in fp16 (DESIGN 3.3a''); the other three by name
    prec = {"mlp-bf16": "bf16", "mlp-split": "split", "mlp-fp32": "fp32"}.get(workload, "split16")
    return htf.Potential.pair_mlp(mlp_params(seed=3), 0.0, 3.0, activation="tanh", precision=prec)


def algorithmic_bytes(N, NN, n_list_entries, n_tot, s4=16):
    """per-launch algorithmic bytes of each kernel; s4 = bytes of a HOOMD Scalar4 (16 fp32, 32 fp64).
    The pair-vector tensor is fp32 either way."""
    eval_b = N * NN * 16 + N * s4
    build_b = N * 8 + n_list_entries * 4 + n_tot * s4 + N * NN * 16
    integ_b = N * s4 * 5  # pos r/w, vel r/w, force r
    return eval_b, build_b, integ_b


def cpu_baseline(sysm, nl, args):
    """Time the C restatement of the same computeForces pass (oracle/htf_oracle_c.c, OpenMP over the host
    cores this process may use) on the SAME inputs, for a bounded ~10 s.  Baseline only, never the product."""
    from oracle import c_oracle
    lib = c_oracle.load()
    pos4 = sysm.pos.cpu().numpy().astype(np.float32)
    nn = nl.n_neigh.cpu().numpy().view(np.uint32)
    head = nl.head_list.cpu().numpy().view(np.uint32)
    nlist = nl.nlist.cpu().numpy().view(np.uint32)
    N, NN = sysm.N, args.nn
    cores = int(lib.htfo_num_threads())
    if args.workload in ("mlp", "mlp-fp32", "mlp-split", "mlp-bf16"):
        # 24.8 kflop per slot with libm tanhf / expf: a full pass takes seconds, so a contiguous row
        # sample is timed (rows are independent) and scaled to the box
        from hoomd_tf_amd.initializers import mlp_params
        params = mlp_params(seed=3)
        rows = min(N, 8192)
        out = np.empty((rows, 4), dtype=np.float32)

        def one():
            pv = c_oracle.prepare_neighbors(lib, pos4, nn, head, nlist, sysm.box3x3, args.rcut, NN, offset=0, batch=rows)
            c_oracle.mlp_from_nlist(lib, pv, params, 0.0, 3.0, act="tanh", out=out)
        what = ("computeForces passes (prepareNeighbors + pair-MLP RBF(0,3,32)-64-64-1 tanh with the analytic backward, "
                "C/OpenMP restatement, fp32) over rows [0, %d) of the same %d x %d workload, scaled to all rows" % (rows, N, NN))
        scale = rows / float(N)
    else:
        rows = N
        scratch = np.empty((N, NN, 4), dtype=np.float32)
        if args.workload == "wca":
            import ctypes as C
            force = np.empty((N, 4), dtype=np.float32)
            lo, hi, tilt, per = c_oracle._box_args(sysm.box3x3, (1, 1, 1))
            p = c_oracle._p

            def one():
                lib.htfo_compute_forces_wca_f32(p(pos4), C.c_uint(N), p(nn), p(head), p(nlist), p(lo), p(hi), p(tilt), p(per),
                                                C.c_double(args.rcut), C.c_uint(NN), C.c_float(1.0), p(scratch), p(force))
            model = "WCA model (WCARepulsion sigma 1.0)"
        else:
            def one():
                c_oracle.compute_forces_lj(lib, pos4, nn, head, nlist, sysm.box3x3, args.rcut, NN, scratch)
            model = "LJModel"
        what = ("computeForces passes (prepareNeighbors + %s, C/OpenMP restatement, fp32) over the same %d x %d workload"
                % (model, N, NN))
        scale = 1.0
    one()
    t0 = time.perf_counter()
    reps = 0
    while True:
        one()
        reps += 1
        el = time.perf_counter() - t0
        if el > args.cpu_seconds or reps >= 400:
            break
    out = {"value": reps / el * scale, "unit": "steps/s", "cores": cores, "kind": "port",
           "sample": "%d %s; integrator not included" % (reps, what)}
    if args.workload != "lj":
        return out
    # SURVEY 8(d) also asks for the GRAPH-STYLE restatement: the reference's op sequence (one pass
    # over [rows, NN] per TF op, forward + tf.gradients) as torch-CPU ops on a bounded row sample
    try:
        from oracle import c_oracle as _co, graph_torch
        ncpu = _co.usable_cpus()
        torch.set_num_threads(ncpu)
        rows = min(sysm.N, 32768)
        x = torch.from_numpy(scratch[:rows].copy())
        graph_torch.lj_model(x)
        t0, r2 = time.perf_counter(), 0
        while time.perf_counter() - t0 < min(args.cpu_seconds, 6.0) and r2 < 50:
            graph_torch.lj_model(x)
            r2 += 1
        dt = (time.perf_counter() - t0) / max(r2, 1)
        out["graph_style"] = {"value": 1.0 / (dt * sysm.N / rows), "unit": "steps/s (evaluator only, extrapolated from the row sample)",
                              "cores": ncpu, "sample": "%d passes of the op-for-op LJModel graph (torch CPU, autograd) over %d of %d rows"
                                                       % (r2, rows, sysm.N)}
    except Exception as e:  # noqa: BLE001 -- the baseline is informational
        out["graph_style"] = {"error": str(e)}
    return out


def run_ref_lj256(args, htf, standin, dev):
    """The one benchmark the reference publishes (htf/test-py/benchmark.py:25-48, ~498-510 steps/s on a
    Xeon Gold 6130 / 6140 node): 256 particles on hoomd.lattice.sq(a=2.0), LJModel(NN=64) attached through
    tfcompute with r_cut 3.0, nlist.cell(check_period=1), dt 0.005, 1000 steps x 5 rounds, median.  Upstream also runs
    HOOMD's own pair.lj and a Langevin thermostat in the same steps; the stand-in integrates NVE at kT = 1 and has no second
    force, so this line measures the plugin path's per-step cost at a size where nothing but overhead counts."""
    n, a, NN, rcut = 16, 2.0, 64, 3.0
    L = np.array([n * a, n * a, 1.0])
    ij = np.stack(np.meshgrid(np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 2)
    pos = np.zeros((n * n, 3))
    pos[:, :2] = (ij + 0.5) * a - L[:2] / 2
    sysm = standin.System(pos, L, dtype=torch.float32, device=dev)
    sysm.randomize_velocities(kT=1.0, seed=42)
    sysm.vel[:, 2] = 0.0  # two-dimensional, as hoomd.lattice.sq

    class LJModel(htf.SimModel):
        def compute(self, nlist, positions, box):
            rinv = htf.nlist_rinv(nlist)
            inv_r6 = rinv**6
            p_energy = 4.0 / 2.0 * (inv_r6 * inv_r6 - inv_r6)
            energy = htf.reduce_sum(p_energy, axis=1)
            return htf.compute_nlist_forces(nlist, energy)

    sim = standin.Simulation(sysm)
    sim.integrate_nve(0.005)
    tfc = htf.tfcompute(LJModel(NN))
    cell = sim.nlist_cell(r_buff=0.4, check_period=1, pitch=NN)  # the 2-D fluid clusters: rows well above the mean density's
    tfc.attach(cell, r_cut=rcut)
    sim.run(max(args.equil, 200))  # first step traces the model; the rest mixes the lattice
    rounds = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sim.run(1000, graph=False)  # (step by step; a bare run() of this length would pick the replay by itself)
        torch.cuda.synchronize()
        rounds.append(time.perf_counter() - t0)
    el = float(np.median(rounds))
    # the same loop with whole steps replayed from a hipGraph (Simulation.run(graph=True)): one launch per step
    g_rounds = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sim.run(1000, graph=True)
        torch.cuda.synchronize()
        g_rounds.append(time.perf_counter() - t0)
    g_el = float(np.median(g_rounds[1:]))  # the first round captures
    f = tfc.force
    assert bool(torch.isfinite(f).all())
    published = 1000.0 / 2.0071  # median of the newer of the two published runs
    out = {
        "metric": "MD steps/sec, the reference's published benchmark workload (256 particles, LJModel NN=64, through tfcompute)",
        "value": 1000.0 / el, "unit": "steps/s", "n_gpus": 1, "steps": 1000, "warmup": max(args.equil, 200),
        "ms_per_step": el, "higher_is_better": True, "scaling": "weak", "dtype": "f32", "data": "synthetic",
        "vs_baseline": (1000.0 / el) / published,
        "baseline": {"value": published, "unit": "steps/s", "where": "BASELINE.md: test_lj_benchmark, median 2.0071 s per 1000 steps, "
                     "Xeon Gold 6130 node, TF2 graph + HOOMD pair.lj + Langevin in the same steps (device mode not recorded)"},
        "config": {"workload": "htf/test-py/benchmark.py: sq lattice 16 x 16, a = 2.0, r_cut 3.0, r_buff 0.4, check_period 1, dt 0.005; "
                               "stand-in NVE at kT = 1 instead of HOOMD Langevin + pair.lj", "rounds_s": rounds},
        "replayed": tfc._plan is not None,
        "graph_variant": {"value": 1000.0 / g_el, "unit": "steps/s", "vs_baseline": (1000.0 / g_el) / published,
                          "captured": getattr(sim, "_graph", None) is not None, "rounds_s": g_rounds,
                          "note": "sim.run(1000, graph=True): the step (device-side list check + gated rebuild + force kernel + "
                                  "integrator) captured once, replayed as one hipGraph launch per step; same trajectory bit for bit "
                                  "(tests/test_gpu_standin.py::test_graphed_run_equals_stepwise)"},
        "energy_per_particle": float(f[:, 3].double().sum().item()) / sysm.N,
        "roofline": None, "cpu_baseline": None,
        "note": "overhead-bound at this size: the whole step is host enqueue (nlist check + one kernel + integrate)",
    }
    print(json.dumps(out))


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


def run_generic_lj(args, htf, standin, dev):
    """What leaving the zoo costs: the reference's defining capability is an ARBITRARY compute() (htf/simmodel.py:87-121) whose
    forces come from tf.gradients (simmodel.py:526-555).  Here a model outside the lowered closed forms / MLPs runs as torch
    eager ops on the zero-copy [N, NN, 4] tensor with torch.autograd for the forces.  The same LJModel twice, at
    C2 (32 768) and C3 (131 072) size, through tfcompute: written with the htf.* expression layer (lowered to the one-kernel step,
    replayed without Python) and written in plain torch ops (generic route: build kernel + ~20 eager ops + autograd every step)."""
    NN, rcut = args.nn, args.rcut

    class LJModel(htf.SimModel):
        def compute(self, nlist, positions, box):
            rinv = htf.nlist_rinv(nlist)
            inv_r6 = rinv**6
            p_energy = 4.0 / 2.0 * (inv_r6 * inv_r6 - inv_r6)
            energy = htf.reduce_sum(p_energy, axis=1)
            return htf.compute_nlist_forces(nlist, energy)

    class TorchLJModel(htf.SimModel):
        def compute(self, nlist, positions, box):
            delta = 3e-6  # nlist_rinv op for op (simmodel.py:618-635)
            r = torch.sqrt(torch.sum((nlist[:, :, :3] + delta / 3 / 10) ** 2, dim=2))
            rinv = torch.where(r > delta, 1.0 / (r + delta), torch.zeros_like(r))
            inv_r6 = rinv ** 6
            p_energy = 4.0 / 2.0 * (inv_r6 * inv_r6 - inv_r6)
            energy = torch.sum(p_energy, dim=1)
            return htf.compute_nlist_forces(nlist, energy)

    class MorseModel(htf.SimModel):
        """Outside the zoo, written with htf.* ops: a Morse well (D = 1, a = 5, r0 = 1.122) masked to the list's live slots.
        Traced into a generated kernel (HTF_POT_JIT, hoomd_tf_amd/codegen.py), replayed as the one-kernel step."""
        def compute(self, nlist, positions, box):
            r = htf.safe_norm(nlist[:, :, :3], axis=2)
            live = htf.cast(htf.nlist_rinv(nlist) > 0.0, torch.float32)
            x = 1.0 - htf.exp(-5.0 * (r - 1.122))
            energy = htf.reduce_sum(0.5 * live * (x * x - 1.0), axis=1)
            return htf.compute_nlist_forces(nlist, energy)

    class YukawaLJModel(htf.SimModel):
        """Outside the zoo: LJ plus a screened Coulomb term 0.5 exp(-r) / r (traced; generated kernel)."""
        def compute(self, nlist, positions, box):
            r = htf.safe_norm(nlist[:, :, :3], axis=2)
            s = htf.nlist_rinv(nlist)
            energy = htf.reduce_sum(2.0 * (s ** 12 - s ** 6) + 0.25 * htf.exp(-1.0 * r) * s, axis=1)
            return htf.compute_nlist_forces(nlist, energy)

    class TorchYukawaLJModel(htf.SimModel):
        def compute(self, nlist, positions, box):
            delta = 3e-6
            t = nlist[:, :, :3] + 1e-7
            r = torch.sqrt(torch.sum(t * t, dim=2))
            s = torch.where(r > delta, 1.0 / (r + delta), torch.zeros_like(r))
            energy = torch.sum(2.0 * (s ** 12 - s ** 6) + 0.25 * torch.exp(-1.0 * r) * s, dim=1)
            return htf.compute_nlist_forces(nlist, energy)

    KA_EPS, KA_SIG = [1.0, 1.5, 1.5, 0.5], [1.0, 0.8, 0.8, 0.88]   # Kob-Andersen 80:20 binary LJ: AA, AB, BA, BB

    class MixtureModel(htf.SimModel):
        """Outside the zoo AND typed: a binary LJ mixture whose epsilon and sigma are looked up by species pair -- tf.gather on
        ti * 2 + tj, the way a multi-component model is written against the reference -- traced into ONE generated kernel."""
        def compute(self, nlist, positions, box):
            s = htf.nlist_rinv(nlist)
            idx = htf.cast(positions[:, 3], torch.int32)[:, None] * 2 + htf.cast(nlist[:, :, 3], torch.int32)
            q = (htf.gather(KA_SIG, idx) * s) ** 6
            energy = htf.reduce_sum(2.0 * htf.gather(KA_EPS, idx) * (q * q - q), axis=1)
            return htf.compute_nlist_forces(nlist, energy)

    class TorchMixtureModel(htf.SimModel):
        def compute(self, nlist, positions, box):
            delta = 3e-6
            t = nlist[:, :, :3] + 1e-7
            r = torch.sqrt(torch.sum(t * t, dim=2))
            s = torch.where(r > delta, 1.0 / (r + delta), torch.zeros_like(r))
            idx = (positions[:, 3:4] * 2 + nlist[:, :, 3]).detach().long()
            q = (torch.tensor(KA_SIG, device=s.device)[idx] * s) ** 6
            energy = torch.sum(2.0 * torch.tensor(KA_EPS, device=s.device)[idx] * (q * q - q), dim=1)
            return htf.compute_nlist_forces(nlist, energy)

    class IonicModel(htf.SimModel):
        """Outside the zoo, typed, with a special function: LJ cores plus the real-space part of Ewald / damped-shifted-force
        electrostatics between +1 / -1 species, q_i q_j erfc(alpha r) / r, the charges gathered by species pair."""
        def compute(self, nlist, positions, box):
            s = htf.nlist_rinv(nlist)
            r = htf.safe_norm(nlist[:, :, :3], axis=2)
            idx = htf.cast(positions[:, 3], torch.int32)[:, None] * 2 + htf.cast(nlist[:, :, 3], torch.int32)
            qq = htf.gather([1.0, -1.0, -1.0, 1.0], idx)
            energy = htf.reduce_sum(2.0 * (s ** 12 - s ** 6) + 0.5 * 2.0 * qq * htf.erfc(0.35 * r) * s, axis=1)
            return htf.compute_nlist_forces(nlist, energy)

    class TorchIonicModel(htf.SimModel):
        def compute(self, nlist, positions, box):
            delta = 3e-6
            t = nlist[:, :, :3] + 1e-7
            r = torch.sqrt(torch.sum(t * t, dim=2))
            s = torch.where(r > delta, 1.0 / (r + delta), torch.zeros_like(r))
            idx = (positions[:, 3:4] * 2 + nlist[:, :, 3]).detach().long()
            qq = torch.tensor([1.0, -1.0, -1.0, 1.0], device=s.device)[idx]
            energy = torch.sum(2.0 * (s ** 12 - s ** 6) + 0.5 * 2.0 * qq * torch.erfc(0.35 * r) * s, dim=1)
            return htf.compute_nlist_forces(nlist, energy)

    def one(lattice, cells, model_cls, steps):
        pos, L, a = (standin.sc_positions if lattice == "sc" else standin.fcc_positions)(cells, 0.8442)
        rng = np.random.default_rng(7)
        pos = pos + 0.03 * a * rng.standard_normal(pos.shape)
        pos -= np.round(pos / L) * L
        types = (rng.random(len(pos)) < 0.2).astype(np.int32) if model_cls in (MixtureModel, TorchMixtureModel) else None
        if model_cls in (IonicModel, TorchIonicModel):
            types = (np.arange(len(pos)) % 2).astype(np.int32)    # (equal numbers of the two species: a neutral system)
        sysm = standin.System(pos, L, types=types, dtype=torch.float32, device=dev)
        sysm.randomize_velocities(kT=1.0, seed=7)
        sim = standin.Simulation(sysm)
        sim.integrate_nve(args.dt)
        tfc = htf.tfcompute(model_cls(NN))
        tfc.attach(sim.nlist_cell(r_buff=args.rbuff, check_period=args.check_period), r_cut=rcut)
        sim.run(max(5, args.warmup))
        torch.cuda.synchronize()
        e_warm = float(tfc.force[:, 3].double().sum().item()) / sysm.N   # (compared between routes: same step count here)
        els = []
        for _ in range(3):   # (windows of 15-400 ms: the median of three keeps a one-off stall -- a lazy module load, the
            t0 = time.perf_counter()                                       # run loop's own graph-or-not measurement -- out of the line)
            sim.run(steps)
            torch.cuda.synchronize()
            els.append(time.perf_counter() - t0)
        el = sorted(els)[1]
        f = tfc.force
        assert bool(torch.isfinite(f).all())
        return {"steps_per_s": steps / el, "ms_per_step": el / steps * 1e3, "particles": sysm.N, "steps": steps,
                "windows_ms_per_step": [e / steps * 1e3 for e in els], "replayed_without_python": tfc._plan is not None,
                "potential_kind": getattr(tfc._plan, "kind", None), "energy_per_particle_after_warmup": e_warm,
                "energy_per_particle": float(f[:, 3].double().sum().item()) / sysm.N}

    sizes = {}
    for tag, lattice, cells in (("C2 (sc 32^3 = 32768)", "sc", 32), ("C3 (fcc 32^3 x 4 = 131072)", "fcc", 32)):
        fast = one(lattice, cells, LJModel, args.steps)
        gen = one(lattice, cells, TorchLJModel, max(20, args.steps // 10))
        assert abs(fast["energy_per_particle_after_warmup"] - gen["energy_per_particle_after_warmup"]) < 1e-3 * abs(fast["energy_per_particle_after_warmup"]) + 1e-3
        # round 5: models OUTSIDE the zoo written with htf.* ops are traced into generated kernels (HTF_POT_JIT)
        yuk = one(lattice, cells, YukawaLJModel, args.steps)
        yuk_torch = one(lattice, cells, TorchYukawaLJModel, max(20, args.steps // 10))
        morse = one(lattice, cells, MorseModel, args.steps)
        mix = one(lattice, cells, MixtureModel, args.steps)
        mix_torch = one(lattice, cells, TorchMixtureModel, max(20, args.steps // 10))
        ion = one(lattice, cells, IonicModel, args.steps)
        ion_torch = one(lattice, cells, TorchIonicModel, max(20, args.steps // 10))
        assert ion["potential_kind"] == 9
        assert abs(ion["energy_per_particle_after_warmup"] - ion_torch["energy_per_particle_after_warmup"]) < 1e-3 * abs(ion_torch["energy_per_particle_after_warmup"]) + 1e-3
        assert yuk["potential_kind"] == 9 and morse["potential_kind"] == 9 and mix["potential_kind"] == 9
        assert abs(mix["energy_per_particle_after_warmup"] - mix_torch["energy_per_particle_after_warmup"]) < 1e-3 * abs(mix_torch["energy_per_particle_after_warmup"]) + 1e-3
        assert abs(yuk["energy_per_particle_after_warmup"] - yuk_torch["energy_per_particle_after_warmup"]) < 1e-3 * abs(yuk_torch["energy_per_particle_after_warmup"]) + 1e-3
        sizes[tag] = {"lowered": fast, "generic": gen, "generic_over_lowered_time": gen["ms_per_step"] / fast["ms_per_step"],
                      "traced_yukawa_lj": yuk, "traced_morse": morse, "torch_yukawa_lj": yuk_torch,
                      "traced_binary_mixture": mix, "torch_binary_mixture": mix_torch,
                      "traced_ionic": ion, "torch_ionic": ion_torch,
                      "ionic_over_lowered_lj_time": ion["ms_per_step"] / fast["ms_per_step"],
                      "torch_over_traced_ionic_time": ion_torch["ms_per_step"] / ion["ms_per_step"],
                      "mixture_over_lowered_lj_time": mix["ms_per_step"] / fast["ms_per_step"],
                      "torch_over_traced_mixture_time": mix_torch["ms_per_step"] / mix["ms_per_step"],
                      "traced_over_lowered_lj_time": yuk["ms_per_step"] / fast["ms_per_step"],
                      "torch_over_traced_time": yuk_torch["ms_per_step"] / yuk["ms_per_step"]}
    c3 = sizes["C3 (fcc 32^3 x 4 = 131072)"]
    out = {
        "metric": "MD steps/sec, LJModel written in plain torch ops (generic autograd route) at 131072 particles NN=%d" % NN,
        "value": c3["generic"]["steps_per_s"], "unit": "steps/s", "n_gpus": 1, "steps": c3["generic"]["steps"], "warmup": max(5, args.warmup),
        "ms_per_step": c3["generic"]["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "LJModel through tfcompute, htf.* expression layer (lowered) vs plain torch ops + torch.autograd (generic), "
                               "jittered lattices at rho 0.8442, r_cut %.1f, r_buff %.1f, NN %d, dt %g" % (rcut, args.rbuff, NN, args.dt)},
        "sizes": sizes,
        "traced_models": "written with htf.* ops outside the zoo (LJ + Yukawa; a masked Morse well; a Kob-Andersen binary LJ mixture whose "
                         "epsilon / sigma are gathered by species pair from positions[:, 3] and nlist[:, :, 3]; LJ cores + erfc-damped electrostatics between "
                         "two charged species): traced, lowered to generated kernels "
                         "(HTF_POT_JIT: hoomd_tf_amd/codegen.py -> hipcc --genco around csrc/jit_unit.hip), replayed as the one-kernel step",
        "note": "the generic route keeps the reference's arbitrary-model capability (htf/simmodel.py:87-121, 526-555); models made of "
                "nlist_rinv polynomials, WCARepulsion, RBFExpansion + Dense stacks, EDS biases and compute_rdf are lowered to fused kernels",
        "roofline": None, "cpu_baseline": None,
    }
    print(json.dumps(out))


def run_small(args, htf, standin, dev):
    """SURVEY 8(d) row C1 as written -- BASELINE configs[0], "LJ pair potential (example 01 Quickstart), 864 particles NN=64"
    -- in both readings: `--workload c1`: LJModel on 864 = 4 x 6^3 fcc particles, rho 0.8442, r_cut 2.5, NN 64, fp64 wire
    (HOOMD's default build) / fp32 model; `--workload ex01`: the notebook itself (examples/01. Quickstart.ipynb cells 3, 5):
    16 x 16 particles on sq(a = 1.2), WCAPotential(64) = r^-12 x cast(r < 2^(1/6)), r_cut 5, compute_rdf averaged every
    step, kT 0.5, dt 0.005 -- the notebook prints 488 steps/s (TF2 CPU path + HOOMD NVT, its own hardware).
    Both are host-enqueue-bound: reported through tfcompute step by step and, where the step is a fixed launch sequence,
    replayed from a hipGraph.  cpu_baseline: the numpy oracle of the same model on the same pair-vector shapes."""
    ex01 = args.workload == "ex01"
    if ex01:
        n, a, NN, rcut = 16, 1.2, 64, 5.0
        L = np.array([n * a, n * a, 1.0])
        ij = np.stack(np.meshgrid(np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 2)
        pos = np.zeros((n * n, 3))
        pos[:, :2] = (ij + 0.5) * a - L[:2] / 2
        sdt = torch.float64
        sysm = standin.System(pos, L, dtype=sdt, device=dev)
        sysm.randomize_velocities(kT=0.5, seed=1)
        sysm.vel[:, 2] = 0.0

        class Model(htf.SimModel):
            def setup(self):
                self.avg_rdf = htf.MeanTensor()  # tf.keras.metrics.MeanTensor in the notebook: on the device, part of the plan

            def compute(self, nlist):
                r12 = htf.nlist_rinv(nlist)**12
                r = htf.norm(nlist[:, :, :3], axis=2)
                pair_energy = htf.cast(r < 2**(1 / 6), torch.float32) * r12
                particle_energy = htf.reduce_sum(pair_energy, axis=1)
                forces = htf.compute_nlist_forces(nlist, particle_energy)
                inst_rdf = htf.compute_rdf(nlist, [0, 3.5])
                self.avg_rdf.update_state(inst_rdf)
                return forces
        what = ("examples/01. Quickstart.ipynb: sq lattice 16 x 16, a = 1.2 (256 particles, 2-D), WCAPotential(64) = rinv^12 x "
                "cast(r < 2^(1/6)), r_cut 5.0, r_buff 0.4, compute_rdf [0, 3.5] averaged every step, kT 0.5, dt 0.005; stand-in NVE "
                "instead of HOOMD NVT")
        published, where = 488.064, "the notebook's own output cell: TPS 488.064 (TF2 CPU path + HOOMD NVT, hardware not recorded)"
        pitch = 80
    else:
        NN, rcut = 64, 2.5
        pos, L, a = standin.fcc_positions(6, 0.8442)
        rng = np.random.default_rng(1)
        pos = pos + 0.02 * a * rng.standard_normal(pos.shape)
        pos -= np.round(pos / L) * L
        sdt = torch.float64
        sysm = standin.System(pos, L, dtype=sdt, device=dev)
        sysm.randomize_velocities(kT=1.0, seed=1)

        class Model(htf.SimModel):
            def compute(self, nlist, positions, box):
                rinv = htf.nlist_rinv(nlist)
                inv_r6 = rinv**6
                p_energy = 4.0 / 2.0 * (inv_r6 * inv_r6 - inv_r6)
                energy = htf.reduce_sum(p_energy, axis=1)
                return htf.compute_nlist_forces(nlist, energy)
        what = ("C1: LJModel (build_examples.py:67-77), 864 = 4 x 6^3 fcc particles, rho 0.8442, r_cut 2.5, r_buff 0.4, NN 64, "
                "fp64 wire / fp32 model, kT 1.0, dt 0.005, stand-in NVE")
        published, where = None, None
        pitch = None
    sim = standin.Simulation(sysm)
    sim.integrate_nve(0.005)
    model = Model(NN)
    tfc = htf.tfcompute(model)
    cell = sim.nlist_cell(r_buff=0.4, check_period=1, pitch=pitch)
    tfc.attach(cell, r_cut=rcut)
    sim.run(max(args.equil, 200))
    steps = 1000
    rounds = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sim.run(steps, graph=False)  # (step by step: the replayed loop is the graph_variant below; a bare run() would pick it by itself)
        torch.cuda.synchronize()
        rounds.append(time.perf_counter() - t0)
    el = float(np.median(rounds))
    graph = None
    if getattr(tfc, "graph_safe", lambda: False)():
        g_rounds = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sim.run(steps, graph=True)
            torch.cuda.synchronize()
            g_rounds.append(time.perf_counter() - t0)
        g_el = float(np.median(g_rounds[1:]))
        graph = {"value": steps / g_el, "unit": "steps/s", "captured": getattr(sim, "_graph", None) is not None, "rounds_s": g_rounds,
                 "vs_baseline": (steps / g_el) / published if published else None,
                 "note": "sim.run(n, graph=True): one check period of steps captured once and replayed as one hipGraph launch"}
    else:
        graph = {"value": None, "note": "not a fixed launch sequence"}
    # what Simulation.run(n) does by itself: its first steps timed both ways, the faster kept (sim.graph_choice)
    sim.graph_choice = None
    sim.run(320)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim.run(steps)
    torch.cuda.synchronize()
    ch = dict(getattr(sim, "graph_choice", None) or {})
    ch.pop("key", None)
    auto = {"note": "sim.run(n) with graph=None: stepwise or replayed, chosen by timing the run's own first steps both ways",
            "choice": ch, "value": steps / (time.perf_counter() - t0), "unit": "steps/s"}
    f = tfc.force
    assert bool(torch.isfinite(f).all())
    # cpu_baseline leg: the only place this workload touches oracle/.  It times the numpy oracle on this run's own pair vectors
    # and, since the oracle's output for them is then in hand, states how far the timed run's last step is from it.
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import htf_oracle as O
        nlv = tfc.get_nlist_array().astype(np.float32).astype(np.float64)
        ref = O.rinv_poly_model(nlv, [1.0], [12], cut=2 ** (1 / 6)) if ex01 else O.lj_model(nlv)
        err = np.abs(tfc.get_forces_array() - ref)
        bound = 1e-5 + 2e-5 * np.abs(ref)
        # the condition scale of a row's fp32 sum, sum_j |f_ij| (an equilibrated liquid's rows cancel 300 -> 10)
        s_, t_, rp_, cond_ = O._rinv_and_grad_factor(nlv)
        if ex01:
            x32 = nlv[:, :, :3].astype(np.float32)
            inside = np.sqrt((x32 * x32).sum(axis=2, dtype=np.float32)) < np.float32(2 ** (1 / 6))
            dEds = np.where(inside, 12.0 * s_ ** 11, 0.0)
        else:
            dEds = 2.0 * (2.0 * s_ ** 6 - 1.0) * (6.0 * s_ ** 5)
        csum = np.abs(2.0 * O._grad_from_dEds(dEds, s_, t_, rp_, cond_)).sum(axis=(1, 2))
        bound_c = bound + 2e-6 * csum[:, None]
        nl32 = nlv.astype(np.float32)
        fn = (lambda: O.rinv_poly_model(nl32, [1.0], [12], cut=2 ** (1 / 6))) if ex01 else (lambda: O.lj_model(nl32))
        fn()
        t0, reps = time.perf_counter(), 0
        while time.perf_counter() - t0 < min(args.cpu_seconds, 10.0):
            fn()
            reps += 1
        cpu = {"value": reps / (time.perf_counter() - t0), "unit": "steps/s", "cores": 1, "kind": "port",
               "sample": "%d evaluator passes of the numpy oracle (fp32, closed-form gradient) over this run's own [%d, %d, 4] pair "
                         "vectors; pair-vector build, neighbor list and integrator not included" % (reps, sysm.N, NN),
               "timed_run_last_step_vs_oracle": {
                   "max_abs_err": float(err.max()), "max_err_over_bound": float((err / bound).max()),
                   "max_err_over_bound_with_condition_term": float((err / bound_c).max()),
                   "energy_max_err_over_bound": float((err[:, 3] / bound[:, 3]).max()),
                   "bound": "1e-5 + 2e-5 |ref| (SURVEY 8(c), as stated; + 2e-6 sum_j |f_ij| for the condition-term figure) "
                            "vs the fp64 oracle on the same pair vectors, after %d MD steps" % (max(args.equil, 200) + 5 * steps)}}
    out = {
        "metric": "MD steps/sec, BASELINE configs[0] (%s)" % ("the Quickstart notebook as written" if ex01 else "864 particles NN=64 LJ"),
        "value": steps / el, "unit": "steps/s", "n_gpus": 1, "steps": steps, "warmup": max(args.equil, 200),
        "ms_per_step": el / steps * 1e3, "higher_is_better": True, "scaling": "weak", "data": "synthetic",
        "dtype": "f32 arithmetic on an f64 wire (HOOMD in double precision)",
        "vs_baseline": (steps / el) / published if published else None,
        "baseline": {"value": published, "unit": "steps/s", "where": where} if published else None,
        "config": {"workload": what, "rounds_s": rounds, "particles": sysm.N, "max_neighbors_listed": int(cell.n_neigh.max())},
        "replayed_as_one_kernel_plan": tfc._plan is not None,
        "graph_variant": graph,
        "auto_run": auto,
        "energy_per_particle": float(f[:, 3].double().sum().item()) / sysm.N,
        "roofline": None,
        "roofline_note": "host-enqueue-bound at this size: every kernel is ~1-3 us; the step is the launch sequence",
        "cpu_baseline": cpu,
    }
    print(json.dumps(out))


def run_eds(args, htf, standin, dev):
    """Config C4 (BASELINE configs[3]): 262 144 particles (sc 64^3), NN 128, LJModel
    + EDS bias on the soft RDF collective variable, hard compute_rdf [0, 3.5] as an observable
    every step.  One sweep over the pair vectors yields the LJ forces, the unit-bias forces
    and the CV; the EDS state machine and the force assembly run on the device."""
    import ctypes as C
    cells = args.cells if args.cells != 32 else 64
    pos, L, a = standin.sc_positions(cells, 0.8442)
    rng = np.random.default_rng(4)
    pos = pos + 0.05 * a * rng.standard_normal(pos.shape)
    pos -= np.round(pos / L) * L
    # --f64: HOOMD built in double precision (TensorflowCompute.h:117-124): fp64 positions in, fp64 forces out, the fp32 tensor
    # of simmodel.py:226-227's cast in between
    sdt = torch.float64 if args.f64 else torch.float32
    sysm = standin.System(pos, L, dtype=sdt, device=dev)
    sysm.randomize_velocities(kT=1.0, seed=4)
    nl = standin.CellNlist(sysm, r_cut=args.rcut, r_buff=args.rbuff, check_period=args.check_period)
    nl.build()
    N, NN = sysm.N, args.nn
    pv = torch.zeros((N, NN, 4), dtype=torch.float32, device=dev)
    bias = torch.empty((N, 4), dtype=sdt, device=dev)
    npart = htf.ops.num_partials(N, NN)
    partials = torch.empty(npart, dtype=torch.float32, device=dev)
    cv = torch.zeros(1, dtype=torch.float32, device=dev)
    lj, gauss = htf.Potential.lj(), htf.Potential.gauss(1.1, 0.05, 1.0)
    from hoomd_tf_amd.simmodel import rdf_from_histogram
    hist = torch.zeros(102, dtype=torch.int32, device=dev)
    eds = None  # created after the relaxation, with the set point 2 % above the natural CV
    nve = standin.NVE(sysm, args.dt)
    ev = {k: [] for k in ("build", "eval2")}
    state = {"ts": 0, "rdf": None, "time": False}

    def mark():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    # The sweep is ~0.4 ms of GPU work per step; keep the host side to a handful of raw C-ABI
    # calls with cached pointers so that the loop stays GPU-bound (torch only allocates once).
    from hoomd_tf_amd._lib import lib, check
    rdf_out = torch.empty(100, dtype=torch.float32, device=dev)
    rs_out = torch.empty(100, dtype=torch.float32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    F32 = 0
    SD = 1 if args.f64 else 0  # htf_dtype of HOOMD's Scalar
    ptr = {"nl": None}

    def refresh_ptrs():
        ptr["nl"] = (nl.n_neigh.data_ptr(), nl.nlist.data_ptr(), nl.head_list.data_ptr(), nl.n_builds)

    refresh_ptrs()

    npart_f = htf.ops.num_partials_fused(N)
    partials_f = torch.empty(npart_f, dtype=torch.float32, device=dev)
    ev["fused2"] = []

    def launch_all(stream, timed):
        """One C4 step on `stream`: a handful of C-ABI launches + one memset, no host synchronisation.
        Default: the whole sweep as ONE kernel (htf_build_eval_forces2 writes the tensor with
        streaming stores and evaluates both potentials, the CV partials and the RDF histogram from
        registers): 251 us.  --two-kernel: the build kernel, then htf_eval_forces2 re-reading the
        tensor: 191 + 157 us.  (Before the tensor stores were nontemporal the one-kernel form took
        369 us and the two kernels were the default.)"""
        hist.zero_()
        t0 = mark() if timed else None
        if not args.two_kernel:
            check(lib.htf_build_eval_forces2(lj.handle, gauss.handle, pv.data_ptr(), sysm.pos.data_ptr(), SD, N, NN, 0, N,
                                             C.byref(sysm.box), ptr["nl"][0], ptr["nl"][1], ptr["nl"][2], args.rcut,
                                             sysm.force.data_ptr(), bias.data_ptr(), SD, partials_f.data_ptr(),
                                             0.0, 3.5, 102, hist.data_ptr(), stream))
            t2 = mark() if timed else None
            if timed:
                ev["fused2"].append((t0, t2))
            check(lib.htf_reduce_partials(partials_f.data_ptr(), npart_f, 1.0 / N, cv.data_ptr(), stream))
        else:
            check(lib.htf_build_pair_vectors(pv.data_ptr(), F32, sysm.pos.data_ptr(), SD, N, NN, 0, N, 0, C.byref(sysm.box),
                                             ptr["nl"][0], ptr["nl"][1], ptr["nl"][2], args.rcut, None, stream))
            t1 = mark() if timed else None
            check(lib.htf_eval_forces2(lj.handle, gauss.handle, pv.data_ptr(), F32, N, NN, sysm.force.data_ptr(),
                                       bias.data_ptr(), SD, partials.data_ptr(), 0.0, 3.5, 102, hist.data_ptr(), stream))
            t2 = mark() if timed else None
            if timed:
                ev["build"].append((t0, t1))
                ev["eval2"].append((t1, t2))
            check(lib.htf_reduce_partials(partials.data_ptr(), npart, 1.0 / N, cv.data_ptr(), stream))
        if eds is not None:  # EDSLayer.__call__ + bias assembly, all on the device
            check(lib.htf_eds_update(eds.state.data_ptr(), cv.data_ptr(), eds.set_point, eds.period,
                                     eds.learning_rate, eds.cv_scale, stream))
            check(lib.htf_bias_combine(sysm.force.data_ptr(), bias.data_ptr(), eds.state.data_ptr() + 8,
                                       cv.data_ptr(), SD, N, stream))
        # compute_rdf(nlist, [0, 3.5]) every step: histogram fused above, tail here
        check(lib.htf_rdf_finalize(hist.data_ptr(), 100, 0.0, 3.5, rdf_out.data_ptr(), rs_out.data_ptr(), stream))

    def step(relax=False):
        ts = state["ts"]
        nl.compute(ts)
        if nl.n_builds != ptr["nl"][3]:
            refresh_ptrs()
        launch_all(stream, state["time"])
        state["rdf"] = rdf_out
        if relax:
            f3 = sysm.force[:, :3]
            f3.mul_(torch.clamp(200.0 / f3.norm(dim=1, keepdim=True).clamp_min(1e-12), max=1.0))
        nve.step()
        if relax:
            v3 = sysm.vel[:, :3]
            v3.mul_(torch.sqrt(1.0 / ((v3 * v3).sum() / (3.0 * N))))
        state["ts"] = ts + 1

    for _ in range(args.equil):
        step(relax=True)
    cv_nat = float(cv)
    eds = htf.EDSLayer(1.02 * cv_nat, 25, 0.05, device=dev)
    # kernel times from event-bracketed steps, wall time from un-instrumented ones: three event
    # objects per step made the loop host-bound (0.62 instead of 0.44 ms/step).  A hipGraph replay
    # of the step was tried as well: no gain, the loop is GPU-bound once the events are gone.
    state["time"] = True
    for _ in range(max(args.warmup, 10)):
        step()
    state["time"] = False
    torch.cuda.synchronize()
    # median, not mean: the first launch of the biased kernel variant can carry its code-object load (tens of ms, once)
    us = {k: 1e3 * float(np.median([a.elapsed_time(b) for a, b in v])) for k, v in ev.items() if v}
    for _ in range(5):
        step()
    b0 = nl.n_builds
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert bool(torch.isfinite(sysm.force).all())
    s4 = 32 if args.f64 else 16  # bytes of a HOOMD Scalar4
    eval_b = N * NN * 16 + 2 * N * s4
    build_b = N * 8 + int(nl.n_neigh.long().sum().item()) * 4 + N * s4 + N * NN * 16
    dom = max(us, key=us.get)
    # the one-kernel sweep is priced against its own compulsory bytes: the build's + the two force writes
    fused_b = build_b + 2 * N * s4
    dom_b = {"build": build_b, "eval2": eval_b, "fused2": fused_b}[dom]
    ach = dom_b / (us[dom] * 1e-6) / 1e9
    names = {"build": ("build_pair_vectors", build_b), "eval2": ("eval_forces2(lj+gauss+rdf)", eval_b),
             "fused2": ("build_eval_forces2(tensor + lj + gauss + cv + rdf)", fused_b)}
    kern = {names[k][0]: {"avg_us": v, "algorithmic_bytes": names[k][1], "GBps": names[k][1] / v / 1e3} for k, v in us.items()}
    if "fused2" in us:
        kern[names["fused2"][0]]["contract_GBps"] = (build_b + eval_b) / us["fused2"] / 1e3
    out = {
        "metric": "MD steps/sec (262144-particle EDS-on-RDF-CV domain steps, NN=128) + achieved HBM GB/s",
        "value": args.steps / elapsed, "unit": "steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if not args.f64 else "f32 arithmetic on an f64 wire (HOOMD in double precision: fp64 positions in, fp64 forces out)",
        "data": "synthetic",
        "config": {"workload": "C4-EDS: sc %d^3 = %d particles, rho 0.8442, r_cut %.1f, r_buff %.1f, NN %d, LJModel + "
                               "EDSLayer(1.02 x natural CV = %.3f, period 25, lr 0.05) on soft RDF bin r0 1.1 gap 0.05, "
                               "compute_rdf [0,3.5] fused into the sweep every step"
                               % (cells, N, args.rcut, args.rbuff, NN, 1.02 * cv_nat),
                   "nlist_rebuilds_in_timed_region": nl.n_builds - b0},
        "cv": float(cv), "alpha": float(eds.state[2]), "energy_per_particle": float(sysm.force[:, 3].double().sum()) / N,
        "rdf_peak": float(state["rdf"].max()),
        "kernels": kern,
        "roofline": {"bound": "hbm", "kernel": names[dom][0], "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ach / HBM_PEAK_GBS, "traffic": None},
        "cpu_baseline": None,
    }
    if not args.no_cpu_baseline:
        # the C/OpenMP restatement of the same step: prepareNeighbors + (LJ + alpha * soft-RDF CV forces, CV, compute_rdf
        # histogram) over the same 262 144 x 128 workload, a bounded number of passes
        from oracle import c_oracle
        clib = c_oracle.load()
        pos4 = sysm.pos.cpu().numpy().astype(np.float32)  # (the CPU port runs the fp32 wire either way)
        nn_h = nl.n_neigh.cpu().numpy().view(np.uint32)
        head_h = nl.head_list.cpu().numpy().view(np.uint32)
        nl_h = nl.nlist.cpu().numpy().view(np.uint32)
        f_h = np.empty((N, 4), dtype=np.float32)
        alpha_h = float(eds.state[2])

        def one():
            pvh = c_oracle.prepare_neighbors(clib, pos4, nn_h, head_h, nl_h, sysm.box3x3, args.rcut, NN)
            c_oracle.eds_from_nlist(clib, pvh, alpha_h, 1.1, 0.05, (0.0, 3.5), 102, out=f_h)
        one()
        t0, reps = time.perf_counter(), 0
        while True:
            one()
            reps += 1
            el = time.perf_counter() - t0
            if el > args.cpu_seconds or reps >= 200:
                break
        out["cpu_baseline"] = {"value": reps / el, "unit": "steps/s", "cores": int(clib.htfo_num_threads()), "kind": "port",
                               "sample": "%d computeForces passes (prepareNeighbors + LJModel + alpha * soft-RDF CV + compute_rdf histogram, "
                                         "C/OpenMP restatement, fp32) over the same %d x %d workload; EDS update and integrator not included"
                                         % (reps, N, NN)}
    print(json.dumps(out))


def count_gpus_sysfs():
    """GPUs of this node from the KFD topology (nodes with SIMDs), without touching the HIP runtime: on ROCm builds without
    amdsmi torch.cuda.device_count() falls through to hipGetDeviceCount, which initialises the GPU in the calling process."""
    import glob
    n = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            for line in open(f):
                k, _, v = line.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    n += 1
        except OSError:
            pass
    vis = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES"))
    if vis is not None and vis.strip() != "":
        n = min(n, len([x for x in vis.split(",") if x.strip() != ""]))
    return n


def self_launch(args):
    """`python bench.py --gpus N` with no launcher: start the N rank processes from here -- plain children
    of a parent that has not touched the GPU (never a re-exec) -- and relay rank 0's JSON line."""
    import socket
    import subprocess
    backend = os.environ.get("HTF_BENCH_BACKEND", "nccl")
    ndev = count_gpus_sysfs()  # the launcher parent stays GPU-free for certain: no HIP call, not even a device count
    if backend == "nccl" and ndev < args.gpus:
        print("bench.py --gpus %d: this node shows %d GPU(s).  RCCL wants one device per rank; "
              "HTF_BENCH_BACKEND=gloo rehearses the multi-rank path with the ranks sharing devices."
              % (args.gpus, ndev), file=sys.stderr)
        return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # rank 0's stdout carries the JSON line; the other ranks' stdout (library chatter) goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed_at = None
    while True:
        rcs = [pr.poll() for pr in procs]
        if all(rc is not None for rc in rcs):
            break
        if failed_at is None and any(rc not in (None, 0) for rc in rcs):
            failed_at = time.monotonic()  # a rank died: the others would wait for it in a collective forever
        if failed_at is not None and time.monotonic() - failed_at > 20.0:
            for pr in procs:
                if pr.poll() is None:
                    pr.kill()  # our own children, by handle
        time.sleep(0.1)
    rcs = [pr.wait() for pr in procs]
    reader.join(timeout=10)
    out = b"".join(chunks)
    for line in out.decode("utf-8", "replace").splitlines():
        # ONE JSON line on stdout; anything else a library printed there (gloo's connection notes) is passed to stderr
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


class Env:
    """What one rank process knows about the job."""


def main():
    args = parse()
    if os.environ.get("HTF_BENCH_WATCHDOG"):
        # a rank that hangs (mismatched collectives ...) dumps every thread's stack and exits instead of holding the box
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["HTF_BENCH_WATCHDOG"]), exit=True)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
    E = Env()
    E.world = world = int(os.environ.get("WORLD_SIZE", "1"))
    E.rank = rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the evaluator has no CPU path")
    # HTF_BENCH_BACKEND=gloo: rehearsal of the multi-rank code path with several ranks on ONE
    # GPU (RCCL refuses that); messages then bounce through host memory -- not a measurement
    E.backend = backend = os.environ.get("HTF_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    E.dev = dev = torch.device("cuda", local_rank)
    E.dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        E.dist = dist

    import hoomd_tf_amd as htf
    from hoomd_tf_amd import standin
    E.htf, E.standin = htf, standin

    if args.workload == "eds":
        if world > 1:
            raise SystemExit("the C4 workload is a single-GPU configuration")
        return run_eds(args, htf, standin, dev)
    if args.workload == "generic-lj":
        run_generic_lj(args, htf, standin, dev)
        return
    if args.workload == "dd-self":
        if world > 1:
            raise SystemExit("dd-self is one rank playing every brick")
        run_dd_self(args, htf, standin, dev)
        return
    if args.workload == "ref-lj256":
        if world > 1:
            raise SystemExit("the reference's own benchmark is a 256-particle, single-device workload")
        return run_ref_lj256(args, htf, standin, dev)
    if args.workload in ("c1", "ex01"):
        if world > 1:
            raise SystemExit("BASELINE configs[0] is a single-device plumbing case")
        return run_small(args, htf, standin, dev)
    headline = args.workload == "lj" and not args.f64 and not args.two_kernel
    E.live = None
    out = run_md(args, E, args.workload, variants=not args.no_fused, cpu=not args.no_cpu_baseline)
    live = E.live          # the LJ run's objects, for the guarded multi-rank section below
    if headline and not args.no_mlp and ((world == 1 and args.cells == 32) or (world > 1 and args.scaling == "strong")):
        # north_star: "LJ AND MLP pair-potential boxes ... at 1/2/4/8": the same C3 system driven by the pair-MLP in its default
        # precision (split16), a bounded number of steps -- on one GPU with the fp32-MFMA and bf16-split evaluators as variants,
        # on N ranks the box cut as the LJ run cuts it (per-rank `roofline`: rank 0's share of the evaluator)
        import copy
        a2 = copy.copy(args)
        a2.steps, a2.warmup, a2.equil, a2.windows = min(args.steps, 40), min(args.warmup, 5), min(args.equil, 100), 1
        sub = run_md(a2, E, "mlp", variants=not args.no_fused and world == 1, cpu=not args.no_cpu_baseline, keep_live=False)
        out["mlp"] = {k: sub[k] for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "kernels", "gpu_state",
                                         "roofline", "fp32_variant", "split_variant", "cpu_baseline", "energy_per_particle", "kT_final") if k in sub}
    if world > 1:
        from benchlib import multirank
        # who ran this: device uuid / PCI bus id per rank and what the process group says about itself
        out["ranks"] = multirank.rank_table(E)
        if live is not None and os.environ.get("HTF_BENCH_GUARDED", "1") != "0":
            def emit(extra):
                print(json.dumps(dict(out, **extra)))
                sys.stdout.flush()
            extra = multirank.guarded_section(E, live, out, emit)
            nat = getattr(live["nl"].domain, "_native", None)
            if nat is not None:
                # what RCCL says about the library's communicator, rank by rank
                mine = multirank.add_native_comm_info({"rank": rank}, nat)
                table = [None] * world
                E.dist.all_gather_object(table, mine)
                for row, t in zip(out["ranks"], table):
                    row["rccl_communicator"] = t.get("rccl_communicator")
            out.update(extra)
    if rank == 0:
        print(json.dumps(out))
    if E.dist is not None:
        E.dist.barrier()
        E.dist.destroy_process_group()


def run_md(args, E, workload, variants=True, cpu=True, keep_live=True):
    """One MD workload (lj | wca | mlp | mlp-split | mlp-bf16 | mlp-train) on this job's ranks -> the JSON record."""
    import copy
    args = copy.copy(args)
    args.workload = workload
    args.no_fused = not variants
    args.no_cpu_baseline = not cpu
    world, rank, dev, dist, htf, standin = E.world, E.rank, E.dev, E.dist, E.htf, E.standin
    if args.workload == "mlp-train":
        args.no_fused = True
    # ---- synthetic system, resident in HBM -------------------------------------------------
    # strong (default): the ONE 4*cells^3-particle box of the metric, every rank generates it identically
    # and keeps the particles of its slab.  weak: each rank owns one such block; the global periodic box
    # is `world` blocks side by side along x (config 5 at 8 ranks: 1.05 M particles, 8 x 1 x 1 slabs).
    strong = world > 1 and args.scaling == "strong"
    # the rank grid: slabs along x by default; --grid PXxPYx1 cuts bricks.  (8 slabs of the 131 072-particle box are 6.72 thick,
    # < 2 r_ghost: no row without a ghost neighbor, where a 4 x 2 cut keeps 37 % of the rows interior -- but a grouped RCCL
    # exchange of 8 messages was measured at 35 us against 17 for 2, profiles/r05_rccl_graph_probe.txt, more than the interior rows
    # can hide at 16 k rows per rank: DESIGN.md 6.4.  The 2-D cut pays with a latency-free transport or larger bricks.)
    if world > 1 and args.grid:
        grid = tuple(int(v) for v in args.grid.lower().split("x"))
        grid = grid + (1,) * (3 - len(grid))
    else:
        grid = (world, 1, 1)
    if int(np.prod(grid)) != world or (not strong and grid != (world, 1, 1)):
        raise SystemExit("--grid %s does not describe %d ranks (weak scaling: slabs along x)" % (args.grid, world))
    domain_kind = os.environ.get("HTF_BENCH_DOMAIN", "brick")   # "slab": round 4's variable-length SlabDomain
    pos, L, a = (standin.sc_positions if args.lattice == "sc" else standin.fcc_positions)(args.cells, 0.8442)
    rng = np.random.default_rng(3 + (0 if strong else rank))
    pos = pos + 0.05 * a * rng.standard_normal(pos.shape)
    pos -= np.round(pos / L) * L
    n_block = len(pos)
    Lg = L.copy()
    vel0 = None
    if strong:
        g = torch.Generator(device="cpu").manual_seed(3)
        vel0 = torch.randn((n_block, 3), generator=g, dtype=torch.float64)
        vel0 -= vel0.mean(dim=0, keepdim=True)
        mine = np.ones(n_block, dtype=bool)
        for d in range(3):   # this rank's brick of the grid (slabs: grid = (world, 1, 1))
            b = -L[d] / 2 + np.linspace(0.0, 1.0, grid[d] + 1) * L[d]
            c = (rank // int(np.prod(grid[:d]))) % grid[d]
            mine &= (pos[:, d] >= b[c]) & ((pos[:, d] < b[c + 1]) | (c == grid[d] - 1))
        pos, vel0 = pos[mine], vel0[torch.from_numpy(mine)]
        n_global = n_block
    else:
        Lg[0] = L[0] * world
        pos[:, 0] += (rank - (world - 1) / 2.0) * L[0]
        n_global = n_block * world
    sdt = torch.float64 if args.f64 else torch.float32
    s4 = 32 if args.f64 else 16
    if args.f64:
        args.no_cpu_baseline = True  # the C port is the fp32 build
    sysm = standin.System(pos, Lg, dtype=sdt, device=dev)
    if vel0 is None:
        sysm.randomize_velocities(kT=1.0, seed=3 + rank)
    else:
        sysm.vel[:, :3] = vel0.to(sdt).to(dev)
    nl = standin.CellNlist(sysm, r_cut=args.rcut, r_buff=args.rbuff, check_period=args.check_period,
                           sort_particles=args.sort,
                           # one rank: the rebuild is gated on the device; several ranks: the all-reduced distance check is
                           # read one check late (standin.DeferredRebuildRule) -- no read-back in the step loop either way
                           device_decision=(not args.sort and not args.host_nlist_decision))
    brick = world > 1 and domain_kind == "brick"
    if brick:
        # fixed-capacity arrays with inert rows: no read-back in a rebuild, addresses never change (hoomd_tf_amd/brick.py).
        # The native RCCL transport (csrc/halo.hip) has never run between two real devices: opt-in until it has
        from hoomd_tf_amd.brick import BrickDomain
        tr = os.environ.get("HTF_HALO_TRANSPORT", "torch")
        nl.domain = BrickDomain(sysm, rank, grid, r_ghost=args.rcut + args.rbuff, r_buff=args.rbuff, n_global=n_global,
                                transport=tr if tr in ("torch", "native") else "torch", replan_every=args.replan_every or 2)
    elif world > 1:
        from hoomd_tf_amd.domain import SlabDomain
        if grid != (world, 1, 1):
            raise SystemExit("SlabDomain cuts along x only")
        nl.domain = SlabDomain(sysm, rank, world, r_ghost=args.rcut + args.rbuff,
                               transport=os.environ.get("HTF_HALO_TRANSPORT", "torch"))
    nl.build()
    # rows of the arrays (a capacity under BrickDomain) and particles on this rank
    N_rows, NN = sysm.N, args.nn
    N = nl.domain.n_local if brick else sysm.N

    # closed-form potentials: ONE kernel builds the pair-vector tensor and evaluates it while it is in
    # registers (htf_config.fused = 2, the tfcompute default); the pair-MLP has its own MFMA evaluator
    closed_form = args.workload in ("lj", "wca", "mlp-train")
    one_kernel = closed_form and not args.two_kernel
    ctx = htf.Context(r_cut=args.rcut, nneighs=NN, scalar_dtype=sdt, max_n=N_rows, fused=2 if one_kernel else 0)
    pot = make_potential(htf, args.workload)
    ctx.set_potential(pot)
    nve = standin.NVE(sysm, args.dt)
    brun = standin.BrickRun(sysm, nl, ctx, nve) if brick else None

    _arr_cache = {}

    def arrays():
        # N changes when particles migrate between ranks at a rebuild; the position array alternates between two under the fused
        # step (standin.FusedStep): one htf_hoomd_arrays per set of addresses
        key = (sysm.pos.data_ptr(), sysm.N, nl.n_neigh.data_ptr(), nl.head_list.data_ptr(), nl.nlist.data_ptr(), sysm.force.data_ptr())
        if key not in _arr_cache:
            if len(_arr_cache) > 8:
                _arr_cache.clear()
            _arr_cache[key] = ctx.make_arrays(sysm.pos, sysm.N, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, sysm.force)
        return _arr_cache[key]

    state = {"builds": nl.n_builds, "ts": 0, "train_s": 0.0, "train_n": 0}

    train = None
    if args.workload == "mlp-train":
        # C5b: every --train-period steps (attach(train=True, period=100), running.rst:77-81)
        # one train_on_batch of the pair-MLP on THIS step's pair vectors, labels = this step's
        # LJ forces: prediction (MFMA evaluator) + loss-gradient sweep -> one RCCL all-reduce
        # of [loss, 6337 gradients, count] -> Adam on the device -> operand images rebuilt on
        # the device.  Inside the timed region.
        layer = make_potential.layer
        pot_mlp = layer.potential()
        opt_desc = htf.optimizers.Adam(1e-3).desc(0, (0.0,))
        opt_state = torch.zeros(htf.ops.optimizer_state_floats(layer.w.numel()), dtype=torch.float32, device=dev)
        # The trained model does not push particles (hoomd2tf), so the training step need not hold the MD
        # up: the step's pair vectors and labels are copied to a staging buffer (~0.1 ms on the main
        # stream) and the 11 ms sweep + all-reduce + optimizer + image refresh run on a SECOND stream
        # beside the following MD steps (MFMA-bound work next to HBM-bound work).  Same arithmetic,
        # same weights at the next training step; --sync-train keeps it on the main stream.
        side = torch.cuda.Stream(device=dev)
        cap = int(sysm.N * 1.1) + 1024
        stage_x = torch.empty((cap, NN, 4), dtype=torch.float32, device=dev)
        stage_y = torch.empty((cap, 4), dtype=torch.float32, device=dev)
        n_global = float(N)
        if dist is not None:
            t = torch.tensor([n_global], dtype=torch.float64, device=dev)
            dist.all_reduce(t)
            n_global = float(t.item())  # particles are conserved: no per-step count exchange
        train_events = []

        def train(timed):
            nonlocal stage_x, stage_y
            n = sysm.N
            main = torch.cuda.current_stream(dev)
            if n > stage_x.shape[0]:
                main.wait_stream(side)
                stage_x = torch.empty((int(n * 1.1), NN, 4), dtype=torch.float32, device=dev)
                stage_y = torch.empty((int(n * 1.1), 4), dtype=torch.float32, device=dev)
            if args.sync_train:
                x, y, where = ctx.nlist_buffer(n, dev), sysm.force[:n], main
            else:
                main.wait_stream(side)  # the previous training step has left the staging buffers
                stage_x[:n].copy_(ctx.nlist_buffer(n, dev))
                stage_y[:n].copy_(sysm.force[:n])
                side.wait_stream(main)
                x, y, where = stage_x[:n], stage_y[:n], side
            with torch.cuda.stream(where):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if timed:
                    e0.record()
                accum = htf.ops.train_pair_grad(pot_mlp, x, y)
                if dist is not None:
                    dist.all_reduce(accum)
                htf.ops.optimizer_step(layer.w, accum, 1.0 / (4.0 * n_global), opt_state, opt_desc)
                layer.after_update()
                if timed:
                    e1.record()
                    train_events.append((e0, e1))

    # the step as ONE launch where the context honours it (round 6, standin.FusedStep: the integrator -- and a brick's halo pack --
    # as the force kernel's epilogue, positions ping-ponging between two arrays); HTF_NO_STEP_EPILOGUE=1: the three pieces, as before
    fstep = None
    if train is None and one_kernel:
        fstep = brun.fstep if brun is not None else (standin.FusedStep(sysm, nl, ctx, nve) if world == 1 else None)
        if fstep is not None and not fstep.available:
            fstep = None
    state["fused_step"] = fstep is not None

    def step(timed=False):
        ts = state["ts"]
        nl.compute(ts)
        state["builds"] = nl.n_builds
        if fstep is not None:
            fstep.forces_and_integrate(ts)   # force rows (interior | halo | boundary, or one launch) with the integrator as their epilogue
            state["ts"] = ts + 1
            return
        if brun is not None:
            brun._force_rows(ts)       # one launch where nothing is in flight to hide, else interior | halo | boundary
        else:
            ctx.compute_forces_overlapped(ts, arrays(), nl.domain)
        if train is not None and ts % args.train_period == 0:
            train(timed)
        if brun is not None:
            brun._integrate()          # integrator + the next step's halo messages in one launch
        else:
            nve.step()
        state["ts"] = ts + 1

    # overflow guard: NN must hold every neighbor within r_cut (check_nlist semantics)
    mc = torch.zeros(1, dtype=torch.int32, device=dev)
    htf.ops.build_pair_vectors(sysm.pos, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, args.rcut, NN, max_count=mc)
    max_kept = int(mc.item())
    if max_kept > NN:
        raise SystemExit("NN=%d too small: a particle has %d neighbors within r_cut" % (NN, max_kept))

    # untimed relaxation: the jittered lattice has a few overlapping pairs; cap the force and
    # rescale velocities to kT = 1 until it is an equilibrium liquid, then run plain NVE.
    for _ in range(args.equil):
        ts = state["ts"]
        nl.compute(ts)
        state["builds"] = nl.n_builds
        ctx.compute_forces_overlapped(ts, arrays(), nl.domain)
        f3 = sysm.force[:, :3]
        fm = f3.norm(dim=1, keepdim=True).clamp_min(1e-12)
        f3.mul_(torch.clamp(200.0 / fm, max=1.0))
        nve.step()
        v3 = sysm.vel[:, :3]
        v3.mul_(torch.sqrt(1.0 / ((v3 * v3).sum() / (3.0 * N))))
        state["ts"] = ts + 1

    def builds_now():
        return nl.n_builds + nl.device_builds()

    def timed_window():
        """EXACTLY args.steps steps between barrier + synchronize on both sides; max over ranks."""
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(True)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    # second half of the equilibration: plain NVE, the loop that is timed below.  (The relaxation above is a chain of small
    # torch ops with the GPU mostly idle; the first ~40 steps after it ran 8 % slower than the rest, kernels included, until
    # the clocks had followed the load.)
    for _ in range(args.settle):
        step()
    for _ in range(args.warmup):
        step()
    # kernel durations: hipEvents around every PROF_EVERY-th htf_compute_forces batch of the timed region
    # (an odd period, so that with slabs interior and boundary launches are sampled alike); bracketing every
    # launch costs the 0.1 ms step about 8 %
    ctx.profile_enable(PROF_EVERY)
    batches_per_step = 2 if (nl.domain is not None and world > 1 and nl.domain.n_interior > 0) else 1
    # `value` = median over the windows: a 20-step window is 2 ms and holds one to three neighbor-list
    # rebuilds (~0.25 ms each), so a single short window swings by +-10 % with where the rebuilds fall
    n_windows = args.windows if args.windows > 0 else (5 if args.steps <= 50 else 1)
    builds0 = builds_now()
    windows, window_prof = [], []
    for _ in range(n_windows):
        windows.append(timed_window())
        window_prof.append(ctx.profile_read())  # (build ms, eval ms, bracketed calls) of this window; resets
    # the kernel durations are those of the window `value` is taken from (the median one)
    elapsed = float(np.median(windows))
    build_ms, eval_ms, ncalls = window_prof[int(np.argsort(windows)[len(windows) // 2])]
    ctx.profile_enable(False)
    rebuilds = (builds_now() - builds0) / n_windows  # per window of args.steps steps
    # clocks / power cap WHILE the loop runs: an extra, untimed stretch of steps with the sysfs read in its middle (behind a
    # synchronize the part has already clocked down: 158 MHz)
    for _ in range(30):
        step()
    gpu_now = gpu_state(dev.index or 0)
    for _ in range(30):
        step()

    # With slabs: where a step's time goes on rank 0, measured AFTER the timed windows (never part of `value`): host time of
    # each phase as the loop enqueues it, and the same phases with the device drained after each (GPU-inclusive).  On real
    # multi-GPU hardware this is what tells a slow halo from a slow host loop.
    phases = None
    if world > 1:
        def phase_pass(drain):
            acc = {"nlist_check_and_halo_post": 0.0, "forces_interior_halo_wait_boundary": 0.0, "integrate": 0.0}
            n = 0
            for _ in range(20):
                ts = state["ts"]
                b_before = nl.n_builds
                t0 = time.perf_counter()
                nl.compute(ts)
                state["builds"] = nl.n_builds
                if drain:
                    torch.cuda.synchronize()
                t1 = time.perf_counter()
                ctx.compute_forces_overlapped(ts, arrays(), nl.domain)
                if drain:
                    torch.cuda.synchronize()
                t2 = time.perf_counter()
                nve.step()
                if drain:
                    torch.cuda.synchronize()
                t3 = time.perf_counter()
                state["ts"] = ts + 1
                if nl.n_builds == b_before:  # steps without a rebuild: the common case, reported; rebuild steps are in `value`
                    acc["nlist_check_and_halo_post"] += t1 - t0
                    acc["forces_interior_halo_wait_boundary"] += t2 - t1
                    acc["integrate"] += t3 - t2
                    n += 1
            return {k: v / max(n, 1) * 1e6 for k, v in acc.items()}
        torch.cuda.synchronize()
        dist.barrier()
        phases = {"host_enqueue_us": phase_pass(False), "drained_after_each_phase_us": phase_pass(True),
                  "note": "rank 0, mean over the steps without a rebuild of a 20-step pass; untimed diagnostics"}
        torch.cuda.synchronize()
        dist.barrier()

    # sanity: the run must still be a valid simulation
    f = sysm.force
    assert bool(torch.isfinite(f).all()), "non-finite forces"
    n_now = nl.domain.n_local if brick else sysm.N     # (inert rows carry zero force and zero velocity)
    e_per_particle = float(f[:, 3].double().sum().item()) / n_now
    kT_final = float((sysm.vel[:, :3].double() ** 2).sum().item()) / (3.0 * n_now)

    n_entries = int(nl.n_neigh.long().sum().item())
    eval_b, build_b, integ_b = algorithmic_bytes(N, NN, n_entries, N + sysm.n_ghost, s4)
    # per STEP (with slabs a step is two row ranges = two launches of each kernel; the
    # algorithmic bytes below are per step as well)
    # per STEP = mean bracketed batch x batches per step; under slabs a rebuild step is ONE whole-range batch
    # (the rebuild's own exchange is blocking), every other step two (interior rows, boundary rows)
    batches_per_step = (batches_per_step * args.steps - (rebuilds if batches_per_step == 2 else 0)) / max(args.steps, 1)
    eval_avg_s = eval_ms / ncalls * batches_per_step * 1e-3 if ncalls else 0.0
    build_avg_s = build_ms / ncalls * batches_per_step * 1e-3 if ncalls else 0.0
    if one_kernel:
        # its own compulsory traffic only: the build's bytes + the force write (the evaluator's
        # N*NN*16 re-read of SURVEY 8(d) no longer happens and is NOT credited)
        be_b = build_b + N * s4
        kern = {"build_eval_forces": {"avg_us": eval_avg_s * 1e6, "algorithmic_bytes": be_b,
                                      "GBps": be_b / eval_avg_s / 1e9 if eval_avg_s > 0 else None,
                                      # SURVEY 8(d) would credit this launch with the build's AND the evaluator's bytes
                                      "contract_GBps": (build_b + eval_b) / eval_avg_s / 1e9 if eval_avg_s > 0 else None,
                                      "what": "pair-vector build with the evaluator as its epilogue: the [N,NN,4] "
                                              "tensor is written once (bit-identical) and not re-read"}}
        dom = "build_eval_forces"
    else:
        kern = {
            "eval_forces": {"avg_us": eval_avg_s * 1e6, "algorithmic_bytes": eval_b,
                            "GBps": eval_b / eval_avg_s / 1e9 if eval_avg_s > 0 else None},
            "build_pair_vectors": {"avg_us": build_avg_s * 1e6, "algorithmic_bytes": build_b,
                                   "GBps": build_b / build_avg_s / 1e9 if build_avg_s > 0 else None},
        }
        dom = "build_pair_vectors" if build_avg_s > eval_avg_s else "eval_forces"
    mfma = args.workload in ("mlp", "mlp-fp32", "mlp-bf16", "mlp-split")
    if train is not None and train_events:
        state["train_n"] = len(train_events)
        state["train_s"] = sum(a.elapsed_time(b) for a, b in train_events) * 1e-3
        kern["train_step"] = {"avg_ms": state["train_s"] / state["train_n"] * 1e3, "count": state["train_n"],
                              "stream": "main" if args.sync_train else "second stream, beside the following MD steps",
                              "period": args.train_period, "loss": float(opt_state[20]),
                              "what": "pair-MLP prediction + loss-gradient sweep + all-reduce + Adam + image refresh"}
    if mfma:
        # Flops of the slots the kernel EXECUTES: a row's live slots are contiguous, 32-slot tiles that hold
        # only padding are skipped (wave-uniform ballot), so the dense N x NN count -- what the reference's
        # graph would do -- overstates the work; it is reported beside as `dense_TFLOPs`.
        per_slot = 4.0 * (32 * 64 + 64 * 64 + 64)
        pv_now = ctx.nlist_buffer(sysm.N, dev)
        live = (pv_now[:, :, :3] != 0).any(dim=2)
        # round 4: the evaluator compacts live pairs across the rows of a wave before they become 32-pair tiles (wave w of the
        # 2 x 256 persistent workgroups of four takes rows w, w + nwaves, ...): it executes ceil(live pairs of the wave / 32)
        # tiles -- 389 k at C3 where the rows' own 32-slot tiles with a live slot number 411-424 k
        nwaves = 4 * min(2 * torch.cuda.get_device_properties(dev).multi_processor_count, (sysm.N + 3) // 4)
        per_row = live.sum(dim=1)
        pad = (-sysm.N) % nwaves
        per_wave = torch.cat([per_row, per_row.new_zeros(pad)]).reshape(-1, nwaves).sum(dim=0)
        tiles = int(((per_wave + 31) // 32).sum().item())
        row_tiles = int(live.reshape(sysm.N, NN // 32, 32).any(dim=2).sum().item()) if NN % 32 == 0 else sysm.N * ((NN + 31) // 32)
        flops = per_slot * 32.0 * tiles
        # split: every algorithmic multiply-add is six bf16 MFMA multiply-adds, so the algorithmic rate is
        # priced against a sixth of the dense bf16 peak
        # split16 (the default): three fp16 MFMA multiply-adds per algorithmic one -> a third of the dense fp16 / bf16 peak
        peak = {"mlp-bf16": 2500.0, "mlp-split": 2500.0 / 6.0, "mlp": 2500.0 / 3.0}.get(args.workload, 157.3)
        ach = flops / eval_avg_s / 1e12
        roof = {"bound": "mfma", "kernel": "eval_forces(pair_mlp)", "achieved": ach, "peak": peak,
                "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
                "executed_tiles_of_32_pairs": tiles, "row_tiles_with_a_live_slot": row_tiles, "dense_tiles": sysm.N * ((NN + 31) // 32),
                "dense_TFLOPs": per_slot * N * NN / eval_avg_s / 1e12}
        if args.workload == "mlp-split":
            roof["peak_note"] = "dense bf16 MFMA peak / 6 partial products per fp32-level multiply (fp32 MFMA peak: 157.3)"
        if args.workload == "mlp":
            roof["peak_note"] = ("dense fp16 MFMA peak (2.5 PFLOP/s) / 3 partial products per fp32-level multiply; against the fp32 "
                                 "MFMA peak (157.3 TFLOP/s), which the fp32-operand evaluator is priced on, frac would be %.2f" % (ach / 157.3))
    else:
        ach = kern[dom]["GBps"]
        # `frac` prices the contract's ALGORITHMIC bytes (the padded [N, NN, 4] tensor counts in full) over this run's
        # launch durations.  `traffic` is null: HBM counters cannot be read from inside the process.  What the memory system
        # itself moved is in `reference_counters` below -- counters of a SEPARATE rocprofv3 --pmc run of this command, committed
        # under profiles/ -- and is the number to lead with: the kernel rewrites a row's zero tail only where the row shrank, so
        # it moves fewer bytes than the contract counts.
        roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "frac_is": "algorithmic (contract) bytes / launch duration / 8 TB/s", "traffic": None}
        refc = None
        try:
            if args.cells != 32 or args.workload != "lj":
                raise KeyError("PMC passes were collected for the default workload at the default size")
            pmc_file = next(f for f in ("r05_bench_lj_pmc_hbm.json", "r04_bench_lj_pmc_hbm.json", "r03_bench_lj_pmc_hbm.json", "r02_bench_lj_pmc_hbm.json", "r01_bench_lj_pmc_hbm.json")
                            if os.path.exists(os.path.join(ROOT, "profiles", f)))
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
            want = {"build_pair_vectors": ("build_pair_vectors_kernel",), "eval_forces": ("eval_pair_kernel<1,",),
                    # the one-kernel step: four rows per wave with merged tails (default) or the two-row form
                    "build_eval_forces": ("fused_forces_tails_kernel<1, true", "fused_forces_rows2_kernel<1, true")}[dom]
            key = next(k for w in want for k in pmc["FETCH_SIZE"] if w in k and k in pmc["WRITE_SIZE"])
            rd, wr = pmc["FETCH_SIZE"][key]["avg_KiB"], pmc["WRITE_SIZE"][key]["avg_KiB"]
            # gfx950: every fabric-side read request of the L2 is 128 B (TCC_EA0_RDREQ_32B = TCC_BUBBLE = 0) and FETCH_SIZE
            # tallies it at 64 B.  Calibrated on known byte counts in THIS kernel's access patterns (tools/fetch_calib.hip,
            # profiles/r03_fetch_calib.json): 4 B/lane index streams, clamped index rows, 16-B gathers from an L2-resident
            # table and 16 B/lane streams all read known / FETCH_SIZE = 1.99-2.00; WRITE_SIZE is exact (0.993-0.998) for
            # the nontemporal 16-B stores, full rows and live-slot rows alike.
            corr = 2.0
            tb = (rd * corr + wr) * 1024.0
            tg = tb / (kern[dom]["avg_us"] * 1e-6) / 1e9
            refc = {"what": "HBM bytes of the dominant kernel per launch, from committed counters of a separate run of this same command -- NOT measured by this run",
                    "source": "profiles/%s (FETCH_SIZE x%g + WRITE_SIZE; factor from tools/fetch_calib.hip, profiles/r03_fetch_calib.json)" % (pmc_file, corr),
                    "kernel": dom, "traffic_bytes_per_launch": tb,
                    # those bytes over THIS run's launch duration: the rate the memory system ran at
                    "traffic_GBps": tg, "traffic_frac": tg / HBM_PEAK_GBS,
                    "traffic_frac_of_achievable": tg / 6290.0,  # 6.29 TB/s: the float4 copy ceiling this part sustains (MI355X_MICROARCH.md)
                    "traffic_over_algorithmic_bytes": tb / kern[dom]["algorithmic_bytes"]}
        except (OSError, KeyError, ValueError, StopIteration):
            pass

    ms_per_step = elapsed / args.steps * 1e3
    # algorithmic bytes of the kernels this run actually launches per step (the one-kernel step does not
    # re-read the tensor, so the evaluator's bytes are not counted for it)
    step_bytes = (build_b + N * s4 if one_kernel else eval_b + build_b) + integ_b
    out = {
        "metric": "MD steps/sec + achieved HBM GB/s, %d particles NN=%d (%s)" % (
            n_global, NN, "131k-particle box of the BASELINE metric" if n_global == 131072 else
            ("config 5 block layout: one 131072-particle block per rank" if not strong and world > 1 else "non-default size")),
        # MD steps per second of the GLOBAL system (every rank advances its share of every step)
        "value": args.steps / elapsed,
        "unit": "steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong" if (strong or world == 1) else "weak", "vs_baseline": None,
        "particle_steps_per_s": n_global * args.steps / elapsed,
        "windows_ms_per_step": [w / args.steps * 1e3 for w in windows],
        # mean bracketed htf_compute_forces batch (build + eval kernels) per window; `kernels` / `roofline` quote the median window's
        "windows_batch_us": [(b + e) / n * 1e3 if n else None for b, e, n in window_prof],
        "value_is": "median of %d timed windows of %d steps each" % (n_windows, args.steps) if n_windows > 1 else "one timed window",
        "dtype": {"mlp-bf16": "bf16 operands, f32 accumulation",
                  "mlp-split": "f32 (each operand split exactly into 3 bf16 parts, 6 partial products, f32 accumulation)",
                  "mlp": "f32 (each operand as hi + lo in fp16, 2^-22; 3 partial products on the fp16 MFMA, f32 accumulation)"
                  }.get(args.workload, "f32" if not args.f64 else "f32 arithmetic on an f64 wire (HOOMD in double precision)"),
        "data": "synthetic",
        # the stated fp32 tolerance of north_star, as the parity tests assert it (tests/test_gpu_parity.py)
        "tolerance": {"pair_vectors": "bit-exact",
                      "forces_energy_virial": "|d| <= 1e-5 + 2e-5 |ref| against the fp64 oracle on the same fp32 inputs (SURVEY 8(c)); "
                                              "+ 2e-6 sum_j |f_ij| on rows whose pair forces cancel (an equilibrated liquid: 300 -> 10), "
                                              "where any fp32 row sum, TensorFlow's included, misses the strict bound",
                      "pair_mlp": "2e-5 + 5e-5 |ref| (fp32, split16 and split operands alike)"},
        "config": {"workload": ("%s: " + ("sc %d^3" if args.lattice == "sc" else "fcc %d^3x4")
                                + " = %d particles %s, rho 0.8442, r_cut %.1f, r_buff %.1f, NN %d, dt %g")
                               % ("C5b (pair-MLP MD + force-matching step every %d steps vs LJ labels)" % args.train_period
                                  if args.workload == "mlp-train" else ("C2-WCA" if args.workload == "wca" and n_block == 32768 else "C3-" + args.workload.upper()),
                                  args.cells, n_block, "in all, cut into %d slabs" % world if strong else ("per GPU" if world > 1 else "on one GPU"),
                                  args.rcut, args.rbuff, NN, args.dt),
                   "preparation": "untimed: %d relaxation steps (force cap + velocity rescale to kT = 1), %d plain NVE steps, then the %d warmup steps"
                                  % (args.equil, args.settle, args.warmup),
                   "global_particles": n_global, "particles_rank0": N, "parallelism": "dd%dx%dx%d" % grid if world > 1 else "dd1x1x1",
                   "nlist_rebuilds_per_window": rebuilds, "max_neighbors_within_rcut": max_kept,
                   "nlist_decision": ("device: distance check all-reduced on the device, read one check late (DeferredRebuildRule), "
                                      "dangerous builds: %d" % nl.dangerous_builds) if nl.device_decision and world > 1 and not args.sort
                                     else "device (gated rebuild kernels, no read-back in the step loop)" if nl.device_decision and world == 1 and not args.sort
                                     else "host (distance check read back every %d steps%s)" % (args.check_period, ", all-reduced over ranks" if world > 1 else ""),
                   "halo": None if world == 1 else {"ghosts_rank0": nl.domain.n_ghosts if brick else sysm.n_ghost,
                                                    "migrated_rank0": nl.domain.n_migrated,
                                                    "interior_rows_rank0": nl.domain.n_interior,
                                                    "replan_every": getattr(nl.domain, "replan_every", 1),
                                                    "rebuilds_without_a_replan_rank0": getattr(nl.domain, "n_light", 0),
                                                    "domain": ("BrickDomain: fixed-capacity arrays (%d rows + %d ghost rows on rank 0), inert "
                                                               "rows, no read-back in a rebuild" % (sysm.N, sysm.n_ghost)) if brick
                                                              else "SlabDomain (variable-length arrays, host-planned rebuild)",
                                                    "transport": ("RCCL: the library's own communicator and halo stream (csrc/halo.hip)" if nl.domain.transport == "native"
                                                                  else (E.backend if E.backend != "nccl" else "RCCL (torch.distributed nccl backend)")),
                                                    "transport_note": getattr(nl.domain, "transport_note", None),
                                                    "exchange": "forward ghost positions, grouped send/recv, every step",
                                                    "step_phases_rank0": phases}},
        # sum over ranks of the algorithmic bytes a step moves (rank 0's count x ranks) / step time
        "gpu_state": gpu_now,
        "hbm_GBps_full_step": world * step_bytes / (elapsed / args.steps) / 1e9,
        "hbm_frac_full_step": step_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
        "energy_per_particle": e_per_particle, "kT_final": kT_final,
        "kernels": kern,
        "roofline": roof,
    }
    out["config"]["integrator"] = ("the stand-in's leapfrog update as the EPILOGUE of the force kernel (one launch per plain step; positions "
                                   "ping-pong between two arrays; same bits as the separate htfs_nve_step launch)" if state["fused_step"]
                                   else "htfs_nve_step, a launch of its own behind the force kernel")
    if not mfma and refc is not None:
        out["reference_counters"] = refc
    # ---- extras, reported separately and never mixed into `roofline`: the same MD (a) with the
    # reference's two-kernel dataflow (build kernel, then evaluator kernel re-reading the tensor)
    # and (b) with the pair vectors kept in registers and NO tensor.
    def run_variant(mode, pot_v=None):
        ctx_v = htf.Context(r_cut=args.rcut, nneighs=NN, scalar_dtype=sdt, max_n=sysm.N, fused=mode)
        ctx_v.set_potential(pot if pot_v is None else pot_v)
        state["arr_v"] = None

        def step_v():
            ts = state["ts"]
            nl.compute(ts)
            if nl.n_builds != state["builds"] or state["arr_v"] is None:
                state["arr_v"] = ctx_v.make_arrays(sysm.pos, sysm.N, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, sysm.force)
                state["builds"] = nl.n_builds
            ctx_v.compute_forces_overlapped(ts, state["arr_v"], nl.domain)
            nve.step()
            state["ts"] = ts + 1

        for _ in range(args.warmup):
            step_v()
        ctx_v.profile_enable(PROF_EVERY)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_v()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        el = time.perf_counter() - t0
        b_ms, e_ms, nc = ctx_v.profile_read()
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        per = 1e-3 * batches_per_step / max(nc, 1)
        return el, (b_ms * per if nc else 0.0), (e_ms * per if nc else 0.0)

    if not args.no_fused and closed_form and train is None:
        if one_kernel:
            el, b_s, e_s = run_variant(0)
            out["two_kernel_variant"] = {
                "note": "htf_config.fused = 0: build kernel, then evaluator kernel re-reading the tensor (SURVEY 8(d) dataflow)",
                "value": args.steps / el, "unit": "steps/s", "ms_per_step": el / args.steps * 1e3,
                "build_pair_vectors": {"avg_us": b_s * 1e6, "algorithmic_bytes": build_b, "GBps": build_b / b_s / 1e9 if b_s > 0 else None},
                "eval_forces": {"avg_us": e_s * 1e6, "algorithmic_bytes": eval_b, "GBps": eval_b / e_s / 1e9 if e_s > 0 else None}}
        el, _, f_s = run_variant(1)
        fb = sysm.N * 8 + n_entries * 4 + (sysm.N + sysm.n_ghost) * s4 + sysm.N * s4
        out["fused_variant"] = {
            "note": "pair vectors evaluated in registers (htf_config.fused=1); the [N,NN,4] tensor is not materialised",
            "value": args.steps / el, "unit": "steps/s", "ms_per_step": el / args.steps * 1e3,
            "kernel_avg_us": f_s * 1e6, "algorithmic_bytes": fb, "GBps": fb / f_s / 1e9 if f_s > 0 else None,
            "energy_per_particle": float(sysm.force[:, 3].double().sum().item()) / sysm.N}
    # (b') pair-MLP: the same network and weights through the other two fp32-level evaluators -- fp32 operands on the fp32
    # MFMA (v_mfma_f32_32x32x2_f32) and the exact three-part bf16 split (3.3a') -- on the same pair vectors;
    # all three are held to the same tolerances against the fp64 oracle (test_pair_mlp_split_operands, test_pair_mlp_fp32_mfma)
    if not args.no_fused and args.workload == "mlp":
        from hoomd_tf_amd.initializers import mlp_params
        pv_now = ctx.nlist_buffer(sysm.N, dev)
        fa = htf.ops.eval_forces(pot, pv_now)
        for key, prec, what in (("fp32_variant", "fp32", "fp32 operands on v_mfma_f32_32x32x2_f32 (exact fp32 products)"),
                                ("split_variant", "split", "fp32 operands split exactly into 3 bf16 parts, 6 partial products per "
                                                           "multiply on v_mfma_f32_32x32x16_bf16")):
            pot_v = htf.Potential.pair_mlp(mlp_params(seed=3), 0.0, 3.0, activation="tanh", precision=prec)
            fv = htf.ops.eval_forces(pot_v, pv_now)
            rel = float((fa - fv).abs().max() / fa.abs().max())
            el, _, e_s = run_variant(0, pot_v)
            out[key] = {
                "note": "precision=%r: %s, fp32 accumulation; forces agree with the default (split16) evaluator on the same "
                        "pair vectors to max|dF|/max|F| = %.1e" % (prec, what, rel),
                "value": args.steps / el, "unit": "steps/s", "ms_per_step": el / args.steps * 1e3,
                "eval_forces_avg_us": e_s * 1e6, "executed_TFLOPs": flops / e_s / 1e12 if e_s > 0 else None,
                "max_rel_force_difference_vs_default": rel}
    # (c) the same MD through the plugin surface a user touches: an htf.SimModel written op by op as in the
    # reference's LJModel (build_examples.py:67-77), htf.tfcompute(model).attach(nlist, r_cut), and the
    # stand-in's System::run loop.  tfcompute traces the model on its first step and replays it as the
    # same one-kernel step afterwards.
    if not args.no_fused and args.workload == "lj" and world == 1:
        class LJModel(htf.SimModel):
            def compute(self, nlist, positions, box):
                rinv = htf.nlist_rinv(nlist)
                inv_r6 = rinv**6
                p_energy = 4.0 / 2.0 * (inv_r6 * inv_r6 - inv_r6)
                energy = htf.reduce_sum(p_energy, axis=1)
                return htf.compute_nlist_forces(nlist, energy)

        sim = standin.Simulation(sysm)
        sim.integrate_nve(args.dt)
        tfc = htf.tfcompute(LJModel(NN))
        cell = sim.nlist_cell(r_buff=args.rbuff, check_period=args.check_period)
        tfc.attach(cell, r_cut=args.rcut)
        sim.run(args.warmup + 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sim.run(args.steps)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        out["tfcompute_variant"] = {
            "note": "LJModel(htf.SimModel) -> htf.tfcompute(model).attach(nlist, r_cut) -> run(steps): the reference's user-facing "
                    "path; traced on the first step, replayed as the one-kernel step",
            "value": args.steps / el, "unit": "steps/s", "ms_per_step": el / args.steps * 1e3,
            "replayed": tfc._plan is not None,
            "energy_per_particle": float(tfc.force[:, 3].double().sum().item()) / sysm.N}
        # what Simulation.run(n) does BY ITSELF on a long run: its first steps timed stepwise and replayed
        # from a hipGraph, the faster kept (sim.graph_choice); then 400 steps under that choice
        sim.run(320)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sim.run(400)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ch = dict(getattr(sim, "graph_choice", None) or {})
        ch.pop("key", None)
        out["tfcompute_variant"]["auto_run"] = {"note": "sim.run(n) with graph=None: stepwise or replayed, chosen by timing the run's own first steps both ways",
                                                "choice": ch, "value": 400 / el, "unit": "steps/s"}
    # (b') the same step loop replayed from a hipGraph: one check period of steps (distance check, gated rebuild, force
    # kernel, integrator) captured once, one launch per period afterwards.  Kernel durations cannot be bracketed inside a
    # replay, so this is reported beside `value`, not as it.
    if (not args.no_fused and args.workload in ("lj", "wca") and world == 1 and train is None and nl._device_ok()
            and nl._stat is not None and args.steps % args.check_period == 0):
        try:  # last GPU work of the run, and optional: a failed capture must not cost the line
            cyc = args.check_period
            nl.build()  # the tfcompute variant above moved the particles under a list of its own
            state["builds"] = nl.n_builds
            while state["ts"] % cyc != 0:
                step()
            torch.cuda.synchronize()
            nl._poll_overflow()
            b_before = nl.n_builds
            g = torch.cuda.CUDAGraph()
            ts0 = state["ts"]
            nl._capturing = True
            try:
                with torch.cuda.graph(g):
                    for _ in range(cyc):
                        step()
            finally:
                nl._capturing = False
                state["ts"] = ts0
            assert nl.n_builds == b_before

            def graph_window():
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps // cyc):
                    g.replay()
                torch.cuda.synchronize()
                state["ts"] += args.steps
                return time.perf_counter() - t0

            graph_window()
            gw = [graph_window() for _ in range(n_windows)]
            nl.mark_check_enqueued()
            torch.cuda.synchronize()
            nl._poll_overflow()
            assert bool(torch.isfinite(sysm.force).all())
            out["graph_variant"] = {
                "note": "the step loop replayed from a hipGraph of %d steps (one check period); same kernels, same decisions on the device" % cyc,
                "value": args.steps / float(np.median(gw)), "unit": "steps/s", "ms_per_step": float(np.median(gw)) / args.steps * 1e3,
                "windows_ms_per_step": [w / args.steps * 1e3 for w in gw]}
        except Exception as e:  # noqa: BLE001
            nl._capturing = False
            out["graph_variant"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(sysm, nl, args)
    elif rank == 0:
        out["cpu_baseline"] = None
    if keep_live and brick and closed_form and train is None:
        # benchlib.multirank.guarded_section continues on this system (the native transport's self-test, the replayed step)
        E.live = {"args": args, "sysm": sysm, "nl": nl, "ctx": ctx, "nve": nve, "brun": brun, "state": state, "step": step}
    return out


if __name__ == "__main__":
    main()
