"""`run_md`: one MD workload (lj | wca | mlp | mlp-split | mlp-bf16 | mlp-fp32 | mlp-train) on this job's ranks -> the JSON record: the
headline line of bench.py (C3-LJ), its pair-MLP sub-record, C2, C5b, and the multi-rank runs."""
import json
import os
import sys
import time

import numpy as np
import torch

from .common import HBM_PEAK_GBS, PROF_EVERY, ROOT, algorithmic_bytes, gpu_state, make_potential  # noqa: F401
from .cpu import cpu_baseline


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
        if state["fused_step"]:
            # the launch also does the integrator's work (velocity read + write, new position write: SURVEY 8(d)'s N * 4 * s * 6 minus
            # the position and force reads it shares with the force step) -- counted, or `frac` would price a longer kernel on fewer bytes
            be_b += N * s4 * 3
        kern = {"build_eval_forces": {"avg_us": eval_avg_s * 1e6, "algorithmic_bytes": be_b,
                                      "GBps": be_b / eval_avg_s / 1e9 if eval_avg_s > 0 else None,
                                      # SURVEY 8(d) would credit this launch with the build's AND the evaluator's bytes
                                      "contract_GBps": (build_b + eval_b) / eval_avg_s / 1e9 if eval_avg_s > 0 else None,
                                      "what": "pair-vector build with the evaluator as its epilogue: the [N,NN,4] "
                                              "tensor is written once (bit-identical) and not re-read"
                                              + ("; the leapfrog update of the row rides on the same launch" if state["fused_step"] else "")}}
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
            pmc_file = next(f for f in ("r06_bench_lj_pmc_hbm.json", "r05_bench_lj_pmc_hbm.json", "r04_bench_lj_pmc_hbm.json", "r03_bench_lj_pmc_hbm.json",
                                        "r02_bench_lj_pmc_hbm.json", "r01_bench_lj_pmc_hbm.json")
                            if os.path.exists(os.path.join(ROOT, "profiles", f)))
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
            want = {"build_pair_vectors": ("build_pair_vectors_kernel",), "eval_forces": ("eval_pair_kernel<1,",),
                    # the one-kernel step: four rows per wave with merged tails (default) or the two-row form
                    "build_eval_forces": ("fused_forces_tails_kernel<1, true", "fused_forces_rows2_kernel<1, true")}[dom]
            keys = [k for w in want for k in pmc["FETCH_SIZE"] if w in k and k in pmc["WRITE_SIZE"]]
            # (round 6: the launch that carries the integrator as its epilogue is its own instantiation, "..., float, 1>")
            key = sorted(enumerate(keys), key=lambda ik: (ik[1].rstrip().endswith(", 1>") != bool(state.get("fused_step")), ik[0]))[0][1]
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
