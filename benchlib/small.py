"""BASELINE configs[0] and the reference's own published benchmark: `--workload c1 | ex01 | ref-lj256` (plumbing-sized systems, bound
by the host's enqueue; stepwise and replayed from a hipGraph)."""
import json
import os
import sys
import time

import numpy as np
import torch

from .common import HBM_PEAK_GBS, PROF_EVERY, ROOT, algorithmic_bytes, gpu_state, make_potential  # noqa: F401


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
