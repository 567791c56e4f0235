"""`--workload eds`: BASELINE configs[3] (C4) -- EDS bias on a soft RDF collective variable, 262 144 particles, one sweep per step."""
import json
import os
import sys
import time

import numpy as np
import torch

from .common import HBM_PEAK_GBS, PROF_EVERY, ROOT, algorithmic_bytes, gpu_state, make_potential  # noqa: F401
from .cpu import cpu_baseline  # noqa: F401


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
