"""`cpu_baseline`: the C / OpenMP restatement of the same computeForces pass (oracle/htf_oracle_c.c) timed on the host's cores for a
bounded ~10 s.  The ONLY place of bench.py's default line that touches oracle/ -- a reported baseline, never the product."""
import json
import os
import sys
import time

import numpy as np
import torch

from .common import HBM_PEAK_GBS, PROF_EVERY, ROOT, algorithmic_bytes, gpu_state, make_potential  # noqa: F401


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
