#!/usr/bin/env python3
"""Timing of the one-kernel step (htf_compute_forces, fused = 2) on the C3 system for A/B runs of
library builds and launch knobs:  HTF_AMD_LIB=build_variants/libhtf_x.so HTF_FUSED_BLOCK=1024
python tools/fused_ab.py [--order lattice|sorted|shuffled] [--fused 2|1] [--relax 200]
(experiment harness, not a test)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hoomd_tf_amd as htf  # noqa: E402
from hoomd_tf_amd import standin  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--order", default="lattice", choices=["lattice", "sorted", "shuffled"])
ap.add_argument("--fused", type=int, default=2)
ap.add_argument("--relax", type=int, default=0, help="MD steps before timing (liquid instead of jittered lattice)")
ap.add_argument("--cells", type=int, default=32)
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--tag", default="")
ap.add_argument("--f64", action="store_true", help="HOOMD in double precision: fp64 positions and forces on the wire")
ap.add_argument("--potential", default="lj", choices=["lj", "wca"])
ap.add_argument("--digest", default="", help="print sha1 digests of the tensor and the forces; save the forces to this .npy")
args = ap.parse_args()

dev = torch.device("cuda:0")
pos, L, a = standin.fcc_positions(args.cells, 0.8442)
rng = np.random.default_rng(3)
pos = pos + 0.05 * a * rng.standard_normal(pos.shape)
pos -= np.round(pos / L) * L
if args.order == "shuffled":
    pos = pos[rng.permutation(len(pos))]
sysm = standin.System(pos, L, dtype=torch.float64 if args.f64 else torch.float32, device=dev)
sysm.randomize_velocities(kT=1.0, seed=3)
nl = standin.CellNlist(sysm, r_cut=3.0, r_buff=0.4, check_period=5, sort_particles=(args.order == "sorted"))
nl.build()
N, NN = sysm.N, 128
ctx = htf.Context(r_cut=3.0, nneighs=NN, max_n=N, fused=args.fused, scalar_dtype=torch.float64 if args.f64 else torch.float32)
ctx.set_potential(htf.Potential.lj() if args.potential == "lj" else htf.Potential.wca(1.0))
nve = standin.NVE(sysm, 0.005)


def arrays():
    return ctx.make_arrays(sysm.pos, sysm.N, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, sysm.force)


arr = arrays()
for ts in range(args.relax):
    nb = nl.n_builds
    nl.compute(ts)
    if nl.n_builds != nb:
        arr = arrays()
    ctx.compute_forces(ts, arr)
    f3 = sysm.force[:, :3]
    f3.mul_(torch.clamp(200.0 / f3.norm(dim=1, keepdim=True).clamp_min(1e-12), max=1.0))
    nve.step()
    v3 = sysm.vel[:, :3]
    v3.mul_(torch.sqrt(1.0 / ((v3 * v3).sum() / (3.0 * N))))
if args.relax:
    nl.build()
    arr = arrays()


def timeit(reps):
    ctx.compute_forces(0, arr)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ctx.compute_forces(0, arr)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


if args.digest:
    # bit-level fingerprint of what one call leaves behind: the [N, NN, 4] tensor and the forces
    import hashlib
    ctx.compute_forces(0, arr)
    torch.cuda.synchronize()
    t = ctx.nlist_buffer(N).cpu().numpy()
    f = sysm.force.cpu().numpy()
    print("digest tensor %s  forces %s  sum|F| %.6f" % (hashlib.sha1(t.tobytes()).hexdigest()[:16],
                                                     hashlib.sha1(f.tobytes()).hexdigest()[:16], float(np.abs(f).sum())))
    np.save(args.digest, f)
    if os.environ.get("HTF_DUMP_TENSOR"):
        np.save(os.environ["HTF_DUMP_TENSOR"], t)
ts = [timeit(args.reps) for _ in range(5)]
print("%-28s lib=%s block=%s rows=%s order=%s relax=%d fused=%d: %s  median %.1f us  (entries/row %.1f)" % (
    args.tag, os.path.basename(os.environ.get("HTF_AMD_LIB", "default")), os.environ.get("HTF_FUSED_BLOCK", "256"),
    os.environ.get("HTF_FUSED_ROWS", "2"), args.order, args.relax, args.fused,
    " ".join("%.1f" % t for t in ts), float(np.median(ts)), float(nl.n_neigh.float().mean())))
