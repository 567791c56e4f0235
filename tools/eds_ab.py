#!/usr/bin/env python3
"""Timing of the C4 sweep variants at 262144 x 128 (experiment harness)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hoomd_tf_amd as htf
from hoomd_tf_amd import standin
dev = torch.device("cuda:0")
pos, L, a = standin.sc_positions(64, 0.8442)
pos = pos + 0.05 * a * np.random.default_rng(4).standard_normal(pos.shape); pos -= np.round(pos / L) * L
sysm = standin.System(pos, L, dtype=torch.float32, device=dev)
nl = standin.CellNlist(sysm, r_cut=3.0, r_buff=0.4); nl.build()
N, NN = sysm.N, 128
pv = torch.zeros((N, NN, 4), device=dev); fa = torch.empty((N, 4), device=dev); fb = torch.empty((N, 4), device=dev)
n = htf.ops.num_partials(N, NN); partials = torch.empty(n, device=dev); hist = torch.zeros(102, dtype=torch.int32, device=dev)
lj, ga = htf.Potential.lj(), htf.Potential.gauss(1.1, 0.05, 1.0)
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
build = lambda: htf.ops.build_pair_vectors(sysm.pos, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, 3.0, NN, out=pv)
print("n_neigh mean", float(nl.n_neigh.float().mean()), "pitch", nl.pitch)
print("build        %.1f us" % timeit(build))
print("eval lj      %.1f us" % timeit(lambda: htf.ops.eval_forces(lj, pv, out=fa)))
print("eval2        %.1f us" % timeit(lambda: htf.ops.eval_forces2(lj, ga, pv, out_a=fa, out_b=fb, partials=partials)))
print("eval2+rdf    %.1f us" % timeit(lambda: htf.ops.eval_forces2(lj, ga, pv, out_a=fa, out_b=fb, partials=partials, rdf=(0.0, 3.5, hist))))
print("rdf alone    %.1f us" % timeit(lambda: htf.compute_rdf(pv, [0, 3.5])))
print("build+eval2+rdf %.1f us" % timeit(lambda: (build(), htf.ops.eval_forces2(lj, ga, pv, out_a=fa, out_b=fb, partials=partials, rdf=(0.0, 3.5, hist)))))
