#!/usr/bin/env python3
"""Timing of the C4 one-kernel sweep (htf_build_eval_forces2) on the C4 system, piece by piece:
   python tools/fused2_ab.py     (HTF_FUSED2_COMPACT=0|1, HTF_FUSED2_GRID=<workgroups>)
tensor written or not, RDF histogram on or off (experiment harness, not a test)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hoomd_tf_amd as htf  # noqa: E402
from hoomd_tf_amd import standin  # noqa: E402

dev = torch.device("cuda:0")
cells = int(os.environ.get("CELLS", "64"))
a = (1.0 / 0.8442) ** (1.0 / 3.0)
ijk = np.stack(np.meshgrid(*[np.arange(cells)] * 3, indexing="ij"), -1).reshape(-1, 3)
L = np.array([cells * a] * 3)
pos = (ijk + 0.5) * a - L / 2
rng = np.random.default_rng(4)
pos = pos + 0.05 * a * rng.standard_normal(pos.shape)
pos -= np.round(pos / L) * L
order = os.environ.get("ORDER", "lattice")  # particle order in memory: lattice (ijk, z fastest) | morton | cells | random
if order != "lattice":
    cw = float(os.environ.get("CW", "1.7"))
    c = np.floor((pos + L / 2) / cw).astype(np.int64)
    if order == "random":
        perm = rng.permutation(len(pos))
    elif order == "cells":
        nc = int(np.ceil(L[0] / cw))
        perm = np.argsort((c[:, 0] * nc + c[:, 1]) * nc + c[:, 2], kind="stable")
    else:
        def spread(v):
            out = np.zeros_like(v)
            for b in range(10):
                out |= ((v >> b) & 1) << (3 * b)
            return out
        perm = np.argsort(spread(c[:, 0]) << 2 | spread(c[:, 1]) << 1 | spread(c[:, 2]), kind="stable")
    pos = pos[perm]
sysm = standin.System(pos, L, dtype=torch.float32, device=dev)
nl = standin.CellNlist(sysm, r_cut=3.0, r_buff=0.4)
nl.build()
N, NN = sysm.N, 128
lj = htf.Potential.lj()
gauss = htf.Potential.gauss(1.1, 0.05) if hasattr(htf.Potential, "gauss") else None
if gauss is None:
    raise SystemExit("no Potential.gauss")
partials = torch.zeros(htf.ops.num_partials_fused(N), dtype=torch.float32, device=dev)
hist = torch.zeros(102, dtype=torch.int32, device=dev)
tensor = torch.empty((N, NN, 4), dtype=torch.float32, device=dev)
fa = torch.empty((N, 4), dtype=torch.float32, device=dev)
fb = torch.empty((N, 4), dtype=torch.float32, device=dev)


def run(with_tensor, with_rdf, reps=30):
    def call():
        htf.ops.build_eval_forces2(lj, gauss, sysm.pos, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, 3.0, NN, partials=partials,
                                   rdf=(0.0, 3.5, hist) if with_rdf else None, pair_vectors=tensor if with_tensor else None,
                                   out_a=fa, out_b=fb)
    call()
    torch.cuda.synchronize()
    best = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / reps * 1e3)
    return min(best)


for wt in (True, False):
    for wr in (True, False):
        print("%s order=%s tensor=%d rdf=%d: %.1f us" % (os.environ.get("TAG", ""), order, wt, wr, run(wt, wr)))
