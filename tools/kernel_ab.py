#!/usr/bin/env python3
"""A/B timing of the two hot kernels on the bench system (interleaved rounds, one process).
Usage: HTF_BUILD_VARIANT=k python tools/kernel_ab.py   (experiment harness, not a test)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hoomd_tf_amd as htf  # noqa: E402
from hoomd_tf_amd import standin  # noqa: E402

dev = torch.device("cuda:0")
pos, L, a = standin.fcc_positions(32, 0.8442)
rng = np.random.default_rng(3)
pos = pos + 0.03 * a * rng.standard_normal(pos.shape)
pos -= np.round(pos / L) * L
sysm = standin.System(pos, L, dtype=torch.float32, device=dev)
nl = standin.CellNlist(sysm, r_cut=3.0, r_buff=0.4)
nl.build()
N, NN = sysm.N, 128
out = torch.empty((N, NN, 4), device=dev)
force = torch.empty((N, 4), device=dev)
pot = htf.Potential.lj()


def timeit(fn, reps=30):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def build():
    htf.ops.build_pair_vectors(sysm.pos, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, 3.0, NN, out=out)


def ev():
    htf.ops.eval_forces(pot, out, out=force)


def both():
    build()
    ev()


for r in range(3):
    print("variant %s round %d: build %.1f us  eval %.1f us  build+eval %.1f us" % (
        os.environ.get("HTF_BUILD_VARIANT", "1"), r, timeit(build), timeit(ev), timeit(both)))
