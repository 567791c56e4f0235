#!/usr/bin/env python3
"""Timing of htf_build_pair_vectors alone on the C3 / C2 systems (experiment harness):
   HTF_BUILD_TAILS=0|2|4 python tools/build_ab.py [--f64] [--cells 32]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hoomd_tf_amd as htf  # noqa: E402
from hoomd_tf_amd import standin  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--f64", action="store_true")
ap.add_argument("--cells", type=int, default=32)
a = ap.parse_args()
dev = torch.device("cuda:0")
pos, L, lat = standin.fcc_positions(a.cells, 0.8442)
rng = np.random.default_rng(3)
pos = pos + 0.05 * lat * rng.standard_normal(pos.shape)
pos -= np.round(pos / L) * L
sysm = standin.System(pos, L, dtype=torch.float64 if a.f64 else torch.float32, device=dev)
nl = standin.CellNlist(sysm, r_cut=3.0, r_buff=0.4)
nl.build()
N, NN = sysm.N, 128
out = torch.empty((N, NN, 4), device=dev)


def timeit(reps=200):
    htf.ops.build_pair_vectors(sysm.pos, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, 3.0, NN, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        htf.ops.build_pair_vectors(sysm.pos, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, 3.0, NN, out=out)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


ts = [timeit() for _ in range(5)]
import hashlib
print("HTF_BUILD_TAILS=%s rows=%d %s: %s median %.1f us  tensor sha1 %s" % (
    os.environ.get("HTF_BUILD_TAILS", "default"), N, "f64" if a.f64 else "f32", " ".join("%.1f" % t for t in ts), float(np.median(ts)),
    hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]))
