"""Duration of the training sweep from the index list (htf_train_pair_grad_list) against the tensor sweep (htf_build_pair_vectors +
htf_train_pair_grad) at C3 (131 072 x 128) and C2, for the trainable LJ (two weights).   python tools/train_list_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hoomd_tf_amd as htf  # noqa: E402
from hoomd_tf_amd import standin  # noqa: E402

dev = torch.device("cuda:0")
for name, lattice, cells in (("C3", "fcc", 32), ("C2", "sc", 32)):
    pos, L, a = (standin.sc_positions if lattice == "sc" else standin.fcc_positions)(cells, 0.8442)
    rng = np.random.default_rng(7)
    pos = pos + 0.05 * a * rng.standard_normal(pos.shape)
    pos -= np.round(pos / L) * L
    sysm = standin.System(pos, L, dtype=torch.float32, device=dev)
    nl = standin.CellNlist(sysm, r_cut=3.0, r_buff=0.4)
    nl.build()
    pot = htf.Potential.lj_param(0.9, 1.05, theta=torch.tensor([0.9, 1.05], device=dev))
    labels = 0.05 * torch.randn((sysm.N, 4), device=dev)

    def timed(fn, n=40):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    t_list = timed(lambda: htf.ops.train_pair_grad_list(pot, sysm.pos, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, 3.0, 128, labels))
    pv = htf.ops.build_pair_vectors(sysm.pos, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, 3.0, 128)
    t_tensor = timed(lambda: htf.ops.train_pair_grad(pot, pv, labels))
    t_build = timed(lambda: htf.ops.build_pair_vectors(sysm.pos, nl.n_neigh, nl.head_list, nl.nlist, sysm.box, 3.0, 128))
    print("%s  N %d  list sweep %.1f us | tensor sweep %.1f us + build %.1f us (each with its 6 us column reduction)" % (name, sysm.N, t_list, t_tensor, t_build))
