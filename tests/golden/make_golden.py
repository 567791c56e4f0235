#!/usr/bin/env python3
"""Regenerates tests/golden/golden_v1.npz.

The reference (hoomd-tf) cannot be imported in this image (needs TensorFlow + HOOMD-blue),
so these vectors are produced by the CPU oracle (oracle/htf_oracle.py, fp64) after it has
been pinned against the reference tests' known answers (tests/test_reference_kats.py).
They guard the oracle against regressions and give the HIP path fixed inputs/outputs that
do not depend on the oracle code at test time.  Cases follow SURVEY 8(c) "Golden vectors".
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from helpers import brute_nlist, fcc_lattice  # noqa: E402
from oracle import htf_oracle as O  # noqa: E402


def main():
    rng = np.random.default_rng(20260101)
    out = {}
    # (ii) 108-particle jittered fcc, two types, NN=16 (overflows: exercises the wrap) and NN=64
    pos, L = fcc_lattice(3, 1.68)
    pos = pos + 0.08 * rng.standard_normal(pos.shape)
    pos -= np.round(pos / L) * L
    types = rng.integers(0, 2, len(pos)).astype(np.int32)
    nn, head, nl = brute_nlist(pos, L, 2.9, shuffle_seed=5)
    box = O.make_box(L)
    out.update(pos=pos, types=types, L=L, n_neigh=nn, head_list=head, nlist=nl, r_cut=np.float64(2.5))
    for NN in (16, 64):
        pv64 = O.prepare_neighbors(pos, types, nn, head, nl, box, 2.5, NN)
        pv32 = O.prepare_neighbors(pos.astype(np.float32), types, nn, head, nl, O.make_box(L, dtype=np.float32), 2.5, NN)
        out["pv64_NN%d" % NN], out["pv32_NN%d" % NN] = pv64, pv32
    x = out["pv32_NN64"].astype(np.float64)
    f, v = O.lj_model(x, virial=True)
    out["lj_force"], out["lj_virial"] = f, v
    out["wca05_force"] = O.wca_model(x, 0.5)
    out["wca10_force"] = O.wca_model(x, 1.0)
    out["rinv_force"] = O.benchmark_potential(x)
    out["simple_force"] = O.simple_potential(x)
    params = O.make_mlp_params(seed=3, K=32, H1=64, H2=64, bias_scale=0.2)
    for k, w in params.items():
        out["mlp_" + k] = w
    out["mlp_tanh_force"] = O.pair_mlp_model(x, params, 0.0, 3.0, "tanh")
    out["mlp_linear_force"] = O.pair_mlp_model(x, params, 0.0, 3.0, "linear")
    # (iii) edge slots
    e = np.zeros((2, 8, 4), np.float32)
    e[0, 0, :3] = [1.6e-6, 1.6e-6, 1.6e-6]
    e[0, 1, :3] = [1.8e-6, 1.8e-6, 1.8e-6]
    e[0, 2, :3] = [0.5 * 10 ** (-1 / 6) * 1.002, 0, 0]
    e[0, 3, :3] = [0, 0.5 * 10 ** (-1 / 6) * 0.998, 0]
    e[0, 4, :3] = [0, 0, 0.5 * 2 ** (1 / 3) * 1.001]
    e[1, 0, :3] = [1.0, 0, 0]
    out["edge_nlist"] = e
    out["edge_rinv"] = O.nlist_rinv(e)
    out["edge_wca05_force"] = O.wca_model(e.astype(np.float64), 0.5)
    # (iv) RBFExpansion(0,2,10) on safe_norm(ones(10,6,3))
    out["rbf_ones"] = O.rbf_expansion(O.safe_norm(np.ones((10, 6, 3), np.float32), axis=2), 0, 2, 10)
    # (v) EDSLayer(4.0, period=5, lr=0.2) on a scripted CV sequence
    cvs = (4.0 + np.random.default_rng(11).standard_normal(200)).astype(np.float32)
    eds = O.EDSLayer(4.0, 5, 0.2)
    out["eds_cv"], out["eds_alpha"] = cvs, np.array([eds(c) for c in cvs], dtype=np.float32)
    # (vi) compute_rdf
    out["rdf_0_35"], _ = O.compute_rdf(out["pv32_NN64"], [0, 3.5], nbins=20)
    out["rdf_typed"], _ = O.compute_rdf(out["pv32_NN64"], [0, 3.5], types.astype(np.float32), nbins=20, type_i=0, type_j=1)
    np.savez_compressed(os.path.join(HERE, "golden_v1.npz"), **out)
    print("wrote", os.path.join(HERE, "golden_v1.npz"), sum(v.nbytes for v in out.values()) // 1024, "KiB raw")


if __name__ == "__main__":
    main()
