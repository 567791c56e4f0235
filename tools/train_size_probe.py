"""Does the pair-MLP training sweep depend on how many rows one launch takes?  1 048 576 x 128 pair vectors (config 5's undivided
box): one launch against the sum of eight 131 072-row launches, split16 and fp32 sweeps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hoomd_tf_amd as htf
from hoomd_tf_amd import standin
from hoomd_tf_amd.initializers import mlp_params

dev = torch.device("cuda:0")
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pos, L, a = standin.fcc_positions(cells, 0.8442)
rng = np.random.default_rng(5)
pos = pos + 0.05 * a * rng.standard_normal(pos.shape)
pos -= np.round(pos / L) * L
s = standin.System(pos, L, dtype=torch.float32, device=dev)
nl = standin.CellNlist(s, r_cut=3.0, r_buff=0.4)
nl.build()
pv = htf.ops.build_pair_vectors(s.pos, nl.n_neigh, nl.head_list, nl.nlist, s.box, 3.0, 128)
N = s.N
print("rows", N, "tensor GiB", pv.numel() * 4 / 2**30)
labels = htf.ops.eval_forces(htf.Potential.lj(), pv)
cap = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
f3 = labels[:, :3]
print("largest |label|", float(f3.norm(dim=1).max()), "cap", cap)
if cap > 0:
    f3.mul_(torch.clamp(cap / f3.norm(dim=1, keepdim=True).clamp_min(1e-12), max=1.0))
outlier = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
if outlier:
    labels[12345, 0] = outlier       # one row with a residual far above the rest: what a launch-wide seed scale must survive
    print("median |label|", float(labels[:, :3].norm(dim=1).median()), "outlier", outlier)
params = mlp_params(seed=3)
flat = np.concatenate([np.asarray(params[k], dtype=np.float32).ravel() for k in ("W1", "b1", "W2", "b2", "W3", "b3")])
out = {}
for prec in ("split16", "fp32"):
    theta = torch.tensor(flat, device=dev)
    pot = htf.Potential.pair_mlp(params, 0.0, 3.0, activation="tanh", precision=prec, theta=theta)
    whole = htf.ops.train_pair_grad(pot, pv, labels).double().cpu().numpy()
    again = htf.ops.train_pair_grad(pot, pv, labels).double().cpu().numpy()
    parts = np.zeros_like(whole)
    nchunk = max(1, N // 131072)
    for c in range(nchunk):
        lo, hi = c * (N // nchunk), (c + 1) * (N // nchunk)
        parts += htf.ops.train_pair_grad(pot, pv[lo:hi].contiguous(), labels[lo:hi].contiguous()).double().cpu().numpy()
    out[prec] = (whole, parts)
    g = np.abs(whole[1:]).max()
    print(prec, "loss whole/parts", whole[0], parts[0], "rel", abs(whole[0] - parts[0]) / whole[0])
    print(prec, "max|g|", g, "whole-parts / max|g|", np.abs(whole[1:] - parts[1:]).max() / g, "deterministic", np.array_equal(whole, again))
    k = int(np.argmax(np.abs(whole[1:] - parts[1:])))
    print(prec, "worst weight", k, whole[1 + k], parts[1 + k])
g = np.abs(out["fp32"][1][1:]).max()
print("split16 parts vs fp32 parts / max|g|", np.abs(out["split16"][1][1:] - out["fp32"][1][1:]).max() / g)
print("split16 whole vs fp32 parts / max|g|", np.abs(out["split16"][0][1:] - out["fp32"][1][1:]).max() / g)
print("fp32 whole vs fp32 parts / max|g|", np.abs(out["fp32"][0][1:] - out["fp32"][1][1:]).max() / g)
