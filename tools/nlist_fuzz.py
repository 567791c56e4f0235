"""The stand-in's binned neighbor search on 60 random boxes (anisotropic, rho 0.3-0.9, list radius 1.1-2.7, both precisions, every
other pair of seeds with the type split of a mapped list) against the O(N^2) brute force: per-row neighbor SETS, fp64 exact,
fp32 up to 8 pairs within rounding of r_list.  Beyond tests/test_gpu_standin.py's eight seeds; every box here has >= 1 024 cells,
i.e. runs the one-wave-per-cell kernel.      gpurun -- 'python tools/nlist_fuzz.py'"""
import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import hoomd_tf_amd as htf
from hoomd_tf_amd import standin
from helpers import brute_nlist
cuda = torch.device("cuda:0")
def rows(nn, head, nl): return [nl[h:h + n] for n, h in zip(nn, head)]
bad = 0
for seed in range(100, 160):
    rng = np.random.default_rng(seed)
    tdt = torch.float64 if seed % 2 else torch.float32
    L = rng.uniform(8.0, 30.0, size=3)
    rho = float(rng.uniform(0.3, 0.9))
    N = int(min(12000, max(64, rho * np.prod(L))))
    pos = (rng.random((N, 3)) - 0.5) * L
    r_cut, r_buff = float(rng.uniform(1.0, 2.2)), float(rng.uniform(0.1, 0.5))
    types = (rng.random(N) < 0.3).astype(np.int32) if seed % 4 >= 2 else None
    sysm = standin.System(pos, L, types=types, dtype=tdt, device=cuda)
    nl = standin.CellNlist(sysm, r_cut=r_cut, r_buff=r_buff)
    nl.type_split = 1 if types is not None else -1
    nl.build()
    p = sysm.pos.cpu().numpy()[:, :3].astype(np.float64)
    bn, bh, bl = brute_nlist(p, L, r_cut + r_buff)
    ref = rows(bn, bh, bl)
    if types is not None: ref = [r[types[r] == types[i]] for i, r in enumerate(ref)]
    got = rows(nl.n_neigh.cpu().numpy(), nl.head_list.cpu().numpy(), nl.nlist.cpu().numpy())
    mism = sum(len(set(g.tolist()) ^ set(r.tolist())) for g, r in zip(got, ref))
    n, w = nl._ncell()
    ok = mism <= (8 if tdt == torch.float32 else 0)
    bad += not ok
    print(seed, "N", N, "cells", int(np.prod(n)), "stencil", [int(v) for v in w], str(tdt)[6:], "types" if types is not None else "-", "mismatch", mism, "OK" if ok else "FAIL")
print("failures:", bad)
