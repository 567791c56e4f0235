"""Randomised cross-check of the three traced modes of htf_compute_forces (two kernels / one kernel
with the tensor / one kernel without it) over system size, density, NN (overflowing and not),
precision, batching, row ranges, virial and potential, with the particles moving between calls so
that the per-row live counts (delta zero-fill) change from step to step."""
import os

import numpy as np
import pytest
import torch

from helpers import brute_nlist

pytestmark = pytest.mark.gpu


def _potentials(htf):
    return [htf.Potential.lj(), htf.Potential.wca(1.0), htf.Potential.rinv_poly([1.0, -0.5], [12, 6]),
            htf.Potential.lj_param(1.1, 0.95), htf.Potential.simple()]


@pytest.mark.parametrize("seed", range(int(os.environ.get("HTF_FUZZ_SEEDS", "12"))))
def test_modes_agree_on_random_systems(htf, cuda, seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(3, 8))
    dim3 = bool(rng.integers(0, 2))
    a = float(rng.uniform(1.0, 1.6))
    if dim3:
        g = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
        L = np.array([n * a] * 3)
    else:
        g = np.stack(np.meshgrid(np.arange(n * 2), np.arange(n * 2), indexing="ij"), -1).reshape(-1, 2).astype(np.float64)
        g = np.concatenate([g, np.zeros((len(g), 1))], axis=1)
        L = np.array([2 * n * a, 2 * n * a, 4.0])
    pos = (g + 0.5) * a - L / 2
    pos[:, :3 if dim3 else 2] += 0.1 * a * rng.standard_normal((len(pos), 3 if dim3 else 2))
    pos -= np.round(pos / L) * L
    N = len(pos)
    r_cut = float(rng.uniform(1.3, min(2.6, 0.45 * L[:2].min())))
    NN = int(rng.choice([4, 8, 24, 64, 128, 160]))
    tdt = torch.float64 if rng.integers(0, 2) else torch.float32
    batch = int(rng.choice([0, 0, 7, 33]))
    pots = _potentials(htf)
    pot = pots[int(rng.integers(0, len(pots)))]
    virial = bool(rng.integers(0, 2)) and pot.kind != htf._lib.POT_SIMPLE
    types = np.zeros(N, dtype=np.int32)
    outs = {}
    moves = [0.05 * a * rng.standard_normal(pos.shape) * ([1, 1, 1] if dim3 else [1, 1, 0]) for _ in range(2)]
    for mode in (0, 2, 1):
        p = pos.copy()
        ctx = htf.Context(r_cut=r_cut, nneighs=NN, batch_size=batch, scalar_dtype=tdt, virial=virial, max_n=N, fused=mode)
        ctx.set_potential(pot)
        res = []
        for step in range(3):
            nn, head, nl = brute_nlist(p, L, r_cut + 0.3, shuffle_seed=seed)
            p4 = htf.ops.stuff_types(torch.from_numpy(p).to(cuda), torch.from_numpy(types).to(cuda), tdt)
            dnn, dhead, dnl = (torch.from_numpy(x.astype(np.int32)).to(cuda) for x in (nn, head, nl))
            force = torch.zeros((N, 4), dtype=tdt, device=cuda)
            vir = torch.zeros(6 * N, dtype=tdt, device=cuda) if virial else None
            arr = ctx.make_arrays(p4, N, dnn, dhead, dnl, htf._lib.make_box(np.array([-L / 2, L / 2, [0, 0, 0]])), force, vir, N)
            if batch == 0 and step == 1 and N > 20:  # a step computed in two row ranges
                n1 = int(rng.integers(1, N - 1))
                ctx.compute_forces(step, arr, rows=(0, n1))
                ctx.compute_forces(step, arr, rows=(n1, N - n1))
            else:
                ctx.compute_forces(step, arr)
            torch.cuda.synchronize()
            nb = ctx.nlist_buffer(N if batch == 0 else batch, cuda).clone() if mode != 1 else None
            res.append((force.clone(), vir.clone() if virial else None, nb))
            if step < 2:
                p = p + moves[step]
                p -= np.round(p / L) * L
        outs[mode] = res
    for step in range(3):
        f0, v0, t0 = outs[0][step]
        f2, v2, t2 = outs[2][step]
        f1, v1, _ = outs[1][step]
        assert torch.equal(t0, t2), "pair-vector tensor differs between the two-kernel and the one-kernel mode"
        scale = max(1.0, float(f0.abs().max()))
        assert float((f2 - f0).abs().max()) <= 5e-5 * scale, (seed, step, float((f2 - f0).abs().max()), scale)
        assert float((f1 - f2).abs().max()) <= 5e-5 * scale
        if virial:
            vs = max(1.0, float(v0.abs().max()))
            assert float((v2 - v0).abs().max()) <= 1e-4 * vs and float((v1 - v2).abs().max()) <= 1e-4 * vs
