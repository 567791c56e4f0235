#!/usr/bin/env python3
"""Compile, into hoomd_tf_amd/_jit_cache, the generated units the test suite and bench.py's generic-lj line use -- no GPU needed
(hipRTC or hipcc cross-compile for gfx950): a GPU box that receives the tree then loads them from the cache instead of spending
~4 s per expression.  Called by __graft_entry__.build() (best effort); the cache key covers the body, every kernel source and the
compiler, so a stale entry is simply never hit."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hoomd_tf_amd as htf  # noqa: E402
from hoomd_tf_amd import codegen as cg  # noqa: E402
from hoomd_tf_amd.simmodel import PositionsInput  # noqa: E402


def bodies():
    from test_codegen_cpu import _models, _random_expression, _typed_models
    x = htf.Nlist(torch.zeros((2, 4, 4)))
    P = PositionsInput.wrap(torch.zeros((2, 4)))
    out = [e.body() for e in _models(htf, x).values()]
    out += [e.body() for e in _typed_models(htf, x, P, 3, big=True).values()]
    s, r = htf.nlist_rinv(x), htf.safe_norm(x[:, :, :3], axis=2)
    # bench.py --workload generic-lj: Morse, LJ + Yukawa, the Kob-Andersen mixture
    live = htf.cast(s > 0.0, torch.float32)
    m = 1.0 - htf.exp(-5.0 * (r - 1.122))
    out.append(htf.reduce_sum(0.5 * live * (m * m - 1.0), axis=1).body())
    out.append(htf.reduce_sum(2.0 * (s ** 12 - s ** 6) + 0.25 * htf.exp(-1.0 * r) * s, axis=1).body())
    idx = htf.cast(P[:, 3], torch.int32)[:, None] * 2 + htf.cast(x[:, :, 3], torch.int32)
    q = (htf.gather([1.0, 0.8, 0.8, 0.88], idx) * s) ** 6
    out.append(htf.reduce_sum(2.0 * htf.gather([1.0, 1.5, 1.5, 0.5], idx) * (q * q - q), axis=1).body())
    qq = htf.gather([1.0, -1.0, -1.0, 1.0], idx)
    out.append(htf.reduce_sum(2.0 * (s ** 12 - s ** 6) + 0.5 * 2.0 * qq * htf.erfc(0.35 * r) * s, axis=1).body())
    # tests/test_gpu_codegen.py: the three-species mixture of test_typed_model_is_replayed..., the random trees
    rng = np.random.default_rng(7)
    eps = rng.uniform(0.6, 1.4, (3, 3))
    eps = 0.5 * (eps + eps.T)
    sig = rng.uniform(0.85, 1.0, (3, 3))
    sig = 0.5 * (sig + sig.T)
    idx3 = htf.cast(P[:, 3], torch.int32)[:, None] * 3 + htf.cast(x[:, :, 3], torch.int32)
    q = (htf.gather(sig.reshape(-1), idx3) * s) ** 6
    out.append(htf.reduce_sum(2.0 * htf.gather(eps.reshape(-1), idx3) * (q * q - q), axis=1).body())
    # models with trainable weights (kernel arguments): tests/test_codegen_cpu.py's and test_gpu_codegen.py's, bench generic-lj's
    from test_codegen_cpu import _weighted_models
    w = torch.nn.Parameter(torch.tensor([0.8, 1.05]))
    a, b = torch.nn.Parameter(torch.tensor(1.3)), torch.nn.Parameter(torch.tensor(0.7))
    out += [htf.reduce_sum(e, axis=1).body() for e in _weighted_models(htf, x, w, a, b).values()]
    # row functions (energy_i = F(sum_j g(r_ij)): embedded-atom terms, coordination restraints): the tests' and the bench's
    from test_codegen_cpu import _row_models
    for e in _row_models(htf, x).values():
        out += [t.body() for t in (e.groups() or [])]
    u = htf.reduce_sum(2.0 * (s ** 12 - s ** 6) * htf.cast(s > 0.4, torch.float32), axis=1)
    out += [t.body() for t in (u + 0.02 * u * u).groups()]
    u = htf.reduce_sum(2.0 * (s ** 12 - s ** 6), axis=1)                         # (bench generic-lj: traced_many_body, traced_embedded_atom)
    out += [t.body() for t in (u + 0.02 * u * u).groups()]
    rho = htf.reduce_sum(htf.exp(-1.7 * r) * s * s, axis=1)
    out += [t.body() for t in (-1.3 * htf.sqrt(rho) + htf.reduce_sum(2.0 * s ** 12, axis=1)).groups()]
    # tests/test_gpu_codegen.py: random row functions, embedded-atom terms with weights (kernel arguments of body and row function)
    from test_gpu_codegen import _random_row_energy
    for seed in range(12):
        out += [t.body() for t in (_random_row_energy(htf, x, seed).groups() or [])]
    x1 = htf.Nlist(torch.zeros((2, 4, 4)))     # (a fresh trace each: its weight vector starts at index 0, as in the test's compute())
    s1, r1 = htf.nlist_rinv(x1), htf.safe_norm(x1[:, :, :3], axis=2)
    amp, decay = torch.nn.Parameter(torch.tensor(1.3)), torch.nn.Parameter(torch.tensor(1.7))
    rho = htf.reduce_sum(htf.exp(-1.0 * decay * r1) * s1 * s1, axis=1)
    out += [t.body() for t in (-1.0 * amp * htf.sqrt(rho + 0.01)).groups()]
    x2 = htf.Nlist(torch.zeros((2, 4, 4)))
    s2, r2 = htf.nlist_rinv(x2), htf.safe_norm(x2[:, :, :3], axis=2)
    amp2 = torch.nn.Parameter(torch.tensor(1.3))
    rho2 = htf.reduce_sum(htf.exp(-1.7 * r2) * s2 * s2, axis=1)
    out += [t.body() for t in (htf.reduce_sum(2.0 * s2 ** 12, axis=1) - amp2 * htf.sqrt(rho2 + 0.01)).groups()]
    return out


def hipcc_bodies():
    """The units tests/test_codegen_cpu.py builds with the FALLBACK compiler (`hipcc --genco`, ~25 s each on a cold cache)."""
    from test_codegen_cpu import _models, _weighted_models
    x = htf.Nlist(torch.zeros((2, 4, 4), dtype=torch.float64))
    w = torch.nn.Parameter(torch.tensor([1.1, 0.95], dtype=torch.float64))
    a, b = torch.nn.Parameter(torch.tensor(0.7, dtype=torch.float64)), torch.nn.Parameter(torch.tensor(2.3, dtype=torch.float64))
    return [_models(htf, x)["yukawa"].body(), _weighted_models(htf, x, w, a, b)["yukawa_mix"].body(),
            (htf.exp(-0.7 * htf.safe_norm(x[:, :, :3], axis=2)) * htf.nlist_rinv(x)).body()]


def main():
    t0, n, hit = time.time(), 0, 0
    for b in dict.fromkeys(bodies()):
        path = os.path.join(cg._cache_dir(), cg._key(b) + ".hsaco")
        hit += os.path.exists(path)
        cg.compile_body(b)
        n += 1
    try:
        cg._hipcc()
        os.environ["HTF_JIT_COMPILER"] = "hipcc"
        for b in hipcc_bodies():
            path = os.path.join(cg._cache_dir(), cg._key(b) + ".hsaco")
            hit += os.path.exists(path)
            cg.compile_body(b)
            n += 1
    except RuntimeError:
        pass
    finally:
        os.environ.pop("HTF_JIT_COMPILER", None)
    print("warm_jit_cache: %d units (%d already cached) with %s in %.0f s" % (n, hit, cg.compiler(), time.time() - t0))


if __name__ == "__main__":
    main()
