"""Shared test fixtures: HOOMD-like lattices and a brute-force neighbor list in
HOOMD's storage layout (n_neigh / head_list / flat nlist, FULL mode).

The reference's tests build these with ``hoomd.init.create_lattice`` +
``hoomd.md.nlist.cell`` (test_tensorflow.py:89-92 etc.); HOOMD is absent here, so
the same systems are generated directly.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def sq_lattice(n, a, dtype=np.float64):
    """hoomd.lattice.sq(a) replicated n x n: 2-D, box (n a, n a, 1), centred."""
    L = np.array([n * a, n * a, 1.0])
    ij = np.stack(np.meshgrid(np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 2)
    pos = np.zeros((n * n, 3))
    pos[:, :2] = (ij + 0.5) * a - L[:2] / 2
    return pos.astype(dtype), L


def sc_lattice(n, a, dtype=np.float64):
    L = np.array([n * a] * 3, dtype=np.float64)
    ijk = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).reshape(-1, 3)
    pos = (ijk + 0.5) * a - L / 2
    return pos.astype(dtype), L


def bcc_lattice(n, a, dtype=np.float64):
    L = np.array([n * a] * 3, dtype=np.float64)
    ijk = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).reshape(-1, 3)
    base = np.array([[0.25, 0.25, 0.25], [0.75, 0.75, 0.75]])
    pos = ((ijk[:, None, :] + base[None]) * a).reshape(-1, 3) - L / 2
    return pos.astype(dtype), L


def fcc_lattice(n, a, dtype=np.float64):
    L = np.array([n * a] * 3, dtype=np.float64)
    ijk = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).reshape(-1, 3)
    base = np.array([[0.25, 0.25, 0.25], [0.75, 0.75, 0.25], [0.75, 0.25, 0.75], [0.25, 0.75, 0.75]])
    pos = ((ijk[:, None, :] + base[None]) * a).reshape(-1, 3) - L / 2
    return pos.astype(dtype), L


def wrap(pos, L):
    return pos - np.round(pos / L) * L


def min_image_np(d, L):
    return d - np.round(d / L) * L


def brute_nlist(pos, L, r_list, n_local=None, shuffle_seed=None, pitch=None):
    """O(N^2) FULL neighbor list in HOOMD layout.

    Returns (n_neigh u32[N], head_list u32[N], nlist u32[sum]).  ``pitch`` gives
    HOOMD's fixed-stride head list (head[i] = i*pitch); default is a packed prefix
    sum.  ``shuffle_seed`` permutes each row (HOOMD's order is cell-walk order, i.e.
    arbitrary) so tests cannot depend on ascending neighbor indices.
    """
    Ntot = pos.shape[0]
    N = Ntot if n_local is None else n_local
    d = pos[None, :, :].astype(np.float64) - pos[:N, None, :].astype(np.float64)
    d = min_image_np(d, np.asarray(L, dtype=np.float64))
    r2 = np.sum(d * d, axis=2)
    m = r2 <= r_list * r_list
    m[np.arange(N), np.arange(N)] = False
    rng = np.random.default_rng(shuffle_seed) if shuffle_seed is not None else None
    rows = []
    for i in range(N):
        k = np.nonzero(m[i])[0]
        if rng is not None:
            k = rng.permutation(k)
        rows.append(k)
    n_neigh = np.array([len(r) for r in rows], dtype=np.uint32)
    if pitch is None:
        head = np.concatenate([[0], np.cumsum(n_neigh)[:-1]]).astype(np.uint32)
        flat = np.concatenate(rows).astype(np.uint32) if rows else np.zeros(0, np.uint32)
    else:
        assert pitch >= n_neigh.max()
        head = (np.arange(N) * pitch).astype(np.uint32)
        flat = np.zeros(N * pitch, dtype=np.uint32)
        for i, r in enumerate(rows):
            flat[i * pitch: i * pitch + len(r)] = r
    return n_neigh, head, flat


def analytic_lj(pos, L, r_cut, eps=1.0, sig=1.0):
    """HOOMD md.pair.lj(eps, sig, r_cut) with no shift, in fp64: per-particle force,
    energy (half of each pair energy) and virial (6 comps xx,xy,xz,yy,yz,zz;
    HOOMD PotentialPair convention 0.5 * force_divr * dx_a * dx_b, dx = r_i - r_j).
    The comparator of test_tensorflow.py:335-382 and :619-671.
    """
    N = pos.shape[0]
    d = pos[:, None, :].astype(np.float64) - pos[None, :, :].astype(np.float64)  # r_i - r_j
    d = min_image_np(d, np.asarray(L, dtype=np.float64))
    r2 = np.sum(d * d, axis=2)
    m = (r2 <= r_cut * r_cut)
    m[np.arange(N), np.arange(N)] = False
    r2s = np.where(m, r2, 1.0)
    s2 = sig * sig / r2s
    s6 = s2 ** 3
    force_divr = np.where(m, eps * (48.0 * s6 * s6 - 24.0 * s6) / r2s, 0.0)
    F = np.sum(force_divr[..., None] * d, axis=1)
    E = 0.5 * np.sum(np.where(m, 4.0 * eps * (s6 * s6 - s6), 0.0), axis=1)
    V = np.zeros((N, 6))
    for c, (a, b) in enumerate(((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))):
        V[:, c] = 0.5 * np.sum(force_divr * d[..., a] * d[..., b], axis=1)
    return F, E, V


def random_nlist(rng, N, NN, fill=0.7, rmin=0.8, rmax=3.0, ntypes=1, dtype=np.float32):
    """Synthetic dense [N,NN,4] pair-vector tensor: each row has a random count of
    real neighbors (isotropic directions, r in [rmin, rmax]) then zero padding --
    the layout prepareNeighbors produces."""
    nl = np.zeros((N, NN, 4), dtype=np.float64)
    cnt = np.minimum(NN, rng.binomial(NN, fill, size=N))
    for i in range(N):
        c = cnt[i]
        v = rng.standard_normal((c, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        r = rng.uniform(rmin, rmax, size=(c, 1))
        nl[i, :c, :3] = v * r
        nl[i, :c, 3] = rng.integers(0, ntypes, size=c)
    return nl.astype(dtype), cnt


# ---------------------------------------------------------------------------- the HOOMD-side shim against the fake HOOMD
SHIM = os.path.join(ROOT, "integration", "hoomd_shim")
STUB = os.path.join(ROOT, "integration", "hoomd_stub")


def shim_flags(extra=()):
    import sysconfig

    import pybind11
    return ["g++", "-std=c++14", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Werror", "-fvisibility=hidden", "-fPIC",
            "-D__HIP_PLATFORM_AMD__", "-I", STUB, "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
            "-I", sysconfig.get_paths()["include"], "-I", pybind11.get_include()] + list(extra)


def build_shim(out_dir, lib_path, single=False):
    """Compile and link integration/hoomd_shim/ (-> _htf_amd.so) and the fake HOOMD's module (-> _hoomd_stub.so) into
    ``out_dir``; returns that directory.  g++ only: the shim is host code over the C ABI."""
    import subprocess
    extra = ["-DSINGLE_PRECISION"] if single else []
    out_dir = str(out_dir)
    libdir = os.path.dirname(lib_path)
    objs = []
    for src in ("TensorflowComputeAMD.cc", "module.cc"):
        o = os.path.join(out_dir, src + ".o")
        r = subprocess.run(shim_flags(extra) + ["-c", os.path.join(SHIM, src), "-o", o], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-4000:]
        objs.append(o)
    stub_o = os.path.join(out_dir, "stub.o")
    r = subprocess.run(shim_flags(extra) + ["-c", os.path.join(STUB, "stub_module.cc"), "-o", stub_o], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    r = subprocess.run(["g++", "-shared", "-o", os.path.join(out_dir, "_hoomd_stub.so"), stub_o, "-L", "/opt/rocm/lib", "-lamdhip64"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    # libamdhip64: the one torch has loaded (our library's DT_NEEDED resolves to it too, see _lib.py)
    r = subprocess.run(["g++", "-shared", "-o", os.path.join(out_dir, "_htf_amd.so")] + objs +
                       ["-L", libdir, "-lhtf_amd", "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    return out_dir


# ---- the A/B variants build: every kernel form that lost its A/B, and the getenv switches that select them, are compiled only with
# -DHTF_AB_VARIANTS (tools/build_variant.sh ab -DHTF_AB_VARIANTS -> build_variants/libhtf_ab.so; __graft_entry__.build() makes it).
# The shipped libhtf_amd.so carries the default forms and the generic fallback; forced-form tests run a child process on this one.
VARIANTS_LIB = os.path.join(ROOT, "build_variants", "libhtf_ab.so")


def variants_env(**switches):
    """Environment of a child process that loads the variants build with the given form switches set."""
    import pytest
    if not os.path.exists(VARIANTS_LIB):
        pytest.skip("no variants build (tools/build_variant.sh ab -DHTF_AB_VARIANTS)")
    return dict(os.environ, HTF_AMD_LIB=VARIANTS_LIB, **switches)
