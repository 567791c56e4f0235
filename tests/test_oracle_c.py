"""The C restatement (cpu_baseline 'port') must agree with the numpy oracle."""
import numpy as np
import pytest

from helpers import brute_nlist, fcc_lattice, random_nlist
from oracle import c_oracle
from oracle import htf_oracle as O


@pytest.fixture(scope="module")
def clib():
    return c_oracle.load()


def _stuff(pos, types, dtype):
    p4 = np.zeros((len(pos), 4), dtype=dtype)
    p4[:, :3] = pos
    if dtype == np.float32:
        p4[:, 3] = types.astype(np.int32).view(np.float32)
    else:
        p4[:, 3] = types.astype(np.int64).view(np.float64)
    return p4


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_prepare_neighbors_bit_exact(clib, dtype):
    pos, L = fcc_lattice(3, 1.7)
    rng = np.random.default_rng(0)
    pos = (pos + 0.1 * rng.standard_normal(pos.shape)).astype(dtype)
    types = rng.integers(0, 3, len(pos))
    nn, head, nl = brute_nlist(pos, L, 3.0, shuffle_seed=1)
    box = O.make_box(L, dtype=dtype)
    for NN in (4, 16, 64):
        ref = O.prepare_neighbors(pos, types, nn, head, nl, box, 2.6, NN)
        got = c_oracle.prepare_neighbors(clib, _stuff(pos, types, dtype), nn, head, nl, box, 2.6, NN)
        np.testing.assert_array_equal(got, ref)
    ref = O.prepare_neighbors(pos, types, nn, head, nl, box, 2.6, 16, offset=5, batch_size=9)
    got = c_oracle.prepare_neighbors(clib, _stuff(pos, types, dtype), nn, head, nl, box, 2.6, 16, offset=5, batch=9)
    np.testing.assert_array_equal(got, ref)


def test_lj_matches_numpy_oracle(clib):
    rng = np.random.default_rng(1)
    nl, _ = random_nlist(rng, 256, 64, fill=0.7, rmin=0.9)
    ref = O.lj_model(nl.astype(np.float64))
    got = c_oracle.lj_from_nlist(clib, nl)
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-4)
    assert clib.htfo_num_threads() >= 1


@pytest.mark.parametrize("sigma", [0.5, 1.0])
def test_wca_matches_numpy_oracle(clib, sigma):
    rng = np.random.default_rng(2)
    nl, _ = random_nlist(rng, 256, 64, fill=0.7, rmin=0.6 * sigma, rmax=1.6 * sigma)
    ref = O.wca_model(nl.astype(np.float64), sigma)
    got = c_oracle.wca_from_nlist(clib, nl, sigma)
    # fp32 port against the fp64 oracle: s^7 amplifies the rounding of s seven-fold
    np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-4)
    assert np.abs(ref[:, :3]).max() > 1.0  # the sample reaches inside the cut


@pytest.mark.parametrize("act", ["tanh", "linear"])
def test_pair_mlp_matches_numpy_oracle(clib, act):
    rng = np.random.default_rng(3)
    nl, _ = random_nlist(rng, 128, 32, fill=0.7, rmin=0.5, rmax=3.0)
    params = O.make_mlp_params(seed=3, bias_scale=0.1)
    ref = O.pair_mlp_model(nl.astype(np.float64), params, 0.0, 3.0, act=act)
    got = c_oracle.mlp_from_nlist(clib, nl, params, 0.0, 3.0, act=act)
    scale = np.abs(ref).max()
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-5 * scale + 2e-5)


def test_eds_rdf_matches_numpy_oracle(clib):
    rng = np.random.default_rng(4)
    nl, _ = random_nlist(rng, 300, 48, fill=0.7, rmin=0.9, rmax=3.4)
    ref, cv = O.eds_rdf_model(nl.astype(np.float64), 0.7, 1.1, 0.05)
    got, cv_c, hist = c_oracle.eds_from_nlist(clib, nl, 0.7, 1.1, 0.05)
    np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-4)
    assert abs(cv_c - cv) < 2e-6 * abs(cv)
    r = np.sqrt(np.sum(nl[:, :, :3] * nl[:, :, :3], axis=2, dtype=np.float32)).astype(np.float32)
    np.testing.assert_array_equal(hist.astype(np.int64), O.histogram_fixed_width(r, np.array([0.0, 3.5], np.float32), 102))


def test_c_oracle_under_sanitizers(tmp_path):
    """the CPU restatement built with -fsanitize=address,undefined and run on a system whose
    rows overflow NN (the slot wrap must stay inside the row), a batch, both precisions, 1 and 4 threads."""
    import os
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "san_drv"
    build = subprocess.run(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fopenmp",
                            os.path.join(root, "oracle", "sanitize_driver.c"), os.path.join(root, "oracle", "htf_oracle_c.c"),
                            "-lm", "-o", str(exe)], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert run.returncode == 0 and "sanitize_driver ok" in run.stdout, run.stdout + run.stderr
