"""ctypes loader for oracle/_build/libhtf_oracle.so (TEST INFRASTRUCTURE ONLY)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libhtf_oracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    lib.htfo_num_threads.restype = C.c_int
    # use the cores this process may actually run on (cgroup/affinity), not the host total
    lib.htfo_set_threads(C.c_int(usable_cpus()))
    return lib


def usable_cpus():
    """CPUs this process can really use: min(affinity mask, cgroup cpu.max quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _box_args(box, periodic):
    lo = np.ascontiguousarray(box[0], dtype=np.float64)
    hi = np.ascontiguousarray(box[1], dtype=np.float64)
    tilt = np.ascontiguousarray(box[2], dtype=np.float64)
    per = np.ascontiguousarray(periodic, dtype=np.int32)
    return lo, hi, tilt, per


def prepare_neighbors(lib, pos4, n_neigh, head, nlist, box, r_cut, NN, offset=0, batch=None, periodic=(1, 1, 1)):
    N = len(n_neigh)
    B = N - offset if batch is None else batch
    f64 = pos4.dtype == np.float64
    dest = np.empty((B, NN, 4), dtype=pos4.dtype)
    lo, hi, tilt, per = _box_args(box, periodic)
    fn = lib.htfo_prepare_neighbors_f64 if f64 else lib.htfo_prepare_neighbors_f32
    fn(_p(dest), _p(np.ascontiguousarray(pos4)), _p(n_neigh), _p(head), _p(nlist), _p(lo), _p(hi), _p(tilt), _p(per),
       C.c_double(r_cut), C.c_uint(NN), C.c_uint(offset), C.c_uint(B))
    return dest


def lj_from_nlist(lib, nl):
    N, NN = nl.shape[:2]
    out = np.empty((N, 4), dtype=np.float32)
    lib.htfo_lj_from_nlist(_p(np.ascontiguousarray(nl, dtype=np.float32)), C.c_uint(N), C.c_uint(NN), _p(out))
    return out


def compute_forces_lj(lib, pos4, n_neigh, head, nlist, box, r_cut, NN, scratch=None, periodic=(1, 1, 1)):
    N = len(n_neigh)
    force = np.empty((N, 4), dtype=np.float32)
    if scratch is None:
        scratch = np.empty((N, NN, 4), dtype=np.float32)
    lo, hi, tilt, per = _box_args(box, periodic)
    lib.htfo_compute_forces_lj_f32(_p(pos4), C.c_uint(N), _p(n_neigh), _p(head), _p(nlist), _p(lo), _p(hi), _p(tilt),
                                   _p(per), C.c_double(r_cut), C.c_uint(NN), _p(scratch), _p(force))
    return force


def wca_from_nlist(lib, nl, sigma):
    N, NN = nl.shape[:2]
    out = np.empty((N, 4), dtype=np.float32)
    lib.htfo_wca_from_nlist(_p(np.ascontiguousarray(nl, dtype=np.float32)), C.c_uint(N), C.c_uint(NN), C.c_float(sigma), _p(out))
    return out


def mlp_from_nlist(lib, nl, params, low=0.0, high=3.0, act="tanh", out=None):
    """Pair-MLP composite on a dense [N, NN, 4] fp32 tensor -> [N, 4]."""
    N, NN = nl.shape[:2]
    ws = [np.ascontiguousarray(params[k], dtype=np.float32) for k in ("W1", "b1", "W2", "b2", "W3", "b3")]
    K, H1 = ws[0].shape
    H2 = ws[2].shape[1]
    assert max(K, H1, H2) <= 64
    if out is None:
        out = np.empty((N, 4), dtype=np.float32)
    nl = nl if (nl.dtype == np.float32 and nl.flags.c_contiguous) else np.ascontiguousarray(nl, dtype=np.float32)
    lib.htfo_mlp_from_nlist(_p(nl), C.c_uint(N), C.c_uint(NN), C.c_int(K), C.c_int(H1), C.c_int(H2), C.c_float(low),
                            C.c_float(high), *[_p(w) for w in ws], C.c_int(1 if act == "tanh" else 0), _p(out))
    return out


def eds_from_nlist(lib, nl, alpha, r0, gap, rdf_range=(0.0, 3.5), nb_total=102, out=None):
    """Config C4's model (LJ + alpha * soft-RDF CV, compute_rdf histogram) -> (forces [N, 4], cv, hist [nb_total])."""
    N, NN = nl.shape[:2]
    if out is None:
        out = np.empty((N, 4), dtype=np.float32)
    hist = np.zeros(nb_total, dtype=np.uint64)
    nl = nl if (nl.dtype == np.float32 and nl.flags.c_contiguous) else np.ascontiguousarray(nl, dtype=np.float32)
    lib.htfo_eds_from_nlist.restype = C.c_double
    cv = lib.htfo_eds_from_nlist(_p(nl), C.c_uint(N), C.c_uint(NN), C.c_float(alpha), C.c_float(r0), C.c_float(gap),
                                 C.c_float(rdf_range[0]), C.c_float(rdf_range[1]), C.c_uint(nb_total), _p(hist), _p(out))
    return out, float(cv), hist
