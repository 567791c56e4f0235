"""CPU oracle for the hoomd-tf per-particle force/energy path (numpy restatement).

TEST INFRASTRUCTURE ONLY.  Nothing under ``hoomd_tf_amd/`` may import this module;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do,
and only as the checker.

Every function restates one piece of the reference (``/root/reference``, hoomd-tf
v2.4.0) and cites the file:line it follows.  The reference itself can be neither
imported nor compiled in this image (it needs TensorFlow >= 2.3 and HOOMD-blue 2.x,
both absent), so the arithmetic that TensorFlow performs at run time
(``tf.gradients`` of the energy) is restated here as hand-derived closed-form
gradients.  Those are cross-checked against ``torch.autograd`` applied to an
op-for-op torch transcription of the same forward graph (``oracle/graph_torch.py``,
``tests/test_oracle.py``).

PINNING STATUS (see DESIGN.md "Oracle"):
  * pinned by the reference's own known-answer tests (restated in
    tests/test_reference_kats.py): pair-vector build + SimplePotential
    (test_tensorflow.py:20-35,81-129), LJModel == analytic LJ within 1e-5
    (test_tensorflow.py:335-382), LJ virial xx/xy (test_tensorflow.py:619-671),
    full-nlist count (:559-579), overflow check (:830-848), compute_nlist KATs
    (test_utils.py:187-270), _make_reverse_indices not on this path.
  * PARITY UNPINNED (no numeric known answer exists upstream; pinned only by the
    formulas cited): RBFExpansion values, WCARepulsion values, EDSLayer step-exact
    trace, compute_rdf values, and the pair-MLP composite (which has no reference
    model at all).

Third-party algorithms restated (absent from /root/reference):
  * HOOMD-blue 2.x ``BoxDim::minImage`` (hoomd/BoxDim.h, v2.9.x), rint form.
  * TensorFlow >= 2.3 op semantics: tf.norm, tf.where, tf.clip_by_value gradient,
    tf.math.divide_no_nan, tf.histogram_fixed_width, tf.math.top_k tie order,
    tf.compat.v1.train.AdamOptimizer update rule.
"""
import math

import numpy as np

# simmodel.py:627  "delta = 3e-6"
RINV_DELTA = 3e-6
# simmodel.py:628  "delta=delta / 3 / 10" -- evaluated exactly as python does
RINV_NORM_DELTA = RINV_DELTA / 3 / 10


# --------------------------------------------------------------------------- #
# Box helpers
# --------------------------------------------------------------------------- #
def make_box(L, tilt=(0.0, 0.0, 0.0), dtype=np.float64):
    """The 3x3 box array the plugin hands to the model.

    TensorflowCompute.cc:271-282 updateBox: row 0 = lo, row 1 = hi,
    row 2 = (xy, xz, yz).  HOOMD boxes are centred on the origin.
    """
    L = np.asarray(L, dtype=np.float64)
    return np.array([-L / 2, L / 2, tilt], dtype=dtype)


def box_size(box):
    """simmodel.py:597-603: ``box[1, :] - box[0, :]``."""
    return box[1, :] - box[0, :]


def wrap_vector(r, box):
    """simmodel.py:606-615: ``r - round(r / bs) * bs`` (orthorhombic only).

    tf.math.round rounds half to even, as np.round does.
    """
    bs = box_size(box)
    return r - np.round(r / bs) * bs


def min_image(dx, box, periodic=(1, 1, 1)):
    """HOOMD-blue 2.x BoxDim::minImage, rint form (the one device code uses).

    Third-party (hoomd/BoxDim.h); called from TensorflowCompute.cc:356 and
    TensorflowCompute.cu:128.  Works in the dtype of ``dx``.
    """
    dt = dx.dtype
    w = np.array(dx, dtype=dt, copy=True)
    L = (box[1] - box[0]).astype(dt)
    Linv = (dt.type(1.0) / L).astype(dt)
    xy, xz, yz = (dt.type(v) for v in box[2])
    if periodic[2]:
        img = np.rint(w[..., 2] * Linv[2])
        w[..., 2] -= L[2] * img
        w[..., 1] -= L[2] * yz * img
        w[..., 0] -= L[2] * xz * img
    if periodic[1]:
        img = np.rint(w[..., 1] * Linv[1])
        w[..., 1] -= L[1] * img
        w[..., 0] -= L[1] * xy * img
    if periodic[0]:
        img = np.rint(w[..., 0] * Linv[0])
        w[..., 0] -= L[0] * img
    return w


# --------------------------------------------------------------------------- #
# a2: prepareNeighbors  (HOOMD index nlist -> dense [B, NN, 4] pair vectors)
# --------------------------------------------------------------------------- #
def prepare_neighbors_loops(pos, types, n_neigh, head_list, nlist, box, r_cut, NN,
                            offset=0, batch_size=None, periodic=(1, 1, 1)):
    """Line-by-line restatement of TensorflowCompute.cc:303-374 (CPU variant).

    pos      [Ntot,3] in the HOOMD Scalar dtype (local + ghost particles)
    types    [Ntot]   integer type ids (HOOMD stuffs them into pos.w bits, :367)
    Returns  [B, NN, 4] in pos.dtype.  Pure-python loops: small cases only.

    Semantics pinned here:  memset 0 (:311); neighbor kept unless
    ``rsq > r_cut*r_cut`` (:359, so r == r_cut is KEPT); slot index wraps
    ``(n + 1) % NN`` (:370) so on overflow later neighbors overwrite earlier ones;
    order = HOOMD nlist order.
    """
    dt = pos.dtype
    N = len(n_neigh)
    B = N - offset if batch_size is None else batch_size
    buf = np.zeros((B, NN, 4), dtype=dt)
    rc = dt.type(r_cut)
    for bi, i in enumerate(range(offset, offset + B)):
        nnoffset = 0
        pi = pos[i]
        head = int(head_list[i])
        for j in range(int(n_neigh[i])):
            k = int(nlist[head + j])
            dx = min_image((pos[k] - pi)[None, :], box, periodic)[0]
            if dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2] > rc * rc:
                continue
            buf[bi, nnoffset, :3] = dx
            buf[bi, nnoffset, 3] = dt.type(int(types[k]))
            nnoffset = (nnoffset + 1) % NN
    return buf


def prepare_neighbors(pos, types, n_neigh, head_list, nlist, box, r_cut, NN,
                      offset=0, batch_size=None, periodic=(1, 1, 1)):
    """Vectorised equivalent of :func:`prepare_neighbors_loops` (same semantics).

    Used for sizes where the python loops are too slow; tests assert both agree.
    """
    dt = pos.dtype
    N = len(n_neigh)
    B = N - offset if batch_size is None else batch_size
    rows = np.arange(offset, offset + B)
    cnt = np.asarray(n_neigh)[rows].astype(np.int64)
    tot = int(cnt.sum())
    buf = np.zeros((B, NN, 4), dtype=dt)
    if tot == 0:
        return buf
    row_of = np.repeat(np.arange(B), cnt)
    start = np.cumsum(cnt) - cnt
    j_of = np.arange(tot) - np.repeat(start, cnt)
    k = np.asarray(nlist)[np.asarray(head_list)[rows][row_of].astype(np.int64) + j_of].astype(np.int64)
    dx = min_image(pos[k] - pos[rows][row_of], box, periodic)
    rc = dt.type(r_cut)
    rsq = dx[:, 0] * dx[:, 0] + dx[:, 1] * dx[:, 1] + dx[:, 2] * dx[:, 2]
    keep = ~(rsq > rc * rc)
    # ordinal q of each kept entry inside its row, and per-row total Q
    kc = np.cumsum(keep) - keep
    row_first = np.repeat(start, cnt)
    q = kc - kc[row_first]
    Q = np.bincount(row_of, weights=keep, minlength=B).astype(np.int64)
    # last writer wins on wrap (:370): entry q survives iff q + NN >= Q
    live = keep & (q + NN >= Q[row_of])
    r_, q_, k_ = row_of[live], q[live] % NN, k[live]
    buf[r_, q_, :3] = dx[live]
    buf[r_, q_, 3] = np.asarray(types)[k_].astype(dt)
    return buf


# --------------------------------------------------------------------------- #
# a10/a11: safe_norm, nlist_rinv
# --------------------------------------------------------------------------- #
def safe_norm(t, delta=1e-7, axis=-1):
    """simmodel.py:581-594: ``tf.norm(tensor + delta)`` -- delta on EVERY component."""
    dt = t.dtype
    u = t + dt.type(delta)
    return np.sqrt(np.sum(u * u, axis=axis))


def nlist_rinv(nlist):
    """simmodel.py:618-635.  Returns [N, NN]; padded slots give exactly 0."""
    dt = nlist.dtype
    delta = dt.type(RINV_DELTA)
    r = safe_norm(nlist[:, :, :3], delta=RINV_NORM_DELTA, axis=2)
    with np.errstate(divide="ignore"):
        inv = dt.type(1.0) / (r + delta)
    return np.where(r > delta, inv, dt.type(0.0))


def _rinv_and_grad_factor(nlist):
    """Shared forward pieces for every rinv-based energy.

    Returns (s, t, rprime, cond) with s = nlist_rinv, t = x + 1e-7, r' = |t|.
    d s / d x_c = cond * (-(1/(r'+delta))^2) * t_c / r'   (tf.where routes zero
    gradient to the unselected branch; RealDiv grad; Sqrt/Sum/Square grads).
    """
    dt = nlist.dtype
    t = nlist[:, :, :3] + dt.type(RINV_NORM_DELTA)
    rp = np.sqrt(np.sum(t * t, axis=2))
    delta = dt.type(RINV_DELTA)
    cond = rp > delta
    with np.errstate(divide="ignore"):
        inv = dt.type(1.0) / (rp + delta)
    s = np.where(cond, inv, dt.type(0.0))
    return s, t, rp, cond


def _grad_from_dEds(dEds, s, t, rp, cond):
    """Chain d(sum E)/ds [N,NN] back to d/d nlist [N,NN,4] (4th comp = 0).

    d s/d(r'+delta) = -1/(r'+delta)^2 = -s*s where cond; d r'/d t_c = t_c / r'.
    """
    dt = s.dtype
    with np.errstate(divide="ignore", invalid="ignore"):
        dr = np.where(cond, dEds * (-(s * s)), dt.type(0.0))
        g3 = np.where(cond[..., None], dr[..., None] * t / rp[..., None], dt.type(0.0))
    g = np.zeros(s.shape + (4,), dtype=dt)
    g[..., :3] = g3
    return g


# --------------------------------------------------------------------------- #
# a12/a13/a14: compute_nlist_forces, _add_energy, _compute_virial
# --------------------------------------------------------------------------- #
def add_energy(forces, energy):
    """simmodel.py:558-578 ``_add_energy``."""
    dt = forces.dtype
    N = forces.shape[0]
    energy = np.asarray(energy, dtype=dt)
    if energy.ndim > 1:
        e = energy.reshape(N, -1).sum(axis=1)
    elif energy.ndim == 0:
        e = np.full((N,), energy, dtype=dt)
    else:
        e = energy
    return np.concatenate([forces[:, :3], e.reshape(N, 1)], axis=-1).astype(dt)


def compute_virial(nlist, nlist_forces):
    """simmodel.py:509-523 ``_compute_virial``: plain tf.norm, norm of nlist_forces
    over ALL 4 components, divide_no_nan, minus sign."""
    dt = nlist.dtype
    n3 = nlist[:, :, :3]
    outer = np.einsum("ijk,ijl->ijkl", n3, n3)
    rmag = np.sqrt(np.sum(n3 * n3, axis=2))
    fmag = np.sqrt(np.sum(nlist_forces * nlist_forces, axis=2))
    den = dt.type(2.0) * rmag
    with np.errstate(divide="ignore", invalid="ignore"):
        F_rs = np.where(den == 0, dt.type(0.0), fmag / den)
    return (dt.type(-1.0) * np.einsum("ij,ijkl->ikl", F_rs, outer)).astype(dt)


def nlist_forces_from_grad(nlist, nlist_grad, energy, virial=False):
    """simmodel.py:526-555 ``compute_nlist_forces`` after ``tf.gradients``:
    nlist_forces = 2 * grad; forces = sum over neighbors; energy into column 3."""
    dt = nlist.dtype
    nf = nlist_grad * dt.type(2.0)
    red = nf.sum(axis=1)
    f = add_energy(red, energy)
    if virial:
        return f, compute_virial(nlist, nf)
    return f


def compute_positions_forces_from_grad(pos_grad, energy):
    """simmodel.py:492-506: ``-tf.gradients(energy, positions)`` then _add_energy."""
    return add_energy(-pos_grad, energy)


def positions_radial_model(positions, coef=1.0, power=-1, ncomp=4):
    """compute_positions_forces (simmodel.py:492-506) of e_i = coef * |positions_i[:ncomp]|^power with
    divide_no_nan at |p| = 0.  power -1, ncomp 4 is build_examples.py:59-64 BenchmarkNonlistModel:
    ps = tf.norm(positions, axis=1) (all four columns, the type included); energy = divide_no_nan(1., ps);
    forces = -d(sum energy)/d positions, first three columns, energy appended."""
    p = np.asarray(positions)
    dt = p.dtype
    n = np.sqrt(np.sum(p[:, :ncomp] * p[:, :ncomp], axis=1))
    ok = n > 0
    safe = np.where(ok, n, dt.type(1))
    e = np.where(ok, dt.type(coef) * safe ** power, dt.type(0))
    grad = (np.where(ok, dt.type(coef) * power * safe ** (power - 2), dt.type(0)))[:, None] * p[:, :3]
    return compute_positions_forces_from_grad(np.concatenate([grad, np.zeros((len(p), 1), dt)], axis=1), e)


# --------------------------------------------------------------------------- #
# a24: reference workloads (build_examples.py)
# --------------------------------------------------------------------------- #
def rinv_poly_model(nlist, coefs, powers, virial=False, cut=None):
    """Per-particle energy E_i = sum_j sum_k c_k * s_ij^p_k, forces via
    compute_nlist_forces.  Generalises LJModel / BenchmarkPotential / example 01.
    ``cut``: examples/01. Quickstart.ipynb cell 3, ``tf.cast(tf.norm(nlist[:, :, :3], axis=2) < cut, tf.float32) *
    energy`` -- the comparison is taken in fp32 as TF takes it, and the cast carries no gradient."""
    dt = nlist.dtype
    s, t, rp, cond = _rinv_and_grad_factor(nlist)
    e = np.zeros_like(s)
    de = np.zeros_like(s)
    for c, p in zip(coefs, powers):
        e = e + dt.type(c) * s ** int(p)
        de = de + dt.type(c * p) * s ** int(p - 1)
    if cut is not None:
        x32 = nlist[:, :, :3].astype(np.float32)
        r32 = np.sqrt((x32 * x32).sum(axis=2, dtype=np.float32)).astype(np.float32)
        inside = r32 < np.float32(cut)
        e = np.where(inside, e, dt.type(0))
        de = np.where(inside, de, dt.type(0))
    g = _grad_from_dEds(de, s, t, rp, cond)
    return nlist_forces_from_grad(nlist, g, e.sum(axis=1), virial)


def lj_model(nlist, virial=False):
    """build_examples.py:67-77 LJModel (and :104-115 LJVirialModel).

    inv_r6 = rinv**6; p_energy = 4/2 * (inv_r6*inv_r6 - inv_r6);
    energy = sum_j p_energy; forces = compute_nlist_forces(nlist, energy).
    """
    dt = nlist.dtype
    s, t, rp, cond = _rinv_and_grad_factor(nlist)
    inv_r6 = s ** 6
    p_energy = dt.type(4.0 / 2.0) * (inv_r6 * inv_r6 - inv_r6)
    energy = p_energy.sum(axis=1)
    # d p / d inv_r6 = 2 * (2 inv_r6 - 1);  d inv_r6 / d s = 6 s^5
    dEds = dt.type(2.0) * (dt.type(2.0) * inv_r6 - dt.type(1.0)) * (dt.type(6.0) * s ** 5)
    g = _grad_from_dEds(dEds, s, t, rp, cond)
    return nlist_forces_from_grad(nlist, g, energy, virial)


def benchmark_potential(nlist):
    """build_examples.py:25-30 BenchmarkPotential: energy = rinv ([N,NN], summed
    per particle by _add_energy)."""
    s, t, rp, cond = _rinv_and_grad_factor(nlist)
    g = _grad_from_dEds(np.ones_like(s), s, t, rp, cond)
    return nlist_forces_from_grad(nlist, g, s)


def wca_pair_energy(nlist, sigma):
    """layers.py:91-98 WCARepulsion.call -> pair energy [N, NN]."""
    dt = nlist.dtype
    s = nlist_rinv(nlist)
    sig = dt.type(sigma)
    rp6 = (sig * s) ** 6
    n3 = nlist[:, :, :3]
    r = np.sqrt(np.sum(n3 * n3, axis=2))
    mask = (r < sig * dt.type(2 ** (1 / 3))).astype(dt)
    return np.clip(mask * rp6, dt.type(0), dt.type(10))


def wca_model(nlist, sigma=0.5):
    """build_examples.py:221-228 WCA model = WCARepulsion(sigma) +
    compute_nlist_forces(nlist, pair_energy[N,NN]).

    clip_by_value gradient passes where 0 <= e <= 10 (inclusive), else 0.
    """
    dt = nlist.dtype
    s, t, rp, cond = _rinv_and_grad_factor(nlist)
    sig = dt.type(sigma)
    rp6 = (sig * s) ** 6
    n3 = nlist[:, :, :3]
    r = np.sqrt(np.sum(n3 * n3, axis=2))
    mask = (r < sig * dt.type(2 ** (1 / 3))).astype(dt)
    e_raw = mask * rp6
    e = np.clip(e_raw, dt.type(0), dt.type(10))
    pass_grad = ((e_raw >= 0) & (e_raw <= 10)).astype(dt)
    # d (sig*s)^6 / d s = 6 (sig s)^5 * sig
    dEds = pass_grad * mask * (dt.type(6.0) * (sig * s) ** 5 * sig)
    g = _grad_from_dEds(dEds, s, t, rp, cond)
    return nlist_forces_from_grad(nlist, g, e)


def simple_potential(nlist):
    """build_examples.py:9-22 SimplePotential (forward only): F_i = -sum_j x/|x|,
    non-finite -> 0.  Returns [N, 3] (compute_outputs pads the 4th column)."""
    dt = nlist.dtype
    n3 = nlist[:, :, :3]
    rs = np.sqrt(np.sum(n3 * n3, axis=2, keepdims=True))
    with np.errstate(divide="ignore", invalid="ignore"):
        fr = dt.type(-1.0) * ((dt.type(1.0) / rs) * n3)
    fr = np.where(np.isfinite(fr), fr, dt.type(0.0))
    return fr.sum(axis=1)


# --------------------------------------------------------------------------- #
# a16: RBFExpansion, a18: Dense, and the pair-MLP composite
# --------------------------------------------------------------------------- #
def rbf_centers(low, high, count):
    """layers.py:31-34: centers = float32 linspace; gap = centers[1]-centers[0]."""
    c = np.linspace(float(low), float(high), count).astype(np.float32)
    return c, np.float32(c[1] - c[0])


def rbf_expansion(x, low, high, count):
    """layers.py:46-49: exp(-(x[..., None] - centers)**2 / gap)  (gap, not gap^2)."""
    c, gap = rbf_centers(low, high, count)
    dt = x.dtype
    d = x[..., None] - c.astype(dt)
    return np.exp(-(d * d) / dt.type(gap))


def dense(x, W, b, act=None):
    """Keras Dense: x @ W + b, activation None (reference default) or tanh."""
    y = x @ W + b
    if act == "tanh":
        y = np.tanh(y)
    return y


def glorot_uniform(rng, fan_in, fan_out, dtype=np.float32):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=(fan_in, fan_out)).astype(dtype)


def make_mlp_params(seed=3, K=32, H1=64, H2=64, dtype=np.float32, bias_scale=0.0):
    """Weights for the pair-MLP (default_rng(3), glorot-uniform,
    zero bias).  bias_scale > 0 gives non-zero biases for stronger parity tests."""
    rng = np.random.default_rng(seed)
    p = {
        "W1": glorot_uniform(rng, K, H1, dtype), "b1": np.zeros(H1, dtype),
        "W2": glorot_uniform(rng, H1, H2, dtype), "b2": np.zeros(H2, dtype),
        "W3": glorot_uniform(rng, H2, 1, dtype), "b3": np.zeros(1, dtype),
    }
    if bias_scale:
        for k in ("b1", "b2", "b3"):
            p[k] = (bias_scale * rng.standard_normal(p[k].shape)).astype(dtype)
    return p


def pair_mlp_model(nlist, params, low=0.0, high=3.0, act="tanh", return_grad=False):
    """Pair-MLP composite (no reference model exists -- PARITY UNPINNED; defined in
    SURVEY 8(a) from a10 + a16 + a18 + a12):

        r   = safe_norm(nlist[:, :, :3], axis=2)            simmodel.py:581-594
        phi = RBFExpansion(low, high, K)(r)                 layers.py:46-49
        u   = Dense(1)(act(Dense(H2)(act(Dense(H1)(phi))))) Keras Dense
        u   = u * [r > 3e-6]                                nlist_rinv criterion
        E_i = 0.5 * sum_j u_ij ; forces = compute_nlist_forces(nlist, E)

    Analytic backward: du/dr through the MLP; d phi_k/dr = -2 (r - c_k)/gap * phi_k.
    """
    dt = nlist.dtype
    K = params["W1"].shape[0]
    c, gap = rbf_centers(low, high, K)
    c = c.astype(dt)
    gap = dt.type(gap)
    W1, b1, W2, b2, W3, b3 = (params[k].astype(dt) for k in ("W1", "b1", "W2", "b2", "W3", "b3"))
    t = nlist[:, :, :3] + dt.type(1e-7)
    r = np.sqrt(np.sum(t * t, axis=2))
    mask = (r > dt.type(RINV_DELTA)).astype(dt)
    d = r[..., None] - c
    phi = np.exp(-(d * d) / gap)
    z1 = phi @ W1 + b1
    h1 = np.tanh(z1) if act == "tanh" else z1
    z2 = h1 @ W2 + b2
    h2 = np.tanh(z2) if act == "tanh" else z2
    u = (h2 @ W3 + b3)[..., 0]
    E = dt.type(0.5) * (u * mask).sum(axis=1)
    # backward
    dh2 = np.broadcast_to(W3[:, 0], h2.shape)
    dz2 = dh2 * (dt.type(1) - h2 * h2) if act == "tanh" else dh2
    dh1 = dz2 @ W2.T
    dz1 = dh1 * (dt.type(1) - h1 * h1) if act == "tanh" else dh1
    dphi = dz1 @ W1.T
    dudr = (dphi * (dt.type(-2.0) * d / gap) * phi).sum(axis=-1)
    dEdr = dt.type(0.5) * mask * dudr
    g = np.zeros(nlist.shape, dtype=dt)
    with np.errstate(divide="ignore", invalid="ignore"):
        g[..., :3] = dEdr[..., None] * t / r[..., None]
    if return_grad:
        return nlist_forces_from_grad(nlist, g, E), g
    return nlist_forces_from_grad(nlist, g, E)


def topk_mlp_model(nlist, params, act=None, return_grad=False):
    """Example 08 / build_examples.py:199-218 NlistNN:
        rinv  = nlist_rinv(nlist)
        top_n = tf.sort(rinv, axis=1, direction='DESCENDING')[:, :K]      K = W1.shape[0]
        E_i   = Dense(1)(act(Dense(H2)(act(Dense(H1)(top_n)))))[:, 0]     Keras Dense, activation None by default
        forces = compute_nlist_forces(nlist, energy)
    tf.sort(DESCENDING) runs on tf.math.top_k: equal values keep their index order (stable).  The gradient of a
    sorted value goes back to the slot it came from."""
    dt = nlist.dtype
    s, t, rp, cond = _rinv_and_grad_factor(nlist)
    W1, b1, W2, b2, W3, b3 = (np.asarray(params[k], dtype=dt) for k in ("W1", "b1", "W2", "b2", "W3", "b3"))
    K = W1.shape[0]
    order = np.argsort(-s, axis=1, kind="stable")[:, :K]
    rows = np.arange(s.shape[0])[:, None]
    top = s[rows, order]
    z1 = top @ W1 + b1
    h1 = np.tanh(z1) if act == "tanh" else z1
    z2 = h1 @ W2 + b2
    h2 = np.tanh(z2) if act == "tanh" else z2
    E = (h2 @ W3.reshape(-1, 1) + b3)[:, 0]
    g2 = np.broadcast_to(W3.reshape(-1), h2.shape) * ((1 - h2 * h2) if act == "tanh" else 1)
    g1 = (g2 @ W2.T) * ((1 - h1 * h1) if act == "tanh" else 1)
    gtop = g1 @ W1.T                                  # dE_i / d top_n[i, k]
    dEds = np.zeros_like(s)
    dEds[rows, order] = gtop
    g = _grad_from_dEds(dEds, s, t, rp, cond)
    out = nlist_forces_from_grad(nlist, g, E)
    return (out, g) if return_grad else out


def lj_param_model(nlist, w0, w1):
    """Example 06 TrainableLJ / build_examples.py:336-372 LJLayer on r = safe_norm(x):
    r6 = divide_no_nan(w1**6, r**6); e = w0 * 4 (r6^2 - r6) / 2.  TensorFlow's kernels flush the
    padded slots' r**6 (~2.7e-41, an fp32 denormal) to zero, so divide_no_nan yields 0 there;
    restated as the r > 3e-6 mask.  PARITY UNPINNED (no reference test pins its values)."""
    dt = nlist.dtype
    t = nlist[:, :, :3] + dt.type(1e-7)
    r = np.sqrt(np.sum(t * t, axis=2))
    mask = r > dt.type(RINV_DELTA)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.where(mask, (dt.type(w1) / r) ** 6, dt.type(0))
        e = dt.type(2.0 * w0) * (q * q - q)
        dedr = np.where(mask, dt.type(2.0 * w0) * (dt.type(2) * q - dt.type(1)) * (dt.type(-6) * q / r), dt.type(0))
        g3 = np.where(mask[..., None], dedr[..., None] * t / r[..., None], dt.type(0))
    g = np.zeros(nlist.shape, dtype=dt)
    g[..., :3] = g3
    return nlist_forces_from_grad(nlist, g, e.sum(axis=1))


def mse_loss(pred, labels):
    """Keras 'MeanSquaredError' over the [B, 4] force/energy columns (running.rst:68-71)."""
    d = np.asarray(pred, dtype=np.float64) - np.asarray(labels, dtype=np.float64)
    return float(np.mean(d * d))


def fd_loss_grad(model_of_theta, theta, labels, h=1e-5):
    """Central finite-difference d(MSE)/d(theta): the test-side check of the training kernels."""
    theta = np.asarray(theta, dtype=np.float64)
    g = np.zeros_like(theta)
    for k in range(len(theta)):
        tp, tm = theta.copy(), theta.copy()
        tp[k] += h
        tm[k] -= h
        g[k] = (mse_loss(model_of_theta(tp), labels) - mse_loss(model_of_theta(tm), labels)) / (2 * h)
    return g


class KerasAdam:
    """tf.keras.optimizers.Adam (optimizer_v2) update rule, numpy."""

    def __init__(self, lr=1e-3, b1=0.9, b2=0.999, eps=1e-7):
        self.lr, self.b1, self.b2, self.eps, self.t, self.m, self.v = lr, b1, b2, eps, 0, None, None

    def step(self, theta, g):
        if self.m is None:
            self.m, self.v = np.zeros_like(theta), np.zeros_like(theta)
        self.t += 1
        lr_t = self.lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        self.m += (g - self.m) * (1 - self.b1)
        self.v += (g * g - self.v) * (1 - self.b2)
        return theta - lr_t * self.m / (np.sqrt(self.v) + self.eps)


class KerasNadam:
    """tf.keras.optimizers.Nadam (optimizer_v2) update rule, numpy."""

    def __init__(self, lr=1e-3, b1=0.9, b2=0.999, eps=1e-7):
        self.lr, self.b1, self.b2, self.eps, self.t, self.m, self.v, self.ms = lr, b1, b2, eps, 0, None, None, 1.0

    def step(self, theta, g):
        if self.m is None:
            self.m, self.v = np.zeros_like(theta), np.zeros_like(theta)
        self.t += 1
        t = self.t
        u_t = self.b1 * (1 - 0.5 * 0.96 ** (0.004 * t))
        u_t1 = self.b1 * (1 - 0.5 * 0.96 ** (0.004 * (t + 1)))
        ms_new = self.ms * u_t
        ms_next = ms_new * u_t1
        g_prime = g / (1 - ms_new)
        self.m = self.b1 * self.m + (1 - self.b1) * g
        m_prime = self.m / (1 - ms_next)
        self.v = self.b2 * self.v + (1 - self.b2) * g * g
        v_prime = self.v / (1 - self.b2 ** t)
        m_bar = (1 - u_t) * g_prime + u_t1 * m_prime
        self.ms = ms_new
        return theta - self.lr * m_bar / (np.sqrt(v_prime) + self.eps)


def gauss_pair_terms(nlist, r0, gap):
    """One RBFExpansion channel (layers.py:46-49) on r = safe_norm(x) (simmodel.py:581-594),
    masked with the nlist_rinv criterion (r > 3e-6).  Returns (phi [N,NN], g [N,NN,4]) with
    g = d(sum phi)/d nlist."""
    dt = nlist.dtype
    t = nlist[:, :, :3] + dt.type(1e-7)
    r = np.sqrt(np.sum(t * t, axis=2))
    mask = (r > dt.type(RINV_DELTA)).astype(dt)
    d = r - dt.type(r0)
    phi = np.exp(-(d * d) / dt.type(gap)) * mask
    dphidr = dt.type(-2.0) * d / dt.type(gap) * phi
    g = np.zeros(nlist.shape, dtype=dt)
    with np.errstate(divide="ignore", invalid="ignore"):
        g[..., :3] = np.where(mask[..., None] > 0, dphidr[..., None] * t / r[..., None], dt.type(0))
    return phi, g


def gauss_model(nlist, r0, gap, coef=1.0):
    """E_i = coef * sum_j phi(r_ij); forces = compute_nlist_forces(nlist, E).  PARITY UNPINNED
    (composite of a10 + a16 + a12 defined for config C4)."""
    dt = nlist.dtype
    phi, g = gauss_pair_terms(nlist, r0, gap)
    return nlist_forces_from_grad(nlist, dt.type(coef) * g, dt.type(coef) * phi.sum(axis=1))


def eds_rdf_model(nlist, alpha, r0, gap):
    """Config C4: LJModel + EDS bias on the soft RDF collective variable
        cv = (1/N) sum_i sum_j phi(r_ij),  energy_i = E_lj,i + alpha * cv,
        forces = compute_nlist_forces(nlist, energy).
    alpha is the EDS coupling (a variable read: no gradient flows through it).  The scalar
    alpha*cv is added to every particle's energy, so d(sum_i energy_i) = dE_lj + N alpha dcv.
    Returns (forces [N,4], cv)."""
    dt = nlist.dtype
    N = nlist.shape[0]
    s, t, rp, cond = _rinv_and_grad_factor(nlist)
    inv_r6 = s ** 6
    e_lj = (dt.type(2.0) * (inv_r6 * inv_r6 - inv_r6)).sum(axis=1)
    dEds = dt.type(2.0) * (dt.type(2.0) * inv_r6 - dt.type(1.0)) * (dt.type(6.0) * s ** 5)
    g_lj = _grad_from_dEds(dEds, s, t, rp, cond)
    phi, g_phi = gauss_pair_terms(nlist, r0, gap)
    cv = phi.sum() / dt.type(N)
    a = dt.type(alpha)
    g = g_lj + a * g_phi  # N * alpha * d cv = alpha * d(sum phi)
    return nlist_forces_from_grad(nlist, g, e_lj + a * cv), cv


# --------------------------------------------------------------------------- #
# a8: compute_inputs checks;  a23: compute_outputs;  a6: receiveVirial
# --------------------------------------------------------------------------- #
def check_nlist_count(nlist):
    """simmodel.py:214-219: max_i sum_j [nlist[i,j,0] > 0] (counts only dx > 0 --
    reference quirk); the model asserts this is < NN."""
    return int(np.max(np.sum((nlist[:, :, 0] > 0).astype(np.int32), axis=1)))


def box_is_skewed(box):
    """simmodel.py:195: assert reduce_sum(box[2]) < 0.0001."""
    return not (float(np.sum(box[2])) < 0.0001)


def compute_outputs(forces, hoomd_dtype):
    """simmodel.py:240-255: pad [N,3] -> [N,4] with zeros, cast to HOOMD dtype."""
    if forces.shape[1] == 3:
        forces = np.concatenate([forces, np.zeros((forces.shape[0], 1), forces.dtype)], axis=1)
    return forces.astype(hoomd_dtype)


def receive_virial(dest, src9, pitch, offset, n):
    """TensorflowCompute.cc:284-301: 3x3 row-major -> HOOMD 6-comp SoA, ``+=``."""
    src = src9.reshape(-1, 9)
    for c, col in enumerate((0, 1, 2, 4, 5, 8)):
        dest[c * pitch + offset: c * pitch + offset + n] += src[:n, col]
    return dest


def compute_forces(pos4, types, n_neigh, head_list, nlist, box, r_cut, NN, model,
                   batch_size=0, model_dtype=np.float32, virial=False, periodic=(1, 1, 1),
                   n_local=None):
    """TensorflowCompute.cc:129-216 computeForces in FORCE_MODE::tf2hoomd, one call.

    pos4 is the HOOMD position array [Ntot, >=3] in the HOOMD Scalar dtype.
    ``model(nlist_model_dtype)`` returns forces [B,3|4] (and virial [B,3,3] if
    ``virial``).  Returns (force[N,4] HOOMD dtype, virial6 [6*pitch] or None).
    """
    hd = pos4.dtype
    N = len(n_neigh) if n_local is None else n_local
    force = np.zeros((N, 4), dtype=hd)
    pitch = N
    vir = np.zeros(6 * pitch, dtype=hd) if virial else None
    bs = N if batch_size == 0 else batch_size
    for i in range(N // bs + 1):
        off = i * bs
        n = min(N - off, bs)
        if n < 1:
            break
        nl = prepare_neighbors(pos4[:, :3], types, n_neigh, head_list, nlist, box, r_cut, NN,
                               offset=off, batch_size=n, periodic=periodic)
        out = model(nl.astype(model_dtype))
        if virial:
            f, v = out
            receive_virial(vir, v.astype(hd), pitch, off, n)
        else:
            f = out
        force[off:off + n] = compute_outputs(f, hd)
    return force, vir


# --------------------------------------------------------------------------- #
# a19: EDSLayer (+ TF1 Adam)
# --------------------------------------------------------------------------- #
class EDSLayer:
    """layers.py:101-195, scalar CV.  tf.compat.v1.train.AdamOptimizer(lr):
    beta1=0.9, beta2=0.999, eps=1e-8, lr_t = lr*sqrt(1-b2^t)/(1-b1^t),
    var -= lr_t * m / (sqrt(v) + eps).  STEP-EXACT VALUES: PARITY UNPINNED."""

    def __init__(self, set_point, period, learning_rate=1e-2, cv_scale=1.0, dtype=np.float32):
        self.dt = np.dtype(dtype).type
        self.set_point = self.dt(set_point)
        self.period = int(period)
        self.cv_scale = cv_scale
        self.lr = learning_rate
        d = self.dt
        self.mean, self.ssd, self.alpha = d(0), d(0), d(0)
        self.n = 0
        self.m, self.v, self.t = d(0), d(0), 0

    def __call__(self, cv):
        d = self.dt
        cv = d(cv)
        reset = d(self.n != 0)
        self.mean = self.mean * reset
        self.ssd = self.ssd * reset
        um = d(self.n > self.period // 2)
        delta = (cv - self.mean) * um
        den = d(self.n - self.period // 2)
        self.mean = self.mean + (d(0) if den == 0 else delta / den)
        self.ssd = self.ssd + delta * (cv - self.mean)
        last = self.n == self.period - 1
        grad = d(last) * d(-2) * (self.mean - self.set_point) * self.ssd / d(self.period) / d(2) / d(self.cv_scale)
        if last:
            self.t += 1
            b1, b2, eps = 0.9, 0.999, 1e-8
            lr_t = d(self.lr * math.sqrt(1 - b2 ** self.t) / (1 - b1 ** self.t))
            self.m = self.m + (grad - self.m) * d(1 - b1)
            self.v = self.v + (grad * grad - self.v) * d(1 - b2)
            self.alpha = self.alpha - lr_t * self.m / (np.sqrt(self.v) + d(eps))
        self.n = (self.n + 1) % self.period
        return self.alpha


# --------------------------------------------------------------------------- #
# a20: compute_rdf, masked_nlist
# --------------------------------------------------------------------------- #
def masked_nlist(nlist, type_tensor, type_i=None, type_j=None):
    """simmodel.py:676-693."""
    if type_i is not None:
        nlist = nlist[type_tensor == type_i]
    if type_j is not None:
        mask = (nlist[:, :, 3] == type_j).astype(nlist.dtype)
        nlist = nlist * mask[:, :, None]
    return nlist


def histogram_fixed_width(values, value_range, nbins):
    """tf.histogram_fixed_width (third-party): idx = floor(nbins*(v-lo)/(hi-lo)),
    clipped to [0, nbins-1]."""
    v = values.reshape(-1)
    lo, hi = value_range
    scaled = (v - lo) / (hi - lo)
    idx = np.clip(np.floor(nbins * scaled), 0, nbins - 1).astype(np.int64)
    return np.bincount(idx, minlength=nbins).astype(np.int32)


def compute_rdf(nlist, r_range, type_tensor=None, nbins=100, type_i=None, type_j=None):
    """simmodel.py:638-673.  Returns (rdf[nbins] float32, bin midpoints)."""
    rr = np.asarray(r_range, dtype=np.float32)
    if type_tensor is not None:
        nlist = masked_nlist(nlist, type_tensor, type_i, type_j)
    n3 = nlist[:, :, :3]
    r = np.sqrt(np.sum(n3 * n3, axis=2))
    hist = histogram_fixed_width(r.astype(np.float32), rr, nbins + 2).astype(np.float32)
    shell_rs = np.linspace(rr[0], rr[1], nbins + 1).astype(np.float32)
    vis_rs = (shell_rs[1:] + shell_rs[:-1]) * np.float32(0.5)
    vols = shell_rs[1:] ** 3 - shell_rs[:-1] ** 3
    return hist[1:-1] / vols, vis_rs


# --------------------------------------------------------------------------- #
# utils.compute_nlist (O(N^2) neighbor source with its own KATs)
# --------------------------------------------------------------------------- #
def compute_nlist(positions, r_cut, NN, box_size_, sorted=False, return_types=False):
    """utils.py:75-161.  mask = (dist <= r_cut) & (dist >= 5e-4); sorted=True ->
    nearest first (top_k of -dist, ties to the lower index); sorted=False keeps the
    FARTHEST NN (top_k of masked distances).  w = neighbor index, or type."""
    p3 = positions[:, :3]
    dt = p3.dtype
    M = p3.shape[0]
    dist_mat = p3[None, ...] - p3[:, None, :]
    box = np.asarray(box_size_, dtype=dt).reshape(1, 1, 3)
    dist_mat = dist_mat - np.round(dist_mat / box) * box
    dist = np.sqrt(np.sum(dist_mat * dist_mat, axis=2))
    mask = (dist <= r_cut) & (dist >= 5e-4)
    mc = mask.astype(dt)
    if sorted:
        key = -(dist * mc + (1 - mc) * dt.type(1e20))
    else:
        key = dist * mc
    idx = np.argsort(-key, axis=1, kind="stable")[:, :NN]
    rows = np.arange(M)[:, None]
    npos = dist_mat[rows, idx]
    nmask = mc[rows, idx][..., None]
    if return_types:
        w = positions[:, 3][idx][..., None]
    else:
        w = idx.astype(dt)[..., None]
    return np.concatenate([npos, w], axis=-1) * nmask
