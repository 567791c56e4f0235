"""Third-party arithmetic pinned to what its vendor PUBLISHES.

The reference delegates its arithmetic to TensorFlow (`tf.histogram_fixed_width`, `clip_by_value`,
`divide_no_nan`, `multiply_no_nan`, `math.top_k`, `compat.v1.train.AdamOptimizer`, Keras Adam / Nadam) and
TensorFlow cannot run here.  What CAN be checked without it: the worked examples and update formulas of
TensorFlow's own API documentation (r2.3 / r2.4, the versions the reference's CI pins), and closed forms
derived by hand from the reference's lines -- all written out below as literal numbers or as a few lines
of scalar arithmetic that do not import `oracle/`.  Each vector is asserted against the oracle restatement
(CPU, this file's first half) AND against the HIP kernels (``-m gpu``, second half), so an error shared
by the oracle and the kernels can no longer hide behind their agreement.
"""
import math

import numpy as np
import pytest

from oracle import htf_oracle as O

# ------------------------------------------------------------------------------------------------
# published vectors
# ------------------------------------------------------------------------------------------------
# tf.histogram_fixed_width docstring example: nbins = 5, value_range = [0.0, 5.0],
# new_values = [-1.0, 0.0, 1.5, 2.0, 5.0, 15]  ->  [2, 1, 1, 0, 2]
HIST_VALUES = np.array([-1.0, 0.0, 1.5, 2.0, 5.0, 15.0], dtype=np.float32)
HIST_RANGE = (0.0, 5.0)
HIST_EXPECT = np.array([2, 1, 1, 0, 2])

# tf.clip_by_value docstring example: t = [[-10., -1., 0.], [0., 2., 10.]], clip to [-1, 1]
CLIP_IN = np.array([[-10.0, -1.0, 0.0], [0.0, 2.0, 10.0]])
CLIP_OUT = np.array([[-1.0, -1.0, 0.0], [0.0, 1.0, 1.0]])
# its registered gradient (math_grad.py _ClipByValueGrad): dy where NOT (x < min) and NOT (x > max): the
# bounds themselves pass the gradient
CLIP_GRAD_MASK = np.array([[0.0, 1.0, 1.0], [1.0, 0.0, 0.0]])

# tf.math.top_k docstring: "If two elements are equal, the lower-index element appears first."
TOPK_IN = np.array([1, 2, 98, 1, 1, 99, 3, 1, 3, 96, 4, 1], dtype=np.float64)
TOPK_VALUES, TOPK_INDICES = [99, 98, 96], [5, 2, 9]
TOPK_TIES_IN = np.array([3.0, 7.0, 7.0, 1.0, 7.0])
TOPK_TIES_INDICES = [1, 2, 4]


def tf1_adam_steps(grads, lr, b1=0.9, b2=0.999, eps=1e-8, x0=0.0):
    """tf.compat.v1.train.AdamOptimizer docstring: t <- t + 1; lr_t <- lr * sqrt(1 - b2^t) / (1 - b1^t);
    m_t <- b1 m + (1 - b1) g; v_t <- b2 v + (1 - b2) g^2; variable <- variable - lr_t m_t / (sqrt(v_t) + eps)."""
    x, m, v, out = x0, 0.0, 0.0, []
    for t, g in enumerate(grads, 1):
        lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g * g
        x = x - lr_t * m / (math.sqrt(v) + eps)
        out.append(x)
    return out


def keras_nadam_steps(grads, lr=0.001, b1=0.9, b2=0.999, eps=1e-7, x0=0.0):
    """tf.keras.optimizers.Nadam (optimizer_v2/nadam.py, TF 2.3): momentum schedule
    u_t = b1 (1 - 0.5 * 0.96^(0.004 t)), m_schedule = prod u, Dozat's update."""
    x, m, v, sched, out = x0, 0.0, 0.0, 1.0, []
    for t, g in enumerate(grads, 1):
        u_t = b1 * (1.0 - 0.5 * 0.96 ** (0.004 * t))
        u_t1 = b1 * (1.0 - 0.5 * 0.96 ** (0.004 * (t + 1)))
        sched_new = sched * u_t
        sched_next = sched_new * u_t1
        g_prime = g / (1.0 - sched_new)
        m = b1 * m + (1 - b1) * g
        m_prime = m / (1.0 - sched_next)
        v = b2 * v + (1 - b2) * g * g
        v_prime = v / (1.0 - b2 ** t)
        m_bar = (1.0 - u_t) * g_prime + u_t1 * m_prime
        x = x - lr * m_bar / (math.sqrt(v_prime) + eps)
        sched = sched_new
        out.append(x)
    return out


# One Adam step from zero state: m_1 / (1 - b1) = g and v_1 / (1 - b2) = g^2, so the variable moves by
# lr * g / (|g| + eps / sqrt(1 - b2)) -- almost lr * sign(g) whatever the size of g.
def test_adam_first_step_identity_of_the_formula():
    for g in (0.3, -2.0, 1e-3):
        (x1,) = tf1_adam_steps([g], lr=0.01)
        assert abs(x1 - (-0.01 * g / (abs(g) + 1e-8 / math.sqrt(1 - 0.999)))) < 1e-12


GRADS = [0.5, -0.25, 1.5, 0.75, -2.0, 0.1, 0.1, 0.1]


def eds_trace_by_hand(cvs, set_point, period, lr, cv_scale=1.0):
    """layers.py:159-195, scalar by scalar (python floats; n, mean, ssd, alpha + the TF1 Adam above)."""
    mean = ssd = alpha = 0.0
    n, t, m, v = 0, 0, 0.0, 0.0
    out = []
    for cv in cvs:
        if n == 0:                       # :163-166 reset_mask
            mean, ssd = 0.0, 0.0
        if n > period // 2:              # :168-176 update_mask, Welford with count n - period // 2
            delta = cv - mean
            mean = mean + delta / (n - period // 2)
            ssd = ssd + delta * (cv - mean)
        if n == period - 1:              # :178-189
            grad = -2.0 * (mean - set_point) * ssd / period / 2.0 / cv_scale
            t += 1
            lr_t = lr * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
            m = 0.9 * m + 0.1 * grad
            v = 0.999 * v + 0.001 * grad * grad
            alpha = alpha - lr_t * m / (math.sqrt(v) + 1e-8)
        n = (n + 1) % period             # :191
        out.append(alpha)
    return out


def wca_pair_by_hand(r, sigma):
    """layers.py:91-98 for ONE neighbor at (r, 0, 0), r >> 1e-6 so the safe_norm / rinv deltas (1e-7, 3e-6)
    shift the result by < 1e-5 relative: e = (sigma / r)^6 inside r < 2^(1/3) sigma, clipped to [0, 10];
    dE/dx = -6 sigma^6 / r^7 where 0 <= e <= 10, else 0; model force on the particle = 2 dE/dx
    (compute_nlist_forces, simmodel.py:548), energy column = clipped e (build_examples.py:221-228)."""
    inside = r < sigma * 2 ** (1 / 3)
    e_raw = (sigma / r) ** 6 if inside else 0.0
    e = min(max(e_raw, 0.0), 10.0)
    fx = 2.0 * (-6.0 * sigma ** 6 / r ** 7) if (inside and 0.0 <= e_raw <= 10.0) else 0.0
    return fx, e


WCA_CASES = [(1.0, 1.0), (0.9, 1.0), (1.25, 1.0), (1.27, 1.0), (0.65, 1.0), (0.7, 1.0), (0.45, 0.5), (0.62, 0.5), (0.64, 0.5)]


def rbf_by_hand(x, low, high, count):
    """layers.py:40-49: centers = linspace(low, high, count); gap = centers[1] - centers[0];
    exp(-(x - centers)^2 / gap)  (the gap is NOT squared)."""
    c = [low + (high - low) * k / (count - 1) for k in range(count)]
    gap = c[1] - c[0]
    return [math.exp(-(x - ck) ** 2 / gap) for ck in c]


# ------------------------------------------------------------------------------------------------
# CPU: the oracle restatement against the published vectors
# ------------------------------------------------------------------------------------------------
def test_oracle_histogram_fixed_width_docstring_example():
    np.testing.assert_array_equal(O.histogram_fixed_width(HIST_VALUES, HIST_RANGE, 5), HIST_EXPECT)


def test_oracle_compute_rdf_uses_that_histogram():
    # compute_rdf (simmodel.py:661-668) = histogram over nbins + 2 bins, end bins dropped, shell volumes of
    # linspace(r0, r1, nbins + 1).  One neighbor per value, along x; with nbins = 3 the histogram above has
    # 5 bins and its inner three counts are [1, 1, 0].
    nl = np.zeros((1, 6, 4), np.float32)
    nl[0, :, 0] = HIST_VALUES + 1.0          # shifted by +1 so that "-1.0" is a legal distance (range [1, 6])
    rdf, rs = O.compute_rdf(nl, [1.0, 6.0], nbins=3)
    shell = np.linspace(1.0, 6.0, 4, dtype=np.float32)
    np.testing.assert_allclose(rdf, np.array([1, 1, 0], np.float32) / (shell[1:] ** 3 - shell[:-1] ** 3), rtol=1e-6)
    np.testing.assert_allclose(rs, (shell[1:] + shell[:-1]) / 2, rtol=1e-6)


def test_oracle_wca_clip_follows_clip_by_value():
    # forward values of the docstring example through numpy.clip (what the oracle calls) ...
    np.testing.assert_array_equal(np.clip(CLIP_IN, -1.0, 1.0), CLIP_OUT)
    # ... and the inclusive-bound gradient rule inside the oracle's WCA model: e exactly 10 still passes.
    # sigma s = 10^(1/6) makes (sigma s)^6 = 10 up to rounding; probe one ulp either side in fp64.
    for scale, passes in ((1.0 - 1e-9, True), (1.0 + 1e-9, False)):
        r = 1.0 / (10 ** (1 / 6) * scale) - (1e-7 + 3e-6)     # (1 / (r + deltas))^6 = 10 * scale^6
        nl = np.zeros((1, 1, 4))
        nl[0, 0, 0] = r
        f = O.wca_model(nl, 1.0)
        assert (f[0, 0] != 0.0) == passes
        assert abs(f[0, 3] - min((1.0 / (r + 1e-7 + 3e-6)) ** 6, 10.0)) < 1e-4


@pytest.mark.parametrize("r,sigma", WCA_CASES)
def test_oracle_wca_closed_form(r, sigma):
    nl = np.zeros((1, 4, 4))
    nl[0, 0, 0] = r
    f = O.wca_model(nl, sigma)
    fx, e = wca_pair_by_hand(r, sigma)
    np.testing.assert_allclose(f[0, 0], fx, rtol=5e-5, atol=1e-9)
    np.testing.assert_allclose(f[0, 3], e, rtol=5e-5, atol=1e-9)
    assert f[0, 1] == pytest.approx(0.0, abs=1e-4 * max(1.0, abs(fx))) and f[0, 2] == pytest.approx(0.0, abs=1e-4 * max(1.0, abs(fx)))


def test_oracle_rbf_closed_form():
    xs = np.array([0.0, 0.35, 1.0, 2.0], dtype=np.float64)
    got = O.rbf_expansion(xs, 0.0, 2.0, 10)
    for i, x in enumerate(xs):
        np.testing.assert_allclose(got[i], rbf_by_hand(float(x), 0.0, 2.0, 10), rtol=1e-6)
    assert got[0, 0] == 1.0 and got[3, 9] == pytest.approx(1.0, abs=1e-6)   # on a centre


def test_oracle_divide_and_multiply_no_nan():
    # tf.math.divide_no_nan: "returns 0 if the denominator is zero"; _compute_virial (simmodel.py:518) relies
    # on it for padded slots: |nf| / (2 |r|) with r = 0 -> 0, so a padded slot adds nothing to the virial
    nl = np.zeros((1, 2, 4))
    nl[0, 0, :3] = (1.0, 0.0, 0.0)
    nf = np.zeros((1, 2, 4))
    nf[0, 0, 0] = -3.0
    nf[0, 1, 0] = 5.0          # a force on a padded slot (cannot happen upstream, but must not produce NaN)
    v = O.compute_virial(nl, nf)
    assert np.all(np.isfinite(v))
    np.testing.assert_allclose(v[0, 0, 0], -(3.0 / 2.0) * 1.0)   # -(|nf| / (2 |r|)) x x
    assert np.count_nonzero(v) == 1


def test_oracle_top_k_tie_order():
    # compute_nlist(sorted=False) keeps the top_k of the masked distances: ties go to the LOWER index
    order = np.argsort(-TOPK_IN, kind="stable")[:3]
    assert list(order) == TOPK_INDICES and list(TOPK_IN[order]) == TOPK_VALUES
    assert list(np.argsort(-TOPK_TIES_IN, kind="stable")[:3]) == TOPK_TIES_INDICES
    # the oracle on a configuration with exactly tied distances: 4 neighbors at distance 1 of particle 0, NN = 2
    pos = np.array([[0, 0, 0, 0], [1, 0, 0, 0], [-1, 0, 0, 0], [0, 1, 0, 0], [0, -1, 0, 0]], dtype=np.float64)
    nl = O.compute_nlist(pos, 1.5, 2, [10.0, 10.0, 10.0], sorted=True)
    assert list(nl[0, :, 3]) == [1.0, 2.0]   # lower indices first among the tie


def test_oracle_optimizers_follow_the_documented_formulas():
    # Keras Adam = the TF1 formula with eps 1e-7 (optimizer_v2/adam.py docstring, non-amsgrad branch)
    want = tf1_adam_steps(GRADS, lr=0.001, eps=1e-7, x0=0.2)
    opt, x = O.KerasAdam(1e-3), np.array([0.2])
    for g, w in zip(GRADS, want):
        x = opt.step(x, np.array([g]))
        assert x[0] == pytest.approx(w, rel=1e-12, abs=1e-15)
    want = keras_nadam_steps(GRADS, lr=0.001, x0=0.2)
    opt, x = O.KerasNadam(1e-3), np.array([0.2])
    for g, w in zip(GRADS, want):
        x = opt.step(x, np.array([g]))
        assert x[0] == pytest.approx(w, rel=1e-12, abs=1e-15)


CVS = [4.3, 4.1, 3.6, 3.9, 4.4, 4.8, 3.7, 4.05, 4.2, 3.95, 4.6, 3.5, 4.15, 4.0, 4.25, 3.85, 4.5, 3.75, 4.1, 4.3]


@pytest.mark.parametrize("period,lr", [(5, 0.2), (6, 0.05), (7, 1.0)])
def test_oracle_eds_trace_by_hand(period, lr):
    want = eds_trace_by_hand(CVS * 3, 4.0, period, lr)
    eds = O.EDSLayer(4.0, period, lr, dtype=np.float64)
    got = [float(eds(cv)) for cv in CVS * 3]
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-14)
    assert abs(want[-1]) > 1e-3            # the coupling constant has moved


# ------------------------------------------------------------------------------------------------
# GPU: the HIP kernels against the same vectors (through the C ABI)
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_hip_histogram_docstring_example(htf, cuda):
    import ctypes as C
    import torch
    from hoomd_tf_amd._lib import lib, check
    nl = torch.zeros((1, 6, 4), dtype=torch.float32, device=cuda)
    nl[0, :, 0] = torch.from_numpy(HIST_VALUES + 1.0).to(cuda)
    hist = torch.zeros(5, dtype=torch.int32, device=cuda)
    check(lib.htf_rdf_histogram(nl.data_ptr(), 0, 1, 6, 1.0, 6.0, 5, None, 0, -1, -1, hist.data_ptr(),
                                C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    np.testing.assert_array_equal(hist.cpu().numpy(), HIST_EXPECT)
    # and through the Python surface (htf.compute_rdf, simmodel.py:638-673)
    rdf, rs = htf.compute_rdf(htf.Nlist(nl) if hasattr(htf, "Nlist") else nl, [1.0, 6.0], nbins=3)
    shell = np.linspace(1.0, 6.0, 4, dtype=np.float32)
    r = rdf.tensor() if hasattr(rdf, "tensor") and callable(rdf.tensor) else rdf
    np.testing.assert_allclose(r.cpu().numpy(), np.array([1, 1, 0], np.float32) / (shell[1:] ** 3 - shell[:-1] ** 3), rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("r,sigma", WCA_CASES)
def test_hip_wca_closed_form(htf, cuda, r, sigma, fused):
    import torch
    fx, e = wca_pair_by_hand(r, sigma)
    pot = htf.Potential.wca(sigma)
    if not fused:
        nl = torch.zeros((1, 4, 4), dtype=torch.float32, device=cuda)
        nl[0, 0, 0] = r
        f = htf.ops.eval_forces(pot, nl).cpu().numpy()[0]
    else:  # the one-kernel gather-evaluate path on a two-particle system
        pos = torch.tensor([[0.0, 0, 0, 0], [r, 0, 0, 0]], dtype=torch.float32, device=cuda)
        nn = torch.tensor([1, 1], dtype=torch.int32, device=cuda)
        head = torch.tensor([0, 1], dtype=torch.int32, device=cuda)
        nlist = torch.tensor([1, 0], dtype=torch.int32, device=cuda)
        box = O.make_box([20.0, 20.0, 20.0])
        f = htf.ops.fused_forces(pot, pos, nn, head, nlist, box, 3.0, 4).cpu().numpy()[0]
    np.testing.assert_allclose(f[0], fx, rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(f[3], e, rtol=1e-4, atol=1e-7)
    assert abs(f[1]) <= 1e-4 * max(1.0, abs(fx)) and abs(f[2]) <= 1e-4 * max(1.0, abs(fx))


@pytest.mark.gpu
def test_hip_wca_clip_gradient_at_the_bound(htf, cuda):
    import torch
    # fp32: walk r across e = (1/(r + deltas))^6 = 10 and require force != 0 exactly where the kernel's own
    # energy column is <= 10 before clipping, i.e. where it reports e < 10 or e == 10 from below
    r0 = 1.0 / 10 ** (1 / 6)
    rs = np.float32(r0) + np.arange(-40, 41, dtype=np.float32) * np.float32(2e-7)
    nl = torch.zeros((len(rs), 1, 4), dtype=torch.float32, device=cuda)
    nl[:, 0, 0] = torch.from_numpy(rs).to(cuda)
    f = htf.ops.eval_forces(htf.Potential.wca(1.0), nl).cpu().numpy()
    ref = O.wca_model(nl.cpu().numpy().astype(np.float32), 1.0)  # fp32 oracle: same rounding class
    e_unclipped = (1.0 / (rs.astype(np.float64) + 1e-7 + 3e-6)) ** 6
    far = np.abs(e_unclipped - 10.0) > 1e-4                     # away from the edge the verdict is unambiguous
    assert np.array_equal(f[far, 0] != 0.0, e_unclipped[far] <= 10.0)
    assert np.all(f[:, 3] <= 10.0) and np.all(ref[:, 3] <= 10.0)
    assert (f[:, 0] != 0).any() and (f[:, 0] == 0).any()


@pytest.mark.gpu
def test_hip_rbf_closed_form(htf, cuda):
    import torch
    xs = torch.tensor([0.0, 0.35, 1.0, 2.0], dtype=torch.float32, device=cuda)
    layer = htf.RBFExpansion(0.0, 2.0, 10)
    got = layer(xs)
    got = (got.tensor() if hasattr(got, "tensor") and callable(got.tensor) else got).cpu().numpy()
    for i, x in enumerate(xs.cpu().numpy()):
        np.testing.assert_allclose(got[i], rbf_by_hand(float(x), 0.0, 2.0, 10), rtol=2e-6, atol=1e-7)


@pytest.mark.gpu
def test_hip_virial_divide_no_nan(htf, cuda):
    import torch
    # one real neighbor + padding: the padded slot must add exactly nothing (divide_no_nan), nothing is NaN
    nl = torch.zeros((1, 4, 4), dtype=torch.float32, device=cuda)
    nl[0, 0, 0] = 1.5
    f, v = htf.ops.eval_forces(htf.Potential.lj(), nl, virial=True)
    v = v.cpu().numpy()[0]
    assert np.all(np.isfinite(v))
    # attractive pair at r = 1.5: nlist force fx = (-48 s^13 + 24 s^7) -> virial xx = -(|fx| / (2 r)) r^2
    s = 1.0 / (1.5 + 1e-7 + 3e-6)
    fx = -48 * s ** 13 + 24 * s ** 7
    np.testing.assert_allclose(f.cpu().numpy()[0, 0], fx, rtol=1e-4)
    np.testing.assert_allclose(v[0, 0], -(abs(fx) / (2 * 1.5)) * 1.5 * 1.5, rtol=1e-4)
    assert np.count_nonzero(np.abs(v) > 1e-6 * abs(v[0, 0])) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["adam", "nadam"])
def test_hip_optimizer_documented_formulas(htf, cuda, kind):
    import torch
    want = (tf1_adam_steps(GRADS, lr=0.001, eps=1e-7, x0=0.2) if kind == "adam" else keras_nadam_steps(GRADS, lr=0.001, x0=0.2))
    opt = (htf.optimizers.Adam(1e-3) if kind == "adam" else htf.optimizers.Nadam(1e-3)).desc(0, (0.0,))
    theta = torch.tensor([0.2], dtype=torch.float32, device=cuda)
    state = torch.zeros(htf.ops.optimizer_state_floats(1), dtype=torch.float32, device=cuda)
    for g, w in zip(GRADS, want):
        accum = torch.tensor([0.0, g], dtype=torch.float32, device=cuda)   # [loss sum, d/dtheta]
        htf.ops.optimizer_step(theta, accum, 1.0, state, opt)
        assert float(theta) == pytest.approx(w, rel=2e-6, abs=2e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("period,lr", [(5, 0.2), (6, 0.05), (7, 1.0)])
def test_hip_eds_trace_by_hand(htf, cuda, period, lr):
    import torch
    want = eds_trace_by_hand(CVS * 3, 4.0, period, lr)
    eds = htf.EDSLayer(4.0, period, lr, device=cuda)
    got = []
    for cv in CVS * 3:
        a = eds(torch.tensor(cv, dtype=torch.float32, device=cuda))
        got.append(float(a.tensor() if hasattr(a, "tensor") and callable(a.tensor) else a))
    scale = max(abs(w) for w in want)
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-5 * scale)


@pytest.mark.gpu
def test_hip_top8_tie_order_is_value_stable(htf, cuda):
    """example 08 sorts rinv descending and keeps 8 (tf.sort + slice): ties can only occur among the padded
    zeros, whose gradient is zero, so any order among equal values gives the same forces; the VALUES of
    the top k must equal the documented top_k example."""
    import torch
    if not hasattr(htf.ops, "topk_desc"):
        pytest.skip("top-k kernel not built")
    x = torch.tensor(TOPK_IN[None, :], dtype=torch.float32, device=cuda)
    vals, idx = htf.ops.topk_desc(x, 3)
    assert vals.cpu().numpy()[0].tolist() == TOPK_VALUES and idx.cpu().numpy()[0].tolist() == TOPK_INDICES
    vals, idx = htf.ops.topk_desc(torch.tensor(TOPK_TIES_IN[None, :], dtype=torch.float32, device=cuda), 3)
    assert idx.cpu().numpy()[0].tolist() == TOPK_TIES_INDICES
