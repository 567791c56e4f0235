"""Graph-style oracle: the reference's TF2 op sequence transcribed op-for-op to
torch-CPU, with ``torch.autograd`` standing in for ``tf.gradients``.

TEST INFRASTRUCTURE ONLY (same rule as htf_oracle.py).  Two uses:
  1. independent check of the hand-derived gradients in htf_oracle.py
     (tests/test_oracle.py);
  2. the "graph-style CPU restatement" timing of BASELINE.md section 3 (same cost
     structure as TF2: one pass over [N,NN] per op, forward + backward).

TF/torch sub-gradient conventions that matter here: tf.where and
torch.where both route zero gradient to the unselected branch; clip_by_value and
torch.clamp both pass the gradient on the closed interval; multiply_no_nan(g, 2)
== 2g.  Plain tf.norm at exactly 0 is NaN in TF and 0 in torch -- only reachable in
reference models that are out of scope, never in the ones below
because the delta is added before the norm.
"""
import torch

RINV_DELTA = 3e-6


def safe_norm(t, delta=1e-7, dim=-1):
    # simmodel.py:594  tf.norm(tensor + delta)
    u = t + delta
    return torch.sqrt(torch.sum(u * u, dim=dim))


def nlist_rinv(nlist):
    # simmodel.py:627-635
    delta = RINV_DELTA
    r = safe_norm(nlist[:, :, :3], delta=delta / 3 / 10, dim=2)
    return torch.where(r > delta, 1.0 / (r + delta), torch.zeros_like(r))


def _add_energy(forces, energy):
    # simmodel.py:558-578
    N = forces.shape[0]
    if energy.dim() > 1:
        e = energy.reshape(N, -1).sum(dim=1, keepdim=True)
    elif energy.dim() == 0:
        e = energy.reshape(1).repeat(N).reshape(N, 1)
    else:
        e = energy.reshape(N, 1)
    return torch.cat([forces[:, :3], e], dim=-1)


def _compute_virial(nlist, nlist_forces):
    # simmodel.py:509-523
    n3 = nlist[:, :, :3]
    outer = torch.einsum("ijk,ijl->ijkl", n3, n3)
    rmag = torch.sqrt(torch.sum(n3 * n3, dim=2))
    fmag = torch.sqrt(torch.sum(nlist_forces * nlist_forces, dim=2))
    den = 2.0 * rmag
    F_rs = torch.where(den == 0, torch.zeros_like(den), fmag / den)
    return -1.0 * torch.einsum("ij,ijkl->ikl", F_rs, outer)


def compute_nlist_forces(nlist, energy, virial=False):
    # simmodel.py:526-555
    (g,) = torch.autograd.grad(energy.sum(), nlist, create_graph=False)
    nf = g * 2.0
    red = nf.sum(dim=1)
    if virial:
        return _add_energy(red, energy.detach()), _compute_virial(nlist.detach(), nf)
    return _add_energy(red, energy.detach())


def lj_model(nlist, virial=False):
    # build_examples.py:67-77 / :104-115
    nlist = nlist.clone().requires_grad_(True)
    rinv = nlist_rinv(nlist)
    inv_r6 = rinv ** 6
    p_energy = 4.0 / 2.0 * (inv_r6 * inv_r6 - inv_r6)
    energy = p_energy.sum(dim=1)
    return compute_nlist_forces(nlist, energy, virial)


def benchmark_potential(nlist):
    # build_examples.py:25-30
    nlist = nlist.clone().requires_grad_(True)
    return compute_nlist_forces(nlist, nlist_rinv(nlist))


def wca_model(nlist, sigma=0.5):
    # layers.py:91-98 + build_examples.py:221-228
    nlist = nlist.clone().requires_grad_(True)
    rinv = nlist_rinv(nlist)
    sig = torch.tensor(sigma, dtype=nlist.dtype)
    rp = (sig * rinv) ** 6
    n3 = nlist[:, :, :3]
    r = torch.sqrt(torch.sum(n3 * n3, dim=2)).detach()
    e = (r < sig * 2 ** (1 / 3)).to(nlist.dtype) * rp
    e = torch.clamp(e, 0, 10)
    return compute_nlist_forces(nlist, e)


def rbf_expansion(x, low, high, count):
    # layers.py:31-34,46-49
    centers = torch.linspace(float(low), float(high), count, dtype=torch.float32)
    gap = centers[1] - centers[0]
    return torch.exp(-(x[..., None] - centers.to(x.dtype)) ** 2 / gap.to(x.dtype))


def pair_mlp_model(nlist, params, low=0.0, high=3.0, act="tanh"):
    # composite defined in SURVEY 8(a); see htf_oracle.pair_mlp_model
    nlist = nlist.clone().requires_grad_(True)
    dt = nlist.dtype
    P = {k: torch.as_tensor(v).to(dt) for k, v in params.items()}
    r = safe_norm(nlist[:, :, :3], dim=2)
    phi = rbf_expansion(r, low, high, P["W1"].shape[0])
    f = torch.tanh if act == "tanh" else (lambda z: z)
    h1 = f(phi @ P["W1"] + P["b1"])
    h2 = f(h1 @ P["W2"] + P["b2"])
    u = (h2 @ P["W3"] + P["b3"])[..., 0]
    mask = (r > RINV_DELTA).to(dt).detach()
    energy = 0.5 * (u * mask).sum(dim=1)
    return compute_nlist_forces(nlist, energy)


def eds_rdf_model(nlist, alpha, r0, gap):
    # config C4: see htf_oracle.eds_rdf_model
    nlist = nlist.clone().requires_grad_(True)
    dt = nlist.dtype
    rinv = nlist_rinv(nlist)
    inv_r6 = rinv ** 6
    lj = (4.0 / 2.0 * (inv_r6 * inv_r6 - inv_r6)).sum(dim=1)
    r = safe_norm(nlist[:, :, :3], dim=2)
    mask = (r > RINV_DELTA).to(dt).detach()
    phi = torch.exp(-(r - r0) ** 2 / gap) * mask
    cv = phi.sum() / nlist.shape[0]
    energy = lj + alpha * cv
    return compute_nlist_forces(nlist, energy), cv.detach()


def lj_param_forces(nlist, w, create_graph=False):
    # example 06: LJLayer on safe_norm; divide_no_nan with TF's flush-to-zero == mask r > 3e-6
    if not nlist.requires_grad:
        nlist = nlist.clone().requires_grad_(True)
    r = safe_norm(nlist[:, :, :3], dim=2)
    mask = (r > RINV_DELTA)
    rs = torch.where(mask, r, torch.ones_like(r))
    r6 = torch.where(mask, w[1] ** 6 / rs ** 6, torch.zeros_like(r))
    energy = (w[0] * 4.0 * (r6 ** 2 - r6) / 2.0).sum(dim=1)
    (g,) = torch.autograd.grad(energy.sum(), nlist, create_graph=create_graph)
    return _add_energy((g * 2.0).sum(dim=1), energy)


def mse_grad_wrt_params(forces_fn, nlist, labels, theta):
    """d(MeanSquaredError over [B,4])/d(theta) by double backward: what Keras' train_on_batch
    computes for a model whose output comes from compute_nlist_forces."""
    w = torch.tensor(theta, dtype=nlist.dtype, requires_grad=True)
    pred = forces_fn(nlist.clone().requires_grad_(True), w)
    loss = ((pred - labels) ** 2).mean()
    (g,) = torch.autograd.grad(loss, w)
    return float(loss.detach()), g.numpy()


def wca_param_forces(nlist, w, create_graph=False):
    # WCARepulsion with sigma = w[0] as a differentiable parameter (layers.py:91-98)
    if not nlist.requires_grad:
        nlist = nlist.clone().requires_grad_(True)
    rinv = nlist_rinv(nlist)
    rp = (w[0] * rinv) ** 6
    n3 = nlist[:, :, :3]
    r = torch.sqrt(torch.sum(n3 * n3, dim=2)).detach()
    e = (r < w[0].detach() * 2 ** (1 / 3)).to(nlist.dtype) * rp  # float cast of a comparison: no gradient
    e = torch.clamp(e, 0, 10)
    (g,) = torch.autograd.grad(e.sum(), nlist, create_graph=create_graph)
    return _add_energy((g * 2.0).sum(dim=1), e)


def pair_mlp_param_forces(nlist, w, dims, low=0.0, high=3.0, act="tanh", create_graph=False):
    """pair_mlp_model with the weights as ONE flat differentiable vector in Keras get_weights()
    order (W1 [K,H1] | b1 | W2 [H1,H2] | b2 | W3 [H2] | b3): the training reference."""
    K, H1, H2 = dims
    if not nlist.requires_grad:
        nlist = nlist.clone().requires_grad_(True)
    o = 0
    W1 = w[o:o + K * H1].reshape(K, H1); o += K * H1
    b1 = w[o:o + H1]; o += H1
    W2 = w[o:o + H1 * H2].reshape(H1, H2); o += H1 * H2
    b2 = w[o:o + H2]; o += H2
    W3 = w[o:o + H2]; o += H2
    b3 = w[o]
    r = safe_norm(nlist[:, :, :3], dim=2)
    phi = rbf_expansion(r, low, high, K)
    f = torch.tanh if act == "tanh" else (lambda z: z)
    h1 = f(phi @ W1 + b1)
    h2 = f(h1 @ W2 + b2)
    u = h2 @ W3 + b3
    mask = (r > RINV_DELTA).to(nlist.dtype).detach()
    energy = 0.5 * (u * mask).sum(dim=1)
    (g,) = torch.autograd.grad(energy.sum(), nlist, create_graph=create_graph)
    return _add_energy((g * 2.0).sum(dim=1), energy)
