"""Drives integration/hoomd_shim/ (TensorflowComputeAMD, the HOOMD-side ForceCompute) against the FAKE HOOMD of
integration/hoomd_stub/ on a GPU and compares every array it leaves in "HOOMD's" m_force / m_virial with the
htf.Context path (the C ABI called from Python) BIT FOR BIT.  Run by tests/test_gpu_shim.py in a child process:
    python tests/shim_driver.py <dir with _htf_amd.so and _hoomd_stub.so> <double|single>
What is exercised is the reference's per-step body, htf/TensorflowCompute.cc:129-216 and :250-301:
period gate, nlist->compute, updateBox, batch loop, forces written into m_force, virial folded into m_virial at its
pitch, reference-force labels, the training step, MaxParticleNumberChange, the log quantity, the half-step hook.
This is a fake HOOMD (no integrator, no cell list of its own): the stand-in driver moves the particles and builds
the neighbor list, and the fake's ParticleData / NeighborList point at those device arrays."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hoomd_tf_amd as htf  # noqa: E402
from hoomd_tf_amd import _lib, ops, optimizers, standin  # noqa: E402
from oracle import htf_oracle as O  # noqa: E402  (the checker; this file is test infrastructure)

sys.path.insert(0, sys.argv[1])
import _hoomd_stub as H  # noqa: E402
import _htf_amd as M  # noqa: E402

single = sys.argv[2] == "single"
sdt = torch.float32 if single else torch.float64
assert H.scalar_bytes == (4 if single else 8)
dev = torch.device("cuda:0")
R_CUT, NN = 2.5, 64


def view(ptr, shape, dtype):
    class _H:
        pass
    h = _H()
    h.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4" if dtype == torch.float32 else "<f8",
                                  "data": (int(ptr), False), "version": 2, "strides": None}
    return torch.as_tensor(h, device=dev)


class FakeHoomdRun:
    """One system: stand-in particles + cell list, the fake HOOMD objects pointed at them, the shim on top."""

    def __init__(self, cells=6, period=1, batch_size=0, mode="tf2hoomd", seed=1):
        pos, L, a = standin.fcc_positions(cells, 0.8442)
        rng = np.random.default_rng(seed)
        pos = pos + 0.03 * a * rng.standard_normal(pos.shape)
        pos -= np.round(pos / L) * L
        self.sysm = s = standin.System(pos, L, dtype=sdt, device=dev)
        s.randomize_velocities(kT=1.0, seed=seed)
        self.nl = standin.CellNlist(s, r_cut=R_CUT, r_buff=0.4, check_period=1, device_decision=False)
        self.nl.build()
        self.nve = standin.NVE(s, 0.005)
        self.sysdef = H.SystemDefinition()
        self.pdata = self.sysdef.getParticleData()
        self.max_n = s.N + 4                      # HOOMD's max N is rarely N: the virial pitch then differs from N too
        self.pdata.setN(s.N, self.max_n, 0)
        half = [float(x) / 2 for x in L]
        self.pdata.setBox([-h for h in half], half, [0.0, 0.0, 0.0], [1, 1, 1])
        self.net_force = torch.zeros((s.N, 4), dtype=sdt, device=dev)
        self.pdata.setNetForcePtr(self.net_force.data_ptr(), s.N)
        self.hnl = H.NeighborList()
        self.hnl.onCompute(self._nlist_compute)
        self._point()
        self.period, self.batch_size = period, batch_size
        self.c = M.TensorflowComputeAMD(self, self.sysdef, self.hnl, R_CUT, NN,
                                        M.FORCE_MODE.tf2hoomd if mode == "tf2hoomd" else M.FORCE_MODE.hoomd2tf, period, batch_size)
        assert self.hnl.isFull()                      # the constructor swaps a half list to full (.cc:73-84)
        assert self.c.isDoublePrecision() == (not single)

    def _point(self):
        s, nl = self.sysm, self.nl
        self.pdata.setPositionsPtr(s.pos.data_ptr(), s.pos.shape[0])
        self.hnl.setArrays(nl.n_neigh.data_ptr(), nl.nlist.data_ptr(), nl.head_list.data_ptr(), s.N, nl.nlist.numel())

    def _nlist_compute(self, timestep):  # what hoomd.md.nlist does behind m_nlist->compute(timestep)
        nb = self.nl.n_builds
        self.nl.compute(timestep)
        if self.nl.n_builds != nb:
            self._point()

    # the two callbacks the reference's protocol makes into tensorflowcompute.py (not reached with a lowered potential)
    def _start_update(self):
        raise AssertionError("no mapped nlist in this test")

    def _finish_update(self, batch):
        raise AssertionError("a lowered potential is installed: Python is not in the step loop")

    def force(self):
        return view(self.c.forcePtr(), (self.sysm.N, 4), sdt)

    def virial6(self):
        pitch = self.c.virialPitch()
        return view(self.c.virialPtr(), (6, pitch), sdt), pitch


def oracle_check(run, virial, batch_size, what):
    """What the shim left in "HOOMD's" m_force (and, from a zeroed m_virial, in m_virial) against the CPU oracle's
    computeForces (oracle/htf_oracle.py, TensorflowCompute.cc:129-216) on the same positions and list: SURVEY 8(c)'s bound
    as stated, plus the named condition term for the virial's cancelling row sums."""
    s, nlc = run.sysm, run.nl
    hd = np.float32 if single else np.float64
    pos = s.pos.cpu().numpy()[:, :3].astype(hd)
    L = [float(s.box3x3[1][i] - s.box3x3[0][i]) for i in range(3)]
    model = (lambda x: O.lj_model(x.astype(np.float64), virial=True)) if virial else (lambda x: O.lj_model(x.astype(np.float64)))
    ref, vref = O.compute_forces(pos, np.zeros(s.N, np.int32), nlc.n_neigh.cpu().numpy().view(np.uint32),
                                 nlc.head_list.cpu().numpy().view(np.uint32), nlc.nlist.cpu().numpy().view(np.uint32),
                                 O.make_box(L, dtype=hd), R_CUT, NN, model, batch_size=batch_size, model_dtype=np.float32, virial=virial)
    got = run.force().cpu().numpy().astype(np.float64)
    err = np.abs(got - ref)
    # sum_j |f_ij| per row, the scale two fp32 summation orders differ on (tests/test_gpu_parity.py: cancelling_rows)
    pv = O.prepare_neighbors(pos, np.zeros(s.N, np.int32), nlc.n_neigh.cpu().numpy().view(np.uint32),
                             nlc.head_list.cpu().numpy().view(np.uint32), nlc.nlist.cpu().numpy().view(np.uint32),
                             O.make_box(L, dtype=hd), R_CUT, NN).astype(np.float32).astype(np.float64)
    sg, tg, rpg, cg = O._rinv_and_grad_factor(pv)
    cond = np.abs(2.0 * O._grad_from_dEds(2.0 * (2.0 * sg ** 6 - 1.0) * (6.0 * sg ** 5), sg, tg, rpg, cg)).sum(axis=(1, 2))[:, None]
    assert np.all(err[:, 3] <= 1e-5 + 2e-5 * np.abs(ref[:, 3])), (what, "energies vs oracle, as stated", float(err[:, 3].max()))
    assert np.all(err <= 1e-5 + 2e-5 * np.abs(ref) + 2e-6 * cond), (what, "forces vs oracle", float(err.max()))
    if virial:
        v6, pitch = run.virial6()
        gotv = v6.cpu().numpy().astype(np.float64)[:, :s.N]
        refv = np.asarray(vref, dtype=np.float64).reshape(6, s.N)
        scale = np.abs(refv).max()
        assert np.all(np.abs(gotv - refv) <= 2e-5 + 5e-5 * np.abs(refv) + 2e-6 * scale), (what, "virial vs oracle", float(np.abs(gotv - refv).max()))
    return float(err.max())


def context_for(run, virial, batch_size=0, period=1):
    ctx = htf.Context(r_cut=R_CUT, nneighs=NN, period=period, batch_size=batch_size, scalar_dtype=sdt, virial=virial,
                      max_n=run.sysm.N, fused=2)
    return ctx


# ------------------------------------------------------------------ 1. ten MD steps, forces + virial, bit for bit
for virial, batch_size, period in ((True, 0, 1), (False, 0, 1), (True, 300, 1), (False, 0, 3)):
    run = FakeHoomdRun(period=period, batch_size=batch_size)
    s = run.sysm
    pot = htf.Potential.lj()
    run.c.setPotential(pot.handle.value, virial, False, 2)
    ctx = context_for(run, virial, batch_size, period)
    ctx.set_potential(pot)
    f_ref = torch.zeros((s.N, 4), dtype=sdt, device=dev)
    v6, pitch = run.virial6()
    assert pitch > s.N and pitch % 16 == 0 and run.c.getVirialPitch() == pitch   # 864 particles, max N 868: pitch 880
    v_ref = torch.zeros((6, pitch), dtype=sdt, device=dev)
    calls0 = run.hnl.computeCalls()
    for ts in range(10):
        run.c.compute(ts)                                       # ForceCompute::compute -> computeForces
        nlc = run.nl
        ctx.compute_forces(ts, ctx.make_arrays(s.pos, s.N, nlc.n_neigh, nlc.head_list, nlc.nlist, s.box, f_ref,
                                               v_ref if virial else None, pitch))
        torch.cuda.synchronize()
        assert torch.equal(run.force(), f_ref), ("forces differ", virial, batch_size, period, ts)
        if virial:
            assert torch.equal(v6, v_ref), ("virial differs", batch_size, ts)   # '+=' and never reset (.cc:284-301)
        if ts % period == 0:
            assert float(run.force().abs().max()) > 0
        # the integrator's job: HOOMD's net force is what moves the particles
        s.force.copy_(run.force())
        run.nve.step()
    assert run.hnl.computeCalls() - calls0 == len([t for t in range(10) if t % period == 0])  # period gate before nlist->compute
    if virial:
        assert float(v6[:, :s.N].abs().max()) > 0 and float(v6[:, s.N:].abs().max()) == 0   # nothing beyond N inside the pitch
    # host copies the reference exports (.cc:409-420)
    fa = np.array([[f.x, f.y, f.z, f.w] for f in run.c.getForcesArray()])
    assert np.array_equal(fa, run.force().cpu().numpy())
    ba = run.c.getBoxArray()
    assert abs(ba[1].x - float(s.box3x3[1][0])) < 1e-6 and ba[2].x == 0.0
    n_b = s.N if batch_size == 0 else min(batch_size, s.N)
    assert run.c.getBatchCapacity() >= n_b
    pv = view(run.c.getNlistBuffer(), (n_b, NN, 4), torch.float32)
    assert torch.equal(pv, ctx.nlist_buffer(n_b, dev))          # the side buffer getNlistBuffer() hands out
    assert abs(run.c.getLogValue("tensorflow", 9) - float(run.force()[:, 3].double().sum())) < 1e-6 * s.N
    # self-standing: one more compute on the moved particles, from a zeroed m_virial, against the CPU oracle
    if virial:
        v6.zero_()
    ts = 10 + (-10) % period
    run.c.compute(ts)
    torch.cuda.synchronize()
    e = oracle_check(run, virial, batch_size, (virial, batch_size, period))
    print("OK forces virial=%s batch_size=%d period=%d (max err vs oracle %.2e)" % (virial, batch_size, period, e))

# ------------------------------------------------------------------ 2. MaxParticleNumberChange -> reallocate
run = FakeHoomdRun()
pot = htf.Potential.lj()
run.c.setPotential(pot.handle.value, False, False, 2)
run.c.compute(0)
before = run.force().clone()
run.pdata.setN(run.sysm.N, run.max_n + 500, 0)   # max N grows: HOOMD re-makes m_force / m_virial, the plugin its side buffers
assert run.c.forceElements() == run.max_n + 500
run.c.compute(1)
torch.cuda.synchronize()
assert torch.equal(run.force(), before)           # same positions, same list: same forces in the new arrays
print("OK reallocate")

# ------------------------------------------------------------------ 3. hoomd2tf: reference forces -> labels -> training step
for n_ref in (0, 2):
    run = FakeHoomdRun(mode="hoomd2tf", batch_size=400 if n_ref else 0)
    s = run.sysm
    theta = torch.tensor([0.7, 1.2], dtype=torch.float32, device=dev)
    theta_py = theta.clone()
    pot = htf.Potential.lj_param(0.7, 1.2, theta=theta)
    pot_py = htf.Potential.lj_param(0.7, 1.2, theta=theta_py)
    opt = optimizers.Adam(0.01)
    desc = opt.desc()
    state = torch.zeros(ops.optimizer_state_floats(2), dtype=torch.float32, device=dev)
    state_py = state.clone()
    run.c.setPotential(pot.handle.value, False, False, 0)   # training evaluates from the staged tensor
    run.c.setTraining(theta.data_ptr(), 2, state.data_ptr(), desc.kind, desc.lr, desc.beta1, desc.beta2, desc.epsilon, 0, 0.0)
    # labels: true LJ forces; with reference forces they arrive as two halves that sumReferenceForces adds (.cc:250-269)
    ctx_lab = htf.Context(r_cut=R_CUT, nneighs=NN, scalar_dtype=sdt, max_n=s.N, fused=2)
    ctx_lab.set_potential(htf.Potential.lj())
    ctx_py = htf.Context(r_cut=R_CUT, nneighs=NN, batch_size=400 if n_ref else 0, scalar_dtype=sdt, max_n=s.N, fused=0)
    ctx_py.set_potential(pot_py)
    refs = [H.FakeForce(run.sysdef) for _ in range(n_ref)]
    for r in refs:
        run.c.addReferenceForce(r)
    dummy = torch.zeros((s.N, 4), dtype=sdt, device=dev)
    hook = run.c.hook()
    for ts in range(6):
        labels = torch.zeros((s.N, 4), dtype=sdt, device=dev)
        nlc = run.nl
        ctx_lab.compute_forces(ts, ctx_lab.make_arrays(s.pos, s.N, nlc.n_neigh, nlc.head_list, nlc.nlist, s.box, labels))
        if n_ref:
            part = labels * 0.25
            refs[0].setForces(part.data_ptr(), s.N)
            rest = labels - part
            refs[1].setForces(rest.data_ptr(), s.N)
            labels = part.clone()
            ops.add_scalar4(labels, rest)
        else:
            run.net_force.copy_(labels)
        hook.update(ts)                                         # the half-step hook calls computeForces (.h:53-71)
        # the same step through the C ABI from Python: stage each batch's tensor, one gradient sweep, one optimizer step
        bs = 400 if n_ref else s.N
        arr = ctx_py.make_arrays(s.pos, s.N, nlc.n_neigh, nlc.head_list, nlc.nlist, s.box, dummy)
        for off in range(0, s.N, bs):
            n = min(bs, s.N - off)
            ctx_py.compute_forces(ts, arr, rows=(off, n))
            acc = ops.train_pair_grad(pot_py, ctx_py.nlist_buffer(n, dev), labels[off:off + n])
            ops.optimizer_step(theta_py, acc, 1.0 / (4.0 * n), state_py, desc)
            pot_py.refresh()
        torch.cuda.synchronize()
        assert torch.equal(theta, theta_py), (ts, theta, theta_py)
        assert torch.equal(state, state_py)
        s.force.copy_(labels)
        run.nve.step()
        run._nlist_compute(ts)
    assert float((theta - torch.tensor([0.7, 1.2], device=dev)).abs().max()) > 1e-3   # it moved (towards w0 w1^12 = 1)
    assert torch.count_nonzero(run.force()) == 0                                      # hoomd2tf never writes m_force (.cc:143-206)
    print("OK training n_ref=%d theta=%s" % (n_ref, theta.tolist()))

# ------------------------------------------------------------------ 4. errors surface as Python exceptions
run = FakeHoomdRun()
pot = htf.Potential.lj()
# (check_nlist counts the slots with dx > 0 per row, simmodel.py:214-219: NN = 2 fills both slots of some row with such
#  neighbors for certain)
small = M.TensorflowComputeAMD(run, run.sysdef, run.hnl, R_CUT, 2, M.FORCE_MODE.tf2hoomd, 1, 0)
small.setPotential(pot.handle.value, False, True, 2)
try:
    small.compute(0)
    raise SystemExit("an overflowing neighbor list must raise (simmodel.py:214-224)")
except (RuntimeError, ValueError) as e:
    assert "neighbor" in str(e).lower() or "nlist" in str(e).lower(), str(e)
try:
    run.c.getLogValue("nonsense", 0)
    raise SystemExit("unknown log quantity must raise (.cc:392-394)")
except RuntimeError:
    pass
print("OK errors")
print("ALL OK")
