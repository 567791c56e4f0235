"""Is the C3-LJ step loop host-bound?  Enqueue time (loop wall before the final synchronize) against
total time, for the bench's own step function pieces."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hoomd_tf_amd as htf
from hoomd_tf_amd import standin

dev = torch.device("cuda", 0)
pos, L, a = standin.fcc_positions(32, 0.8442)
rng = np.random.default_rng(3)
pos = pos + 0.05 * a * rng.standard_normal(pos.shape)
pos -= np.round(pos / L) * L
s = standin.System(pos, L, dtype=torch.float32, device=dev)
s.randomize_velocities(kT=1.0, seed=3)
nl = standin.CellNlist(s, r_cut=3.0, r_buff=0.4, check_period=5)
nl.build()
ctx = htf.Context(r_cut=3.0, nneighs=128, scalar_dtype=torch.float32, max_n=s.N, fused=2)
ctx.set_potential(htf.Potential.lj())
nve = standin.NVE(s, 0.0005)
st = {"arr": ctx.make_arrays(s.pos, s.N, nl.n_neigh, nl.head_list, nl.nlist, s.box, s.force), "b": nl.n_builds, "ts": 0}

def step(check=True):
    ts = st["ts"]
    if check:
        nl.compute(ts)
        if nl.n_builds != st["b"]:
            st["arr"] = ctx.make_arrays(s.pos, s.N, nl.n_neigh, nl.head_list, nl.nlist, s.box, s.force)
            st["b"] = nl.n_builds
    ctx.compute_forces_overlapped(ts, st["arr"], None)
    nve.step()
    st["ts"] = ts + 1

for _ in range(50):
    step()
for label, check in (("with nlist checks/rebuilds", True), ("force + integrate only (no nlist check)", False)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(400):
        step(check)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-42s enqueue %.1f us/step, total %.1f us/step" % (label, (t1 - t0) / 400 * 1e6, (t2 - t0) / 400 * 1e6))
# host cost of the individual calls
torch.cuda.synchronize()
for name, fn in (("compute_forces_overlapped", lambda: ctx.compute_forces_overlapped(0, st["arr"], None)), ("nve.step", nve.step)):
    t0 = time.perf_counter()
    for _ in range(300):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("%-28s host %.1f us/call" % (name, (t1 - t0) / 300 * 1e6))
# latency of one distance check on an idle GPU (kernel + D2H + host wake-up)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    nl.needs_update()
t1 = time.perf_counter()
print("needs_update() on an idle GPU: %.1f us" % ((t1 - t0) / 200 * 1e6))
x = torch.zeros(1, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    float(x.item())
t1 = time.perf_counter()
print(".item() of a resident scalar:  %.1f us" % ((t1 - t0) / 200 * 1e6))
ph = torch.zeros(1).pin_memory()
ev = torch.cuda.Event()
t0 = time.perf_counter()
for _ in range(200):
    ph.copy_(x, non_blocking=True)
    ev.record()
    ev.synchronize()
    float(ph[0])
t1 = time.perf_counter()
print("pinned copy + event sync:      %.1f us" % ((t1 - t0) / 200 * 1e6))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    nl.build()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("nl.build(): host enqueue %.0f us, total %.0f us per rebuild" % ((t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6))
