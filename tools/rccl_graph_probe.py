"""Can this box's RCCL be captured into a hipGraph?  Each variant in its own child process (a crash must not end the probe).
    python tools/rccl_graph_probe.py            -> one line per variant
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes as C, os, sys, time
sys.path.insert(0, %(root)r)
import torch
import hoomd_tf_amd
from hoomd_tf_amd._lib import lib, check
variant, mode = sys.argv[1], sys.argv[2]
dev = torch.device("cuda:0")
ident = (C.c_char * 128)()
check(lib.htf_halo_unique_id(ident))
h = C.c_void_p()
check(lib.htf_halo_create(ident, 0, 1, C.byref(h)))
n = 4096
send = torch.arange(n * 4, dtype=torch.float32, device=dev).reshape(n, 4)
recv = torch.zeros_like(send)
nm = int(variant.split(":")[1]) if ":" in variant else 2
rows = n // nm
VP, SZ, IN = C.c_void_p * nm, C.c_size_t * nm, C.c_int * nm
sp = VP(*[send.data_ptr() + i * rows * 16 for i in range(nm)])
rp = VP(*[recv.data_ptr() + (nm - 1 - i) * rows * 16 for i in range(nm)])
sb = SZ(*[rows * 16] * nm)
pe = IN(*[0] * nm)
side2 = torch.cuda.Stream()
def exchange(async_):
    if variant.startswith("torchfork") or variant.startswith("kernelfork"):
        # fork / join with torch's own events and a second torch stream; the exchange itself synchronous on that stream
        ev, ev2 = torch.cuda.Event(), torch.cuda.Event()
        main = torch.cuda.current_stream()
        ev.record(main)
        side2.wait_event(ev)
        with torch.cuda.stream(side2):
            if variant.startswith("kernelfork"):
                recv.copy_(send.flip(0).reshape(nm, rows, 4).flip(1).reshape(n, 4))   # (no RCCL: the same data movement as kernels)
            else:
                s2 = C.c_void_p(side2.cuda_stream)
                check(lib.htf_halo_exchange_n(h, nm, sp, sb, pe, nm, rp, sb, pe, s2, 0))
            ev2.record(side2)
        main.wait_event(ev2)
        return
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(lib.htf_halo_exchange_n(h, nm, sp, sb, pe, nm, rp, sb, pe, s, async_))
    if async_:
        check(lib.htf_halo_exchange_end(h, s))
async_ = 1 if variant.startswith("async") else 0
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(3):
        exchange(async_)
    torch.cuda.synchronize()
    assert torch.equal(recv[:rows], send[(nm - 1) * rows:nm * rows])
    print("eager ok", flush=True)
    if variant.startswith("allreduce"):
        v = torch.ones(1, device=dev)
        check(lib.htf_halo_allreduce_max_f32(h, v.data_ptr(), 1, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    recv.zero_()
    torch.cuda.synchronize()
with torch.cuda.graph(g, capture_error_mode=mode):
    send.add_(1.0)
    if variant.startswith("allreduce"):
        check(lib.htf_halo_allreduce_max_f32(h, v.data_ptr(), 1, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    else:
        exchange(async_)
    recv.mul_(2.0)
print("captured", flush=True)
t = []
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    t.append((time.perf_counter() - t0) / 200 * 1e6)
if not variant.startswith("allreduce"):
    assert torch.equal(recv[:rows], 2.0 * send[(nm - 1) * rows:nm * rows]), "replay delivered something else"
print("replay ok %%.1f us per replay (3 kernels + exchange)" %% min(t), flush=True)
# eager timing of the same sequence
with torch.cuda.stream(side):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        send.add_(1.0)
        if not variant.startswith("allreduce"):
            exchange(async_)
        recv.mul_(2.0)
    torch.cuda.synchronize()
    print("eager %%.1f us per pass" %% ((time.perf_counter() - t0) / 200 * 1e6), flush=True)
'''


def main():
    variants = [("sync:2", "thread_local"), ("async:2", "thread_local"), ("sync:8", "thread_local"), ("async:8", "thread_local"),
                ("torchfork:2", "thread_local"), ("kernelfork:2", "thread_local"), ("allreduce", "thread_local")]
    envs = [{}]
    for env_extra in envs:
        for v, mode in variants:
            env = dict(os.environ, **env_extra)
            r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, v, mode], capture_output=True, text=True, timeout=300, env=env)
            out = " | ".join(l for l in r.stdout.strip().splitlines())
            err = r.stderr.strip().splitlines()
            print("%-10s %-12s %-32s rc=%4d  %s %s" % (v, mode, env_extra or "", r.returncode, out,
                                                     ("ERR: " + err[-1][:200]) if r.returncode and err else ""), flush=True)
        if env_extra == {}:
            pass


if __name__ == "__main__":
    main()
