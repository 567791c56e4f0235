"""Do kernels of two PROCESSES run concurrently on one GPU here?  Process A keeps the GPU busy with a ~1.5 s spin kernel
(torch.cuda._sleep); process B, started while it runs, times a trivial kernel.  If B's kernel returns in microseconds the
two queues are served side by side; if it takes as long as A's spin, they are serialised (a device-side wait on another
process's kernel -- transport "peer" between ranks SHARING a GPU -- can then never be satisfied)."""
import time

import torch
import torch.multiprocessing as mp


def spinner(q):
    torch.cuda.init()
    x = torch.zeros(1, device="cuda:0")
    torch.cuda.synchronize()
    q.put("spinning")
    t0 = time.perf_counter()
    torch.cuda._sleep(int(1.5 * 2.0e9))     # ~1.5 s at ~2 GHz
    torch.cuda.synchronize()
    q.put(("spun", time.perf_counter() - t0))


def main():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=spinner, args=(q,))
    y = torch.zeros(1024, device="cuda:0")
    y.add_(1.0)
    torch.cuda.synchronize()
    p.start()
    assert q.get(timeout=120) == "spinning"
    time.sleep(0.2)
    t0 = time.perf_counter()
    y.add_(1.0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _, spun = q.get(timeout=60)
    p.join(timeout=30)
    print("other process's spin kernel: %.2f s; my kernel beside it: %.6f s => %s" % (spun, dt, "concurrent" if dt < 0.2 * spun else "SERIALISED"))


if __name__ == "__main__":
    main()
