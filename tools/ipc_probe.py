"""Does CUDA-IPC tensor sharing between two processes on ONE GPU work here (what a peer-to-peer halo without RCCL needs)?
Parent allocates a tensor, shares it with a child through torch.multiprocessing's reductions; the child writes through the mapping
while the parent's kernel-visible view changes.  Prints one line."""
import os
import sys
import time

import torch
import torch.multiprocessing as mp


def child(q_in, q_out):
    try:
        fn, args = q_in.get(timeout=60)
        t = fn(*args)                       # the parent's memory, mapped here
        t.add_(5.0)
        torch.cuda.synchronize()
        q_out.put(("ok", float(t.sum().item())))
        time.sleep(1.0)
    except Exception as e:  # noqa: BLE001
        q_out.put(("err", repr(e)))


def main():
    from torch.multiprocessing.reductions import reduce_tensor
    ctx = mp.get_context("spawn")
    q_in, q_out = ctx.Queue(), ctx.Queue()
    p = ctx.Process(target=child, args=(q_in, q_out))
    p.start()
    x = torch.arange(1024, dtype=torch.float32, device="cuda:0")
    before = float(x.sum().item())
    fn, args = reduce_tensor(x)
    q_in.put((fn, args))
    status, val = q_out.get(timeout=120)
    torch.cuda.synchronize()
    after = float(x.sum().item())
    p.join(timeout=30)
    print("ipc", status, "child sum", val, "parent before", before, "after", after, "=> shared" if after == before + 5 * 1024 else "=> NOT shared")


if __name__ == "__main__":
    main()
