"""A/B timing of the pair-MLP evaluator alone (no MD loop): synthetic liquid-like pair vectors
[131072, 128, 4] with ~95 live slots per row, every precision x activation.
    python tools/mlp_ab.py [fp32 split bf16] [--act tanh|linear] [--reps 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hoomd_tf_amd as htf  # noqa: E402
from hoomd_tf_amd.initializers import mlp_params  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("precisions", nargs="*", default=["fp32", "split", "bf16"])
    ap.add_argument("--act", default="tanh")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rows", type=int, default=131072)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1)
    N, NN = a.rows, 128
    d = torch.randn((N, NN, 3), generator=g, device=dev)
    d = d / d.norm(dim=2, keepdim=True)
    r = 0.8 + 2.2 * torch.rand((N, NN, 1), generator=g, device=dev) ** (1.0 / 3.0)
    x = torch.zeros((N, NN, 4), device=dev)
    x[:, :, :3] = d * r
    live = (torch.arange(NN, device=dev)[None, :] < (88 + torch.randint(0, 15, (N, 1), generator=g, device=dev)))
    x = x * live[:, :, None]
    params = mlp_params(seed=3, K=32, H1=64, H2=64, bias_scale=0.2)
    flops = 4.0 * (32 * 64 + 64 * 64 + 64) * N * NN
    ref = None
    for prec in a.precisions:
        pot = htf.Potential.pair_mlp(params, 0.0, 3.0, activation=a.act, precision=prec)
        out = torch.empty((N, 4), device=dev)
        for _ in range(3):
            htf.ops.eval_forces(pot, x, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            htf.ops.eval_forces(pot, x, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        if ref is None:
            ref = out.double().clone()
        err = float((out.double() - ref).abs().max() / ref.abs().max())
        print("%-6s %-6s %7.3f ms  %7.1f TFLOP/s (algorithmic)  max|diff vs first|/max = %.2e"
              % (prec, a.act, ms, flops / ms / 1e9, err))


if __name__ == "__main__":
    main()
