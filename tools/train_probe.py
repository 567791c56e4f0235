"""Times the pair-MLP training sweep (htf_train_pair_grad + optimizer + image refresh) at the
C3/C5 per-GPU size.  usage: python tools/train_probe.py [N] [NN]"""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, ".")
import hoomd_tf_amd as htf

N = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
NN = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
x = (torch.rand((N, NN, 4), device=dev, generator=g) - 0.5) * 3.4
live = torch.rand((N, NN), device=dev, generator=g) < 0.75
live, _ = torch.sort(live.to(torch.int8), dim=1, descending=True)
x = x * live.unsqueeze(-1)
labels = torch.randn((N, 4), device=dev, generator=g)
layer = htf.PairMLP(32, 64, 64, 0.0, 3.0, activation="tanh", seed=3)
pot = layer.potential()
opt = htf.optimizers.Adam(1e-3)
desc = opt.desc(0, (0.0,))
state = torch.zeros(htf.ops.optimizer_state_floats(layer.w.numel()), device=dev)


def step():
    accum = htf.ops.train_pair_grad(pot, x, labels)
    htf.ops.optimizer_step(layer.w, accum, 1.0 / (4 * N), state, desc)
    layer.after_update()


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 5
for _ in range(K):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
pairs = float(live.sum())
print("train step: %.2f ms  (%d live pairs, %.1f G pair-updates/s, loss %.4g)" % (dt * 1e3, pairs, pairs / dt / 1e9, float(state[20])))
t0 = time.perf_counter()
for _ in range(K):
    htf.ops.eval_forces(pot, x)
torch.cuda.synchronize()
print("inference: %.2f ms" % ((time.perf_counter() - t0) / K * 1e3))
