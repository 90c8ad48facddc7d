"""Micro-benchmark of the pure batched SDF-MLP kernel (the "8x256 SDF MLP" roofline sub-target):
N random points, 918 016 algorithmic FLOP per point, fp32 MFMA peak 157.3 TFLOP/s."""
import argparse
import os
import sys
import time

import torch
torch.set_grad_enabled(False)  # measurement / inspection of the inference kernels: nothing is attached

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iron_amd import scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 20)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
net = scenes.build_networks("S1")["sdf_network"].cuda()
x = (torch.rand(a.n, 3, device="cuda") * 2 - 1)
net.sdf(x)
torch.cuda.synchronize()
for _ in range(2):
    net.sdf(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    net.sdf(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
fl = 918016.0 * a.n
print("n=%d  %.3f ms  %.1f Mevals/s  %.2f TFLOP/s algorithmic  (%.1f %% of 157.3 fp32-MFMA peak)" %
      (a.n, dt * 1e3, a.n / dt / 1e6, fl / dt / 1e12, 100 * fl / dt / 157.3e12))
