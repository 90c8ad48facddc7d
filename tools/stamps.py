"""Timeline of the ring steps of one SDF evaluation from an IRON_H2_STAMP diagnostic build (GPU box).
usage: IRON_HIP_LIB=<stamp build .so> python tools/stamps.py
Stamp slots per step: 0 after barrier, 1 refill issued, 2 after k-step 0, 3 after k-step 7, 4 after k-step 15,
5 arrival at the next boundary (before vmcnt wait), 6 after the vmcnt wait (before the barrier)."""
import ctypes as C
import os
import sys

import numpy as np
import torch
torch.set_grad_enabled(False)  # measurement / inspection of the inference kernels: nothing is attached

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iron_amd import _lib, scenes  # noqa: E402

net = scenes.build_networks("S1")["sdf_network"].cuda()
x = torch.rand(1 << 20, 3, device="cuda") * 2 - 1
for _ in range(3):
    net.sdf(x)
torch.cuda.synchronize()
lib = _lib.load()
buf = (C.c_ulonglong * (4 * 72 * 8))()
fn = lib.iron_debug_h2_stamps
fn.restype = C.c_int
assert fn(buf) == 0
a = np.array(buf[:], dtype=np.uint64).reshape(4, 72, 8).astype(np.int64)
t0 = a[:, 0, 0].min()
print("step kind |  per wave: barrier-wait  refill-issue  ks0  ks1-7  ks8-15 | step total (wave 0)")
tot = np.zeros(7)
for q in range(72):
    row = []
    for w in range(4):
        r = a[w, q]
        nxt = a[w, q + 1] if q + 1 < 72 else None
        bw = r[0] - r[6] if r[6] else 0          # barrier wait of THIS step (stamps 5,6 precede stamp 0 of the same step)
        vw = r[6] - r[5] if r[5] else 0          # vmcnt wait
        row.append((vw, bw, r[1] - r[0], r[2] - r[7], r[3] - r[2], r[4] - r[3], r[7] - r[1]))
    row = np.array(row)
    hidden = a[0, q, 4] != 0
    if hidden:
        tot += row.mean(axis=0)
    span = (a[0, q + 1, 0] - a[0, q, 0]) if q + 1 < 72 else 0
    print("%2d %s | vmw %s  bar %s  dma %s  ks0 %s  ks1-7 %s  ks8-15 %s  stamp-cost %s | %d" % (
        q, "H" if hidden else "h", *[str(row[:, i].tolist()) for i in range(7)], span))
print("hidden-step means (vmcnt wait, barrier wait, refill issue, ks0, ks1-7, ks8-15):", (tot / 64).round(0).tolist())
print("evaluation span (wave 0): %d cycles" % (a[0, 71, 4] - a[0, 0, 0]))
