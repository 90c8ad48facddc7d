#!/usr/bin/env python3
"""iron_sdf_get_all: reverse-mode kernel (tape workspace) against the forward-mode kernel, same points.
    python tools/bench_getall.py [n ...]     (default: 297248 = C1's hits, 524288 = C2's points)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iron_amd import _lib, scenes  # noqa: E402

FLOP_SDF_GRAD = 2 * (524544 + 459008)


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [297248, 524288]
    net = scenes.build_networks("S0")["sdf_network"].cuda()
    lib = _lib.load()
    h = net.hip_net()
    for n in sizes:
        x = (torch.rand(n, 3, device="cuda") * 1.2 - 0.6)
        sdf = torch.empty(n, device="cuda"); feat = torch.empty(n, 256, device="cuda"); grad = torch.empty(n, 3, device="cuda")
        for rev in (True, False, True, False):
            nbytes = lib.iron_sdf_get_all_workspace_bytes(h.handle, n) if rev else 0
            ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device="cuda")

            def call():
                _lib.check(lib.iron_sdf_get_all(h.handle, x.data_ptr(), n, sdf.data_ptr(), feat.data_ptr(), grad.data_ptr(),
                                                ws.data_ptr() if rev else None, nbytes, _lib.stream_ptr(x.device)))
            for _ in range(2):
                call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                call()
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print("n %7d  %-12s %.3f ms  %.2f ns/point  %.0f TFLOP/s algorithmic (tape %d MiB)" %
                  (n, "reverse-mode" if rev else "forward-mode", ms, ms * 1e6 / n, FLOP_SDF_GRAD * n / ms / 1e9, nbytes >> 20), flush=True)


if __name__ == "__main__":
    main()
