"""Time the row GEMM of the backward passes (csrc/gemm_h2.h: k_gemm_rows) on the shapes of a C3 step, next to a plain device copy of
the same bytes (what this box's HBM gives a streaming kernel):  python tools/bench_rows.py [--rows 131072]
IRON_TRAIN_LIB selects a variant build of libiron_train.so."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=131072)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    from iron_amd import _lib
    lib = _lib.load_train()
    dev = torch.device("cuda", 0)
    R, K, N = a.rows, 256, 256
    X = torch.randn(R, K, device=dev)
    W = torch.randn(N, K, device=dev) / 16.0
    out = torch.empty(R, N, device=dev)
    nbytes = lib.iron_train_gemm_workspace_bytes(0, R, N)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)

    def timed(fn):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / a.iters * 1e3   # us

    def gemm():
        rc = lib.iron_train_gemm(0, 1, R, N, K, X.data_ptr(), K, W.data_ptr(), K, 0.0, out.data_ptr(), N, ws.data_ptr(), nbytes, _lib.stream_ptr(dev))
        assert rc == 0, rc
    t_g = timed(gemm)
    ref = X.double() @ W.double().t()
    err = float((out.double() - ref).norm() / ref.norm())
    t_c = timed(lambda: out.copy_(X))
    byt = 4.0 * R * (K + N)
    print("rows %d: Z = X W^T %.1f us = %.2f TB/s (A in + C out);  device copy of the same bytes %.1f us = %.2f TB/s;  rel-L2 %.1e"
          % (R, t_g, byt / t_g / 1e6, t_c, byt / t_c / 1e6, err))


if __name__ == "__main__":
    main()
