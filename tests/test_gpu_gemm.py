"""GPU: the hand-written split-fp16 MFMA GEMMs of libiron_train.so (csrc/gemm_h2.h) on their own, against an fp64 product of the
same fp32 inputs: every shape class the backward passes use (forward recompute, dX, dW with split-K), aligned and ragged
leading dimensions, tiny outputs, and gradient operands whose magnitude is far outside fp16's range (the power-of-two scale)."""
import ctypes as C

import numpy as np
import pytest
import torch

from iron_amd import _lib

pytestmark = pytest.mark.gpu


def gemm(op_a, op_b, A, B, m, n, k, beta=0.0, C0=None):
    lib = _lib.load_train()
    out = torch.zeros((m, n), dtype=torch.float32, device="cuda") if C0 is None else C0.clone()
    nbytes = lib.iron_train_gemm_workspace_bytes(op_a, m, n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    rc = lib.iron_train_gemm(op_a, op_b, m, n, k, A.data_ptr(), A.shape[1], B.data_ptr(), B.shape[1], float(beta), out.data_ptr(), n,
                             ws.data_ptr(), nbytes, _lib.stream_ptr(A.device))
    assert rc == 0, rc
    return out


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


@pytest.mark.parametrize("m,n,k", [(4517, 256, 256), (70001, 256, 39), (3000, 217, 256), (2999, 257, 256), (513, 3, 256), (1000, 298, 256),
                                    (1024, 256, 349), (64, 256, 256)])
def test_forward_recompute_and_dx_shapes(m, n, k):
    g = torch.Generator().manual_seed(m + n + k)
    X = torch.randn(m, k, generator=g).cuda()
    W = (torch.randn(n, k, generator=g) / np.sqrt(k)).cuda()
    ref = X.double() @ W.double().t()
    out = gemm(0, 1, X, W, m, n, k)                     # Z = X W^T
    e = rel(out, ref)
    # dX = dZ W with dZ tiny (a gradient): the operand scale must keep it out of fp16's subnormals
    dZ = (torch.randn(m, n, generator=g) * 3e-7).cuda()
    ref2 = dZ.double() @ W.double()
    out2 = gemm(0, 0, dZ, W, m, k, n)
    e2 = rel(out2, ref2)
    print("m %6d n %3d k %3d: Z = X W^T rel-L2 %.2e;  dX = dZ W (|dZ| ~ 3e-7) rel-L2 %.2e" % (m, n, k, e, e2))
    assert e <= 5e-7 and e2 <= 5e-7
    assert float((out - ref).abs().max() / ref.abs().max()) <= 3e-6


@pytest.mark.parametrize("out_dim,in_dim,rows", [(256, 256, 131072), (257, 256, 20000), (217, 256, 9000), (256, 39, 70001), (3, 256, 5000),
                                                 (1, 256, 4517), (256, 349, 4517), (256, 256, 100)])
def test_weight_gradient_shapes(out_dim, in_dim, rows):
    g = torch.Generator().manual_seed(out_dim * 7 + in_dim + rows)
    dZ = (torch.randn(rows, out_dim, generator=g) * (1.0 + 50.0 * torch.rand(rows, 1, generator=g)) * 1e3).cuda()   # rows of very different size
    X = torch.randn(rows, in_dim, generator=g).cuda()
    ref = dZ.double().t() @ X.double()
    out = gemm(1, 0, dZ, X, out_dim, in_dim, rows)
    e = rel(out, ref)
    prev = torch.randn(out_dim, in_dim, generator=g).cuda()
    out_b = gemm(1, 0, dZ, X, out_dim, in_dim, rows, beta=1.0, C0=prev)   # accumulation over chunks of points
    print("dW [%3d x %3d] over %6d rows: rel-L2 %.2e; with beta = 1 %.2e" % (out_dim, in_dim, rows, e, rel(out_b, ref + prev.double())))
    assert e <= 5e-7 and rel(out_b, ref + prev.double()) <= 5e-7
    assert torch.equal(out, gemm(1, 0, dZ, X, out_dim, in_dim, rows))      # fixed summation order: run-to-run identical


def test_operand_outside_the_split_range_raises_the_flag():
    """activations / weights go into the fp16 split unscaled; an element beyond 65 504 must not pass silently
    (include/iron_train.h: iron_train_numeric_status)."""
    from iron_amd import autograd
    g = torch.Generator().manual_seed(11)
    X = torch.randn(3000, 256, generator=g).cuda()
    W = (torch.randn(256, 256, generator=g) / 16.0).cuda()
    autograd.numeric_status(reset=True)
    gemm(0, 1, X, W, 3000, 256, 256)
    assert autograd.numeric_status(reset=True) is False
    Xb = X.clone()
    Xb[1234, 77] = 1.0e5                                   # a forward operand (row kernel's loader)
    out = gemm(0, 1, Xb, W, 3000, 256, 256)
    assert not torch.isfinite(out[1234]).all()            # loud, not plausible
    assert autograd.numeric_status(reset=False) is True   # sticky until reset
    assert autograd.numeric_status(reset=True) is True
    assert autograd.numeric_status(reset=True) is False
    Wb = W.clone()
    Wb[5, 9] = float("inf")                                # the packed operand (k_gemm_pack_b)
    gemm(0, 1, X, Wb, 3000, 256, 256)
    assert autograd.numeric_status(reset=True) is True
    # a gradient operand of 1e5 is inside the range: it carries a power-of-two scale from its absolute maximum
    dZ = (torch.randn(3000, 256, generator=g) * 1.0e5).cuda()
    gemm(1, 0, dZ, X, 256, 256, 3000)
    assert autograd.numeric_status(reset=True) is False
    gemm(1, 0, dZ, Xb, 256, 256, 3000)                     # the un-scaled operand of dW (split-K kernel's loader)
    assert autograd.numeric_status(reset=True) is True
