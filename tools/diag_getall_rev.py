"""Diagnostic for the reverse-mode get_all kernel: with a -DIRON_REV_DEBUG build (IRON_HIP_LIB) the kernel writes d_l = d sdf / d z_l
of layer IRON_REV_DEBUG_LAYER into the feature output; compared here with torch autograd on the same folded weights."""
import math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iron_amd import _lib, scenes
netc = scenes.build_networks("S1")["sdf_network"]
sd = {k: v.detach().double() for k, v in netc.state_dict().items()}
net = netc.cuda()
lib = _lib.load(); h = net.hip_net()
g = torch.Generator().manual_seed(11)
n = 256
x = (torch.rand(n, 3, generator=g) * 1.6 - 0.8)


def W(l):
    v, gg = sd["lin%d.weight_v" % l], sd["lin%d.weight_g" % l]
    return gg * v / v.norm(dim=1, keepdim=True)


def forward(x64):
    freqs = [2.0 ** k for k in range(6)]
    pe = torch.cat([x64] + [f(x64 * fr) for fr in freqs for f in (torch.sin, torch.cos)], dim=1)
    hcur, zs = pe, []
    for l in range(9):
        if l == 4:
            hcur = torch.cat([hcur, pe], dim=1) / math.sqrt(2)
        z = hcur @ W(l).T + sd["lin%d.bias" % l]
        if l < 8:
            z.retain_grad(); zs.append(z)
            hcur = torch.nn.functional.softplus(z, beta=100)
        else:
            return z, zs


x64 = x.double().requires_grad_(True)
y, zs = forward(x64)
y[:, 0].sum().backward()
print("autograd grad[0]", x64.grad[0].tolist())


def run(layer):
    os.environ["IRON_REV_DEBUG_LAYER"] = str(layer)
    xc = x.cuda()
    sdf = torch.empty(n, device="cuda"); feat = torch.zeros(n, 256, device="cuda"); grad = torch.empty(n, 3, device="cuda")
    nbytes = lib.iron_sdf_get_all_workspace_bytes(h.handle, n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    _lib.check(lib.iron_sdf_get_all(h.handle, xc.data_ptr(), n, sdf.data_ptr(), feat.data_ptr(), grad.data_ptr(), ws.data_ptr(), nbytes, _lib.stream_ptr(xc.device)))
    torch.cuda.synchronize()
    return sdf.cpu(), feat.cpu(), grad.cpu()


for layer in (7, 6, 5, 4, 3, 2, 1, 0):
    _, d, gr = run(layer)
    want = zs[layer].grad
    w = want.shape[1]
    err = (d[:, :w].double() - want).abs()
    print("layer %d: |d_l| max %.3e  err max %.3e  rel-L2 %.3e   bad units (col err > 1e-3*max): %s" %
          (layer, float(want.abs().max()), float(err.max()), float(err.norm() / want.norm()),
           (err.amax(dim=0) > 1e-3 * want.abs().max()).nonzero().flatten()[:24].tolist()))
print("kernel grad[0]", gr[0].tolist())
print("grad err max %.3e" % float((gr.double() - x64.grad).abs().max()))
_, d, _ = run(7)
want = zs[7].grad
wl = W(8)[0]
sig = torch.sigmoid(100 * zs[7].detach())
print("unit   kernel d7      want d7       w_last        sigma'      kernel/w_last")
for u in list(range(12)) + [32, 33, 224, 225, 226, 255]:
    print("%4d  % .6e  % .6e  % .6e  %.6f  %.6f" % (u, float(d[0, u]), float(want[0, u]), float(wl[u]), float(sig[0, u]), float(d[0, u]) / float(wl[u])))
