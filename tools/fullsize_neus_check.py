"""One-off larger-batch check of the stage-1 path against the oracle (too slow for the suite): NeuSRenderer.render under autograd on
RAYS rays, loss = colour + eikonal statistic + weight_sum, all parameter gradients vs torch.autograd over oracle/neus_ref.render_train
on the GPU box's host cores.    python tools/fullsize_neus_check.py [RAYS=1024]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import cpu_sd  # noqa: E402
from iron_amd.fields import NeRF, RenderingNetwork, SDFNetwork, SingleVarianceNetwork  # noqa: E402
from iron_amd.renderer import NeuSRenderer  # noqa: E402
from oracle import iron_ref as R  # noqa: E402
from oracle import neus_ref as N  # noqa: E402
from oracle import train_ref as T  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.set_num_threads(min(16, os.cpu_count() or 8))
torch.manual_seed(0)
nets = {"sdf_network": SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5, scale=1.0, geometric_init=True,
                                  weight_norm=True),
        "color_network": RenderingNetwork(d_feature=256, mode="idr", d_in=9, d_out=3, d_hidden=256, n_layers=8, skip_in=[4], weight_norm=True,
                                          multires=10, multires_view=4, squeeze_out=True),
        "nerf": NeRF(D=8, d_in=4, d_in_view=3, W=256, multires=10, multires_view=4, output_ch=4, skips=[4], use_viewdirs=True),
        "deviation_network": SingleVarianceNetwork(0.3)}
g = torch.Generator().manual_seed(1)
d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g) * torch.tensor([0.3, 0.3, 0.0]) + torch.tensor([0.0, 0.0, 1.0]), dim=-1)
o = torch.tensor([[0.03, -0.02, -2.5]]).expand(n, 3).contiguous()
mid = -(o * d).sum(-1, keepdim=True)
near, far = mid - 1.0, mid + 1.0
wc, ww = torch.rand(n, 3, generator=g) - 0.3, torch.rand(n, 1, generator=g) - 0.5

sd = {k: T.leaf_state(cpu_sd(nets[k])) for k in ("sdf_network", "color_network", "nerf")}
var = nets["deviation_network"].variance.detach().clone().requires_grad_(True)
sc = N.NeusScene(sd["sdf_network"], R.SDFSpec(), sd["color_network"], sd["nerf"], var)
t0 = time.time()
print("oracle: %d rays on %d threads ..." % (n, torch.get_num_threads()), flush=True)
ref = N.render_train(sc, o, d, near, far, background_rgb=None, cos_anneal_ratio=0.6)
print("oracle: forward done after %.1f s" % (time.time() - t0), flush=True)
((ref["color_fine"] * wc).sum() + 0.1 * ref["gradient_error"] + (ref["weight_sum"] * ww).sum()).backward()
t_cpu = time.time() - t0
print("oracle: done after %.1f s" % t_cpu, flush=True)

gn = {k: v.cuda() for k, v in nets.items()}
r = NeuSRenderer(gn["nerf"], gn["sdf_network"], gn["deviation_network"], gn["color_network"], n_samples=64, n_importance=64, n_outside=32,
                 up_sample_steps=4, perturb=0.0)
out = r.render(o.cuda(), d.cuda(), near.cuda(), far.cuda(), perturb_overwrite=0, cos_anneal_ratio=0.6)
((out["color_fine"] * wc.cuda()).sum() + 0.1 * out["gradient_error"] + (out["weight_sum"] * ww.cuda()).sum()).backward()
torch.cuda.synchronize()
dc = float((out["color_fine"].detach().cpu() - ref["color_fine"].detach()).abs().max())
worst = [(float(abs(gn["deviation_network"].variance.grad.item() - var.grad.item()) / abs(var.grad.item())), "deviation_network/variance")]
for name in sd:
    for pname, p in gn[name].named_parameters():
        rg = sd[name][pname].grad
        if rg is None or float(rg.abs().max()) < 1e-9:
            continue
        a, b = p.grad.double().cpu().numpy().ravel(), rg.double().numpy().ravel()
        worst.append((float(np.linalg.norm(a - b) / np.linalg.norm(b)), "%s/%s" % (name, pname)))
worst.sort(reverse=True)
print("rays %d: colour max|d| %.2e, oracle %.1f s; worst gradient rel-L2: %s; median %.2e" % (
    n, dc, t_cpu, worst[:3], float(np.median([w[0] for w in worst]))))
