"""One stage-1 (NeuS) training step as render_volume.py:160-200 forms it, on one GPU: NeuSRenderer.render under autograd on a
batch of rays (perturb = 1), L1 colour + eikonal + BCE mask loss, backward through libiron_train.so, Adam on the four networks.
Seeded random-init networks of confs/womask_iron.conf, synthetic rays / targets.
    python tools/neus_train_step.py [--batch 512] [--steps 10]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(batch: int = 512, steps: int = 10, warmup: int = 2) -> dict:
    from iron_amd.fields import NeRF, RenderingNetwork, SDFNetwork, SingleVarianceNetwork
    from iron_amd.renderer import NeuSRenderer
    torch.manual_seed(0)
    nets = [SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5, scale=1.0, geometric_init=True,
                       weight_norm=True).cuda(),
            RenderingNetwork(d_feature=256, mode="idr", d_in=9, d_out=3, d_hidden=256, n_layers=8, skip_in=[4], weight_norm=True, multires=10,
                             multires_view=4, squeeze_out=True).cuda(),
            NeRF(D=8, d_in=4, d_in_view=3, W=256, multires=10, multires_view=4, output_ch=4, skips=[4], use_viewdirs=True).cuda(),
            SingleVarianceNetwork(0.3).cuda()]
    sdf, col, nerf, dev = nets
    r = NeuSRenderer(nerf, sdf, dev, col, n_samples=64, n_importance=64, n_outside=32, up_sample_steps=4, perturb=1.0)
    opt = torch.optim.Adam([p for n in nets for p in n.parameters()], lr=5e-4)
    g = torch.Generator().manual_seed(1)
    d = torch.nn.functional.normalize(torch.randn(batch, 3, generator=g) * torch.tensor([0.25, 0.25, 0.0]) + torch.tensor([0.0, 0.0, 1.0]), dim=-1)
    o = torch.tensor([[0.0, 0.0, -2.5]]).expand(batch, 3).contiguous()
    mid = -(o * d).sum(-1, keepdim=True)
    o, d, near, far = o.cuda(), d.cuda(), (mid - 1.0).cuda(), (mid + 1.0).cuda()
    true_rgb = torch.rand(batch, 3, generator=g).cuda()
    mask = (torch.rand(batch, 1, generator=g) > 0.3).float().cuda()
    times = []
    for step in range(steps + warmup):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = r.render(o, d, near, far, background_rgb=None, cos_anneal_ratio=min(1.0, step / 50.0))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        mask_sum = mask.sum() + 1e-5
        color_loss = ((out["color_fine"] - true_rgb) * mask).abs().sum() / mask_sum
        mask_loss = torch.nn.functional.binary_cross_entropy(out["weight_sum"].clip(1e-3, 1.0 - 1e-3), mask)
        loss = color_loss + 0.1 * out["gradient_error"] + 0.1 * mask_loss
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        opt.step()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        if step >= warmup:
            times.append((t1 - t0, t2 - t1, t3 - t2))
    f, b, a = (sum(x[i] for x in times) / len(times) * 1e3 for i in range(3))
    return {"config": "stage-1 NeuS training step, %d rays x (64 + 4x16 + 32 outside) samples, perturb = 1" % batch, "ms_render": round(f, 2),
            "ms_loss_backward": round(b, 2), "ms_adam": round(a, 2), "ms_step": round(f + b + a, 2), "steps_per_s": round(1e3 / (f + b + a), 2),
            "krays_per_s": round(batch / (f + b + a), 2), "loss": float(loss.detach())}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    print(json.dumps(run(a.batch, a.steps)))
