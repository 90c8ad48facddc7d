"""Time the stage-1 NeuS forward render (row f-3) on one GPU: rays/s at the reference's validation batch (512 rays,
exp_runner-style) and at larger batches.  Seeded random-init networks of confs/womask_iron.conf, fixture camera.
    python tools/neus_frames.py [--rays 40000] [--batches 512 4096 16384]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=40000)
    ap.add_argument("--batches", type=int, nargs="+", default=[512, 4096, 16384])
    a = ap.parse_args()
    for line in run(a.rays, a.batches):
        print(json.dumps(line))


def build_networks(cuda: bool = True):
    """The seeded stage-1 networks of confs/womask_iron.conf (CPU or GPU resident)."""
    from iron_amd.fields import NeRF, RenderingNetwork, SDFNetwork, SingleVarianceNetwork
    torch.manual_seed(0)
    nets = {"sdf": SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5, scale=1.0, geometric_init=True,
                              weight_norm=True),
            "color": RenderingNetwork(d_feature=256, mode="idr", d_in=9, d_out=3, d_hidden=256, n_layers=8, skip_in=[4], weight_norm=True,
                                      multires=10, multires_view=4, squeeze_out=True),
            "nerf": NeRF(D=8, d_in=4, d_in_view=3, W=256, multires=10, multires_view=4, output_ch=4, skips=[4], use_viewdirs=True),
            "deviation": SingleVarianceNetwork(0.3)}
    return {k: v.cuda() for k, v in nets.items()} if cuda else nets


def rays(n: int):
    """n synthetic rays towards the unit sphere from (0, 0, -2.5) with near / far = mid -+ 1 (models/dataset.py:335-343), CPU tensors."""
    g = torch.Generator().manual_seed(1)
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g) * torch.tensor([0.25, 0.25, 0.0]) + torch.tensor([0.0, 0.0, 1.0]), dim=-1)
    o = torch.tensor([[0.0, 0.0, -2.5]]).expand(n, 3).contiguous()
    mid = -(o * d).sum(-1, keepdim=True)
    return o, d, mid - 1.0, mid + 1.0


def run(rays: int = 40000, batches=(4096,), repeats: int = 1, warm: bool = True, before_timed=None):
    """-> one dict per batch size.  `before_timed()` runs after the warm-up frame (same networks: packed, workspaces allocated) and before the clock starts."""
    class A:
        pass
    a = A()
    a.rays, a.batches = rays, list(batches)
    from iron_amd.renderer import NeuSRenderer
    nets = build_networks()
    r = NeuSRenderer(nets["nerf"], nets["sdf"], nets["deviation"], nets["color"], n_samples=64, n_importance=64, n_outside=32, up_sample_steps=4, perturb=0.0)
    o, d, near, far = (t.cuda() for t in globals()["rays"](a.rays))
    out = []
    for b in a.batches:
        @torch.no_grad()
        def frame():
            for s in range(0, a.rays, b):
                r.render(o[s:s + b], d[s:s + b], near[s:s + b], far[s:s + b], perturb_overwrite=0, cos_anneal_ratio=1.0)
        if warm:
            frame()
        torch.cuda.synchronize()
        if before_timed is not None:
            before_timed()
        t0 = time.perf_counter()
        for _ in range(repeats):
            frame()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / repeats
        out.append({"rays": a.rays, "batch": b, "ms": round(dt * 1e3, 2), "krays_per_s": round(a.rays / dt / 1e3, 1),
                    "mlp_points_per_ray": 64 + 48 + 128 + 128 + 160})
    return out


if __name__ == "__main__":
    main()
