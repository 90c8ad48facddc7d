"""BASELINE config C3: one stage-2 training step at 512x512 on one GPU -- render_camera(handle_edges=True,
is_training=True) through the HIP operators, image loss + eikonal term as render_surface.py:533-653 forms them, backward
through libiron_train.so, Adam step on all networks.  Seeded synthetic scene (S1), random target image.
    python tools/train_step.py [--size 512] [--steps 5] [--no-edges]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--no-edges", action="store_true")
    a = ap.parse_args()
    print(json.dumps(run(a.size, a.steps, a.no_edges)))


def run(size: int = 512, steps: int = 5, no_edges: bool = False, warmup: int = 1) -> dict:
    class A:
        pass
    a = A()
    a.size, a.steps, a.no_edges = size, steps, no_edges
    from iron_amd import scenes
    from iron_amd.raytracer import Camera, RayTracer, render_camera
    from iron_amd.renderer_ggx import GGXColocatedRenderer
    from iron_amd.rendering_func import make_render_fn
    dev = torch.device("cuda", 0)
    nets = {k: v.to(dev) for k, v in scenes.build_networks("S1").items()}
    K, W2C = scenes.fixture_camera_matrices(a.size, a.size)
    cam = Camera(a.size, a.size, K.to(dev), W2C.to(dev))
    fn = make_render_fn(GGXColocatedRenderer(use_cuda=True))
    params = [p for n in nets.values() for p in n.parameters()]
    opt = torch.optim.Adam(params, lr=1e-5)
    target = torch.rand(a.size, a.size, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    tracer = RayTracer()
    times = []
    for step in range(a.steps + warmup):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = render_camera(cam, nets["sdf_network"], tracer, nets, fn, fill_holes=False, handle_edges=not a.no_edges, is_training=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        mask = res["convergent_mask"] | res["edge_mask"] if "edge_mask" in res else res["convergent_mask"]
        img_loss = (res["color"][mask] - target[mask]).abs().mean()
        eik_pts = torch.empty(a.size * a.size // 2, 3, device=dev).uniform_(-1.0, 1.0)
        eik = ((nets["sdf_network"].gradient(eik_pts).norm(dim=-1) - 1.0) ** 2).mean()
        loss = img_loss + 0.1 * eik
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        opt.step()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        if step >= warmup:  # warm-up: library load, rocBLAS kernels, allocator
            times.append((t1 - t0, t2 - t1, t3 - t2))
        hits, edges = int(res["convergent_mask"].sum()), int(res["edge_mask"].sum()) if "edge_mask" in res else 0
    f, b, o = (sum(x[i] for x in times) / len(times) * 1e3 for i in range(3))
    return {"config": "C3 stage-2 training step %dx%d S1%s" % (a.size, a.size, "" if not a.no_edges else " (no edge sampling)"),
            "hits": hits, "edge_pixels": edges, "eikonal_points": a.size * a.size // 2, "ms_forward_render": round(f, 2),
            "ms_loss_backward": round(b, 2), "ms_adam": round(o, 2), "ms_step": round(f + b + o, 2),
            "steps_per_s": round(1e3 / (f + b + o), 2), "loss": float(loss.detach())}


if __name__ == "__main__":
    main()
