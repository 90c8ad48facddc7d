"""Larger-than-headline frames (1600^2 = BASELINE config C4's per-view size, 2400^2) through render_camera with hole filling and
edge sampling: sizes, memory high-water mark, finite outputs."""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iron_amd import scenes
from iron_amd.raytracer import Camera, RayTracer, render_camera
from iron_amd.renderer_ggx import GGXColocatedRenderer
from iron_amd.rendering_func import make_render_fn
nets = {k: v.cuda() for k, v in scenes.build_networks("S0").items()}
fn = make_render_fn(GGXColocatedRenderer(use_cuda=True))
for res in (1600, 2400):
    K, W2C = scenes.fixture_camera_matrices(res, res)
    cam = Camera(res, res, K.cuda(), W2C.cuda())
    render_camera(cam, nets["sdf_network"], RayTracer(), nets, fn, fill_holes=False, handle_edges=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = render_camera(cam, nets["sdf_network"], RayTracer(), nets, fn, fill_holes=True, handle_edges=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(res, "hits", int(out["convergent_mask"].sum()), "edge px", int(out["edge_mask"].sum()), "%.1f ms  %.2f Mrays/s (with edges)" % (dt * 1e3, res * res / dt / 1e6),
          "finite", bool(torch.isfinite(out["color"]).all()), "mem %.1f GiB" % (torch.cuda.max_memory_allocated() / 2**30))
