"""Where the silhouette path's time goes (row f-1): stage-by-stage device+host time of render_camera(fill_holes=True,
handle_edges=True) at 800x800 S0, each stage bracketed by a synchronize (so the sum exceeds the pipelined frame time).
    python tools/edge_profile.py [--res 800] [--scene S0]"""
import argparse
import os
import sys
import time

import torch
torch.set_grad_enabled(False)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iron_amd import scenes  # noqa: E402
import iron_amd.raytracer as rt  # noqa: E402
from iron_amd.renderer_ggx import GGXColocatedRenderer  # noqa: E402
from iron_amd.rendering_func import make_render_fn  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=800)
ap.add_argument("--scene", default="S0")
a = ap.parse_args()
dev = torch.device("cuda", 0)
nets = {k: v.to(dev) for k, v in scenes.build_networks(a.scene).items()}
K, W2C = scenes.fixture_camera_matrices(a.res, a.res)
cam = rt.Camera(a.res, a.res, K.to(dev), W2C.to(dev))
fn = make_render_fn(GGXColocatedRenderer(use_cuda=True))
tracer = rt.RayTracer()
sdf = nets["sdf_network"]


def timed(label, f, acc):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = f()
    torch.cuda.synchronize()
    acc[label] = acc.get(label, 0.0) + (time.perf_counter() - t0) * 1e3
    return r


def frame(acc):
    res = timed("1 trace full image", lambda: rt.raytrace_camera(cam, sdf, tracer, max_num_rays=50000), acc)
    res2 = timed("2 trace + fill_holes + detect_edges (incl. 1)", lambda: rt.raytrace_camera(cam, sdf, tracer, max_num_rays=50000, fill_holes=True, detect_edges=True), acc)
    timed("3 shade full image", lambda: rt.render_normal_and_color(res2, sdf, nets, fn), acc)
    timed("4 render_edge_pixels", lambda: rt.render_edge_pixels(res2, cam, sdf, tracer, nets, fn), acc)
    return res2


for _ in range(2):
    frame({})
acc = {}
n = 5
for _ in range(n):
    r = frame(acc)
for k, v in acc.items():
    print("%-50s %8.3f ms" % (k, v / n))
print("edge pixels:", int(r["edge_mask"].sum()))
for flags in ((False, False), (True, False), (False, True), (True, True)):
    f = lambda: rt.render_camera(cam, sdf, tracer, nets, fn, fill_holes=flags[0], handle_edges=flags[1])
    f(); f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    print("render_camera fill_holes=%s handle_edges=%s: %.3f ms" % (flags[0], flags[1], (time.perf_counter() - t0) / n * 1e3))
