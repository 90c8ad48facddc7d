"""ShardedRenderer.render() with world = 1 against render_camera on the same view: what the sharded step's own glue (record packing,
assemble / un-tile, result split) costs per frame, and where the host sits."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_grad_enabled(False)
from iron_amd import scenes
from iron_amd import raytracer as rt
from iron_amd import rendering_func as rf
from iron_amd.renderer_ggx import GGXColocatedRenderer
from iron_amd.sharding import ShardedRenderer

dev = torch.device("cuda", 0)
nets = {k: v.to(dev) for k, v in scenes.build_networks("S0").items()}
K, W2C = scenes.fixture_camera_matrices(800, 800)
cam = rt.Camera(800, 800, K.to(dev), W2C.to(dev))
fn = rf.make_render_fn(GGXColocatedRenderer(use_cuda=True))
tr = rt.RayTracer()
sh = ShardedRenderer(nets["sdf_network"], nets, rt.RayTracer(), fn, tile=32, chunk=50000, world=1, rank=0)


def loop(f, n=8):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("render_camera        %.2f ms per frame" % loop(lambda: rt.render_camera(cam, nets["sdf_network"], tr, nets, fn, handle_edges=False)))
print("ShardedRenderer(w=1) %.2f ms per frame" % loop(lambda: sh.render([cam])))
marks = []
for name in ("trace_begin", "trace_finish", "shade", "assemble"):
    f = getattr(sh, name)
    def g(*a, _f=f, _n=name, **k):
        t0 = time.perf_counter(); r = _f(*a, **k); marks.append((_n, t0, time.perf_counter())); return r
    setattr(sh, name, g)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    sh.render([cam])
torch.cuda.synchronize()
for n, a, b in marks:
    print("%-14s start %8.2f dur %7.2f" % (n, (a - t0) * 1e3, (b - a) * 1e3))
