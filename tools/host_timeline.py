"""Host-side timeline of C1 frames (no profiler): when does the host enter / leave each phase of render_camera, and how long does it
sit in the one host sync of the frame (the light scalar)?  Prints per-frame host durations in ms."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_grad_enabled(False)
from iron_amd import scenes
from iron_amd import raytracer as rt
from iron_amd import rendering_func as rf
from iron_amd.renderer_ggx import GGXColocatedRenderer

dev = torch.device("cuda", 0)
nets = {k: v.to(dev) for k, v in scenes.build_networks("S0").items()}
K, W2C = scenes.fixture_camera_matrices(800, 800)
cam = rt.Camera(800, 800, K.to(dev), W2C.to(dev))
fn = rf.make_render_fn(GGXColocatedRenderer(use_cuda=True))
tr = rt.RayTracer()
marks = []


def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        marks.append((name, t0, time.perf_counter()))
        return r
    setattr(mod, name, g)


wrap(rt, "raytrace_camera")
wrap(rt, "render_normal_and_color")
wrap(rt, "raytrace_pixels")
orig_float = None
for _ in range(3):
    rt.render_camera(cam, nets["sdf_network"], tr, nets, fn, handle_edges=False)
torch.cuda.synchronize()
marks.clear()
t_start = time.perf_counter()
N = 6
for _ in range(N):
    marks.append(("frame", time.perf_counter(), 0))
    rt.render_camera(cam, nets["sdf_network"], tr, nets, fn, handle_edges=False)
torch.cuda.synchronize()
t_end = time.perf_counter()
print("wall per frame %.2f ms" % ((t_end - t_start) / N * 1e3))
for name, a, b in marks:
    print("%-26s start %8.2f  dur %7.2f" % (name, (a - t_start) * 1e3, (b - a) * 1e3 if b else 0))
