"""Times the HIP tracer on a full synthetic view and prints Mrays/s + evals."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iron_amd import scenes
from iron_amd.raytracer import Camera, RayTracer, raytrace_camera
import iron_amd.raytracer as rt
ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=800)
ap.add_argument("--scene", default="S0")
ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
nets = scenes.build_networks(a.scene)
sdf = nets["sdf_network"].cuda()
K, W2C = scenes.fixture_camera_matrices(a.res, a.res)
cam = Camera(a.res, a.res, K.cuda(), W2C.cuda())
tr = RayTracer()
rt.VERBOSE_MODE = True
res = raytrace_camera(cam, sdf, tr, max_num_rays=50000)
torch.cuda.synchronize()
print("stats", tr.last_stats, "conv frac", res["convergent_mask"].float().mean().item())
rt.VERBOSE_MODE = False
t0 = time.perf_counter()
for _ in range(a.iters):
    res = raytrace_camera(cam, sdf, tr, max_num_rays=50000)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
n = a.res * a.res
ev = tr.last_stats["n_evals"]
print("%s %dx%d trace: %.2f ms  %.3f Mrays/s  hip evals/ray %.1f  -> %.1f TFLOP/s on executed evals" %
      (a.scene, a.res, a.res, dt * 1e3, n / dt / 1e6, ev / n, ev * 918016 / dt / 1e12))
