"""Tracer work counters of one 800x800 S0 frame (and of a 128x128 one): python tools/trace_stats.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_grad_enabled(False)
from iron_amd import scenes
import iron_amd.raytracer as rt
nets = {k: v.cuda() for k, v in scenes.build_networks("S0").items()}
for res in (128, 800):
    K, W2C = scenes.fixture_camera_matrices(res, res)
    cam = rt.Camera(res, res, K.cuda(), W2C.cuda())
    tr = rt.RayTracer()
    rt.VERBOSE_MODE = True
    out = rt.raytrace_camera(cam, nets["sdf_network"], tr, max_num_rays=50000)
    rt.VERBOSE_MODE = False
    torch.cuda.synchronize()
    print(res, "hits", int(out["convergent_mask"].sum()), tr.last_stats)
