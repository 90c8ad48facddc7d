"""G15 setting: HIP silhouette walk vs the CPU oracle's, edge pixel by edge pixel (position difference, n.v at the end),
to tell a discrete walk-termination flip from a defect.   python tools/edge_walk_diag.py"""
import os
import sys

import numpy as np
import torch
torch.set_grad_enabled(False)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from iron_amd import scenes  # noqa: E402
from iron_amd.raytracer import Camera, RayTracer, raytrace_camera  # noqa: E402
from oracle import iron_ref as R  # noqa: E402
from _util import oracle_scene  # noqa: E402

g = dict(np.load(os.path.join(ROOT, "tests", "golden", "g15_train_edges_S1.npz")))
size = int(g["W"])
K, W2C = torch.from_numpy(g["K"]), torch.from_numpy(g["W2C"])
dem = torch.from_numpy(g["depth_edge_mask_input"])
cpu_nets = scenes.build_networks("S1")
sc = oracle_scene(cpu_nets)
ref = R.raytrace_camera_full(sc, R.CameraSpec(size, size, K, W2C), max_num_rays=50000, detect_edges=True, depth_edge_mask=dem)
nets = {k: v.cuda() for k, v in cpu_nets.items()}
cam = Camera(size, size, K.cuda(), W2C.cuda())
res = raytrace_camera(cam, nets["sdf_network"], RayTracer(), max_num_rays=50000, detect_edges=True, depth_edge_mask=dem.cuda())
ih, ir = res["edge_pixel_idx"].cpu().numpy(), ref["edge_pixel_idx"].numpy()
assert np.array_equal(np.sort(ih), np.sort(ir))
oh, orr = np.argsort(ih), np.argsort(ir)
ph, pr = res["edge_points"].cpu().numpy()[oh], ref["edge_points"].numpy()[orr]
d = np.linalg.norm(ph - pr, axis=1)
print("edge pixels %d; |dp| median %.2e p90 %.2e max %.2e; > 1e-4: %d, > 5e-4: %d" % (len(d), np.median(d), np.percentile(d, 90), d.max(),
                                                                                   int((d > 1e-4).sum()), int((d > 5e-4).sum())))
cam_o = cam.get_camera_origin().reshape(1, 3).cpu()
for name, pts in (("hip", ph), ("oracle", pr)):
    p = torch.from_numpy(pts)
    s, _, n = R.sdf_get_all(sc.sdf_sd, sc.sdf_spec, p)
    n = n / n.norm(dim=-1, keepdim=True)
    v = cam_o - p
    v = v / v.norm(dim=-1, keepdim=True)
    dot = (n * v).sum(-1).numpy()
    print(name, "|n.v| at the edge points: min %.4f max %.4f; |sdf| max %.2e" % (np.abs(dot).min(), np.abs(dot).max(), float(s.abs().max())))
top = np.argsort(-d)[:8]
for i in top:
    print("pixel %5d  |dp| %.3e  hip %s  oracle %s" % (np.sort(ih)[i], d[i], ph[i].round(5).tolist(), pr[i].round(5).tolist()))
# the same against the reference's own stored edge colours is not possible (G15 keeps gradients only)
