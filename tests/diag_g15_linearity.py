"""Diagnostic (not collected): linearity of the HIP backward over the loss terms of G15 -- the gradient of (A + B + C) in one
backward pass must equal the sum of the three separate passes.    python tests/diag_g15_linearity.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iron_amd import scenes  # noqa: E402
from iron_amd.raytracer import Camera, RayTracer, render_camera  # noqa: E402
from iron_amd.renderer_ggx import GGXColocatedRenderer  # noqa: E402
from iron_amd.rendering_func import make_render_fn  # noqa: E402

g = dict(np.load(os.path.join(ROOT, "tests", "golden", "g15_train_edges_S1.npz")))
size = int(g["W"])
K, W2C = torch.from_numpy(g["K"]).cuda(), torch.from_numpy(g["W2C"]).cuda()
dem = torch.from_numpy(g["depth_edge_mask_input"]).cuda()
wt = torch.from_numpy(g["loss_weights"]).cuda()
em = torch.from_numpy(g["edge_mask"]).bool().cuda()


def run(which, retain=False):
    nets = {k: v.cuda() for k, v in scenes.build_networks("S1").items()}
    cam = Camera(size, size, K, W2C)
    res = render_camera(cam, nets["sdf_network"], RayTracer(), nets, make_render_fn(GGXColocatedRenderer(use_cuda=True)),
                        fill_holes=False, handle_edges=True, is_training=True, depth_edge_mask=dem)
    t = {"A": ((res["color"] * wt)[~em]).sum() + 0.1 * ((res["normal"] * wt)[~em]).sum(), "B": ((res["color"] * wt)[em]).sum(),
         "C": 0.1 * ((res["normal"] * wt)[em]).sum()}
    t["ABC"] = t["A"] + t["B"] + t["C"]
    t["full"] = (res["color"] * wt).sum() + 0.1 * (res["normal"] * wt).sum()
    t[which].backward()
    return torch.cat([p.grad.reshape(-1).double() for p in nets["sdf_network"].parameters()])


gs = {k: run(k) for k in ("A", "B", "C", "ABC", "full")}
s = gs["A"] + gs["B"] + gs["C"]
for k in ("ABC", "full"):
    print("%-4s vs A+B+C: rel-L2 %.3e   (|g| %.4e)" % (k, float((gs[k] - s).norm() / s.norm()), float(gs[k].norm())))
print("full vs ABC: %.3e;  run-to-run (full twice): %.3e" % (float((gs["full"] - gs["ABC"]).norm() / s.norm()),
                                                          float((run("full") - gs["full"]).norm() / s.norm())))
for k in ("A", "B", "C"):
    print(k, "%.4e" % float(gs[k].norm()))
