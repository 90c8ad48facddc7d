"""Diagnostic for the edge-sampling training gradients (G15 / G17): per-parameter-tensor error of the HIP path against the
reference's fp32 AND fp64 runs, next to the reference's own fp32-vs-fp64 discrepancy (the conditioning floor).
    python tools/g15_diag.py [g15|g17]"""
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

G = os.path.join(ROOT, "tests", "golden")
NETS = ("sdf_network", "diffuse_albedo_network", "specular_albedo_network", "specular_roughness_network")
which = sys.argv[1] if len(sys.argv) > 1 else "g15"


def sample_idx(n):
    return np.concatenate([np.arange(min(16, n)), np.linspace(0, n - 1, 32).astype(np.int64)])


def analytic_weights(h, w):
    y, x, c = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), np.arange(3, dtype=np.float64), indexing="ij")
    return (0.2 + 0.5 * np.sin(0.37 * x + 0.11 * y + 1.3 * c) + 0.3 * np.cos(0.05 * x - 0.23 * y)).astype(np.float32)


if which == "g15":
    g = dict(np.load(os.path.join(G, "g15_train_edges_S1.npz")))
    f = dict(np.load(os.path.join(G, "g15_floor_fp64.npz")))
    size = int(g["W"])
    wt = torch.from_numpy(g["loss_weights"]).cuda()
    dem = torch.from_numpy(g["depth_edge_mask_input"]).cuda()
    g64n = {k[6:]: float(v) for k, v in f.items() if k.startswith("gnorm:")}
    g64s = {k[8:]: v for k, v in f.items() if k.startswith("gsample:")}
elif which == "g15s":
    g0 = dict(np.load(os.path.join(G, "g15_train_edges_S1.npz")))
    g = dict(np.load(os.path.join(G, "g15s_train_edges_stable_S1.npz")))
    for k in ("W", "K", "W2C", "depth_edge_mask_input"):
        g[k] = g0[k]
    size = int(g["W"])
    wt = (torch.from_numpy(g0["loss_weights"]) * torch.from_numpy(g["stable_pixel_mask"])[..., None].float()).cuda()
    dem = torch.from_numpy(g["depth_edge_mask_input"]).cuda()
    g64n = {k[8:]: float(v) for k, v in g.items() if k.startswith("gnorm64:")}
    g64s = {k[10:]: v for k, v in g.items() if k.startswith("gsample64:")}
else:
    g = dict(np.load(os.path.join(G, "g17_train_c3_S1_512.npz")))
    size = int(g["W"])
    wt = torch.from_numpy(analytic_weights(size, size)).cuda()
    dem = torch.from_numpy(np.unpackbits(g["depth_edge_mask_input_bits"])[: size * size].reshape(size, size).astype(bool)).cuda()
    g64n = {k[8:]: float(v) for k, v in g.items() if k.startswith("gnorm64:")}
    g64s = {k[10:]: v for k, v in g.items() if k.startswith("gsample64:")}
g32n = {k[6:]: float(v) for k, v in g.items() if k.startswith("gnorm:")}
g32s = {k[8:]: v for k, v in g.items() if k.startswith("gsample:")}

nets = {k: v.cuda() for k, v in scenes.build_networks("S1").items()}
cam = Camera(size, size, torch.from_numpy(g["K"]).cuda(), torch.from_numpy(g["W2C"]).cuda())
res = render_camera(cam, nets["sdf_network"], RayTracer(), nets, make_render_fn(GGXColocatedRenderer(use_cuda=True)),
                    fill_holes=False, handle_edges=True, is_training=True, depth_edge_mask=dem)
loss = (res["color"] * wt).sum() + 0.1 * (res["normal"] * wt).sum()
loss.backward()
print("loss %.6f (ref32 %.6f)" % (loss.item(), float(g["loss"])), "edge pixels", int(res["edge_mask"].sum()))
print("%-52s %10s %10s %10s | %10s %10s %10s" % ("tensor", "n:hip~32", "n:hip~64", "n:32~64", "s:hip~32", "s:hip~64", "s:32~64"))
worst = np.zeros(6)
for name in NETS:
    for pname, p in nets[name].named_parameters():
        key = "%s/%s" % (name, pname)
        gr = p.grad.reshape(-1).double().cpu().numpy()
        idx = sample_idx(gr.size)
        n = np.linalg.norm(gr)
        row = [abs(n - g32n[key]) / max(g32n[key], 1e-12), abs(n - g64n[key]) / max(g64n[key], 1e-12), abs(g32n[key] - g64n[key]) / max(g64n[key], 1e-12),
               np.abs(gr[idx] - g32s[key]).max() / max(np.abs(g32s[key]).max(), 1e-12),
               np.abs(gr[idx] - g64s[key]).max() / max(np.abs(g64s[key]).max(), 1e-12),
               np.abs(g32s[key] - g64s[key]).max() / max(np.abs(g64s[key]).max(), 1e-12)]
        worst = np.maximum(worst, row)
        print("%-52s %10.2e %10.2e %10.2e | %10.2e %10.2e %10.2e" % (key, *row))
print("%-52s %10.2e %10.2e %10.2e | %10.2e %10.2e %10.2e" % ("WORST", *worst))
