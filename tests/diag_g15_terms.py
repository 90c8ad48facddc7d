"""Diagnostic (not collected by pytest): which term of the edge-sampling training loss carries the HIP-vs-reference
gradient discrepancy of G15.  The loss is split into (A) non-edge pixels, (B) edge pixels' colour, (C) edge pixels' normal;
for each, the SDF network's gradient from the HIP path is compared with torch.autograd over the CPU oracle.
    python tests/diag_g15_terms.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from iron_amd import scenes  # noqa: E402
from iron_amd.raytracer import Camera, RayTracer, render_camera  # noqa: E402
from iron_amd.renderer_ggx import GGXColocatedRenderer  # noqa: E402
from iron_amd.rendering_func import make_render_fn  # noqa: E402
from oracle import iron_ref as R  # noqa: E402
from oracle import train_ref as T  # noqa: E402
from _util import cpu_sd, golden_meta, tables  # noqa: E402

NETS = ("sdf_network", "diffuse_albedo_network", "specular_albedo_network", "specular_roughness_network")
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "g15_train_edges_S1.npz")))
size = int(g["W"])
K, W2C = torch.from_numpy(g["K"]), torch.from_numpy(g["W2C"])
dem = torch.from_numpy(g["depth_edge_mask_input"])
wt = torch.from_numpy(g["loss_weights"])
em = torch.from_numpy(g["edge_mask"]).bool()
mt, md = tables()
torch.set_num_threads(16)


def terms(res, wt, em):
    return {"A non-edge colour+normal": ((res["color"] * wt)[~em]).sum() + 0.1 * ((res["normal"] * wt)[~em]).sum(),
            "B edge colour": ((res["color"] * wt)[em]).sum(),
            "C edge normal": 0.1 * ((res["normal"] * wt)[em]).sum()}


def flat_grads(params):
    return {k: (p.grad.detach().reshape(-1).double().cpu().numpy().copy() if p.grad is not None else None) for k, p in params.items()}


out = {}
for name in ("A non-edge colour+normal", "B edge colour", "C edge normal"):
    cpu_nets = scenes.build_networks("S1")
    sd = {k: T.leaf_state(cpu_sd(cpu_nets[k])) for k in NETS}
    sc = R.Scene(sd["sdf_network"], R.SDFSpec(), {k: (sd[k], R.GGX_SPECS[k]) for k in R.GGX_SPECS}, golden_meta()["light"], mt, md)
    ref = T.render_camera_edges_train(sc, R.CameraSpec(size, size, K, W2C), dem)
    terms(ref, wt, em)[name].backward()
    gr = {"%s/%s" % (n, k): (v.grad.reshape(-1).double().numpy().copy() if v.grad is not None else None) for n in NETS for k, v in sd[n].items()}
    nets = {k: v.cuda() for k, v in scenes.build_networks("S1").items()}
    cam = Camera(size, size, K.cuda(), W2C.cuda())
    res = render_camera(cam, nets["sdf_network"], RayTracer(), nets, make_render_fn(GGXColocatedRenderer(use_cuda=True)),
                        fill_holes=False, handle_edges=True, is_training=True, depth_edge_mask=dem.cuda())
    terms(res, wt.cuda(), em.cuda())[name].backward()
    worst = {}
    for n in NETS:
        for pname, p in nets[n].named_parameters():
            r = gr.get("%s/%s" % (n, pname))
            if r is None or p.grad is None or np.abs(r).max() < 1e-12:
                continue
            h = p.grad.reshape(-1).double().cpu().numpy()
            e = np.linalg.norm(h - r) / np.linalg.norm(r)
            worst[n] = max(worst.get(n, 0.0), e)
    sdf_norm = np.sqrt(sum(np.sum(v ** 2) for k, v in gr.items() if k.startswith("sdf_network/") and v is not None))
    print("%-28s |grad sdf_network| %.4e   worst rel-L2 per network: %s" % (name, sdf_norm, {k: "%.2e" % v for k, v in worst.items()}), flush=True)
