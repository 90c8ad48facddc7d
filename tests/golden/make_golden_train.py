"""Golden vectors for SURVEY 8 row f-2 (stage-2 training forward + backward, BASELINE config C3) -- BUILD CONTAINER ONLY.

The real reference's render_camera(is_training=True, fill_holes=False, handle_edges=False) on the 32x32 centre crop of
the fixture camera (scene S1), a fixed linear functional of the rendered colour as loss, torch.autograd through the
reference's graph (reparam_points, double-backward through the SDF MLP, material nets, GGX).  Stored: the training-mode
colour image, the loss, and for EVERY parameter tensor of the four networks its gradient norm and 48 sampled entries.

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_train.py
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

from models.raytracer import (RayTracer, locate_edge_points, raytrace_camera, render_camera, render_edge_pixels,  # noqa: E402  (reference)
                              render_normal_and_color)
from models.renderer_ggx import GGXColocatedRenderer  # noqa: E402

npf = MG.npf
NETS = ("sdf_network", "diffuse_albedo_network", "specular_albedo_network", "specular_roughness_network")


def sample_idx(n: int) -> np.ndarray:
    """48 fixed positions of a flattened tensor of n elements: the first 16, then 32 on an even stride."""
    head = np.arange(min(16, n))
    rest = np.linspace(0, n - 1, 32).astype(np.int64)
    return np.concatenate([head, rest])


def main():
    nets = MG.build_reference_networks("S1")
    cam512 = MG.fixture_camera(512, 512)
    cam, _, _ = cam512.crop_region(32, 32, ul_corner=(240, 240))
    fn = MG.make_render_fn(nets, GGXColocatedRenderer(use_cuda=False), torch.float32)
    res = render_camera(cam, nets["sdf_network"], RayTracer(), nets, fn, fill_holes=False, handle_edges=False, is_training=True)
    gen = torch.Generator().manual_seed(31)
    wt = torch.rand(32, 32, 3, generator=gen) - 0.3
    loss = (res["color"] * wt).sum() + 0.1 * (res["normal"] * wt).sum()
    loss.backward()
    out = {"K": npf(cam.K), "W2C": npf(cam.W2C), "W": np.int64(32), "H": np.int64(32), "loss_weights": npf(wt),
           "loss": np.float64(loss.item()), "color": npf(res["color"]), "normal": npf(res["normal"]),
           "convergent_mask": npf(res["convergent_mask"])}
    n_params = 0
    for name in NETS:
        for pname, p in nets[name].named_parameters():
            assert p.grad is not None, (name, pname)
            g = p.grad.reshape(-1).double().numpy()
            key = "%s/%s" % (name, pname)
            out["gnorm:" + key] = np.float64(np.linalg.norm(g))
            out["gsample:" + key] = g[sample_idx(g.size)]
            n_params += 1
    np.savez_compressed(os.path.join(HERE, "g14_train_S1_c32.npz"), **out)
    meta_path = os.path.join(HERE, "meta.json")
    meta = json.load(open(meta_path))
    meta["n_param_tensors_train_golden"] = n_params
    json.dump(meta, open(meta_path, "w"), indent=1, sort_keys=True)
    print("loss", loss.item(), "hits", int(res["convergent_mask"].sum()), "param tensors", n_params,
          "sdf lin0.weight_v grad norm", float(nets["sdf_network"].lin0.weight_v.grad.norm()))


def store_grads(nets, out):
    n_params = 0
    for name in NETS:
        for pname, p in nets[name].named_parameters():
            assert p.grad is not None, (name, pname)
            g = p.grad.reshape(-1).double().numpy()
            key = "%s/%s" % (name, pname)
            out["gnorm:" + key] = np.float64(np.linalg.norm(g))
            out["gsample:" + key] = g[sample_idx(g.size)]
            n_params += 1
    return n_params


def edges():
    """G15: the same with silhouette edge sampling in the graph (render_camera(handle_edges=True, is_training=True), the
    setting render_surface.py trains with): 96x96 view of S1, the depth-edge mask an INPUT as in G8 (kornia is absent, so
    detect_edges itself cannot run in the reference), then the reference's locate_edge_points -> render_normal_and_color ->
    render_edge_pixels, all with is_training=True."""
    from oracle import iron_ref as R  # only for the unpinned sobel mask, exactly as make_golden.py does for G8
    nets = MG.build_reference_networks("S1")
    sdf_net = nets["sdf_network"]
    tracer = RayTracer()
    cam = MG.fixture_camera(96, 96)
    fn = MG.make_render_fn(nets, GGXColocatedRenderer(use_cuda=False), torch.float32)
    res = raytrace_camera(cam, sdf_net, tracer, max_num_rays=50000, fill_holes=False, detect_edges=False)
    depth_edge_mask = (R.sobel_magnitude(res["depth"]) > 1e-2) & res["convergent_mask"]
    with torch.no_grad():
        res.update(locate_edge_points(cam, res["points"], sdf_net, max_step=16, step_size=1e-3, dot_threshold=5e-2,
                                      max_num_rays=50000, mask=depth_edge_mask))
    res["convergent_mask"] &= ~res["edge_mask"]
    render_normal_and_color(res, sdf_net, nets, fn, is_training=True, max_num_pts=320000)
    _cuda = torch.Tensor.cuda  # empty-chunk branch of render_normal_and_color calls .cuda() (raytracer.py:627-633); see make_golden.py
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        render_edge_pixels(res, cam, sdf_net, tracer, nets, fn, is_training=True)
    finally:
        torch.Tensor.cuda = _cuda
    gen = torch.Generator().manual_seed(33)
    wt = torch.rand(96, 96, 3, generator=gen) - 0.3
    loss = (res["color"] * wt).sum() + 0.1 * (res["normal"] * wt).sum()
    loss.backward()
    out = {"K": npf(cam.K), "W2C": npf(cam.W2C), "W": np.int64(96), "H": np.int64(96), "loss_weights": npf(wt),
           "loss": np.float64(loss.item()), "color": npf(res["color"]), "normal": npf(res["normal"]),
           "convergent_mask": npf(res["convergent_mask"]), "edge_mask": npf(res["edge_mask"]),
           "depth_edge_mask_input": npf(depth_edge_mask)}
    n_params = store_grads(nets, out)
    np.savez_compressed(os.path.join(HERE, "g15_train_edges_S1.npz"), **out)
    print("G15 loss", loss.item(), "edge pixels", int(res["edge_mask"].sum()), "param tensors", n_params)


if __name__ == "__main__":
    if "--edges" in sys.argv:
        edges()
    else:
        main()
