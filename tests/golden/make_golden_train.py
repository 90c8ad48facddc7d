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


if __name__ == "__main__" and not ({"--floor", "--c3"} & set(sys.argv)):
    if "--edges" in sys.argv:
        edges()
    else:
        main()


# ----------------------------------------------------------------------------------------------------------------------
# Round 2: (a) the reference's OWN fp32-vs-fp64 gradient discrepancy for the G15 setting (the conditioning floor the
# G15 tolerance is a multiple of), (b) G17 = BASELINE config C3 at its real size (512x512, edge sampling in the graph).
# ----------------------------------------------------------------------------------------------------------------------
def analytic_weights(h: int, w: int) -> np.ndarray:
    """Loss weights that need no storage: a smooth, sign-changing pattern computed in float64 and rounded to fp32."""
    y, x, c = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), np.arange(3, dtype=np.float64), indexing="ij")
    return (0.2 + 0.5 * np.sin(0.37 * x + 0.11 * y + 1.3 * c) + 0.3 * np.cos(0.05 * x - 0.23 * y)).astype(np.float32)


def edge_training_render(size: int, dtype, depth_edge_mask=None, nets=None):
    """The reference chain of G15 at `size` x `size` in `dtype`: raytrace_camera -> (mask input) -> locate_edge_points ->
    render_normal_and_color(is_training) -> render_edge_pixels(is_training).  Returns (nets, results, depth_edge_mask)."""
    from oracle import iron_ref as R  # only for the unpinned sobel mask, as in G8 / G15
    nets = nets or MG.build_reference_networks("S1")
    if dtype == torch.float64:
        nets = {k: v.double() for k, v in nets.items()}
    sdf_net = nets["sdf_network"]
    tracer = RayTracer()
    cam = MG.fixture_camera(size, size)
    if dtype == torch.float64:
        cam = MG.Camera64(cam.W, cam.H, cam.K.double(), cam.W2C.double())
    fn = MG.make_render_fn(nets, GGXColocatedRenderer(use_cuda=False), dtype)
    res = raytrace_camera(cam, sdf_net, tracer, max_num_rays=50000, fill_holes=False, detect_edges=False)
    if depth_edge_mask is None:
        depth_edge_mask = (R.sobel_magnitude(res["depth"].float()) > 1e-2) & res["convergent_mask"]
    mask_in = depth_edge_mask & res["convergent_mask"]
    with torch.no_grad():
        res.update(locate_edge_points(cam, res["points"], sdf_net, max_step=16, step_size=1e-3, dot_threshold=5e-2,
                                      max_num_rays=50000, mask=mask_in))
    res["convergent_mask"] &= ~res["edge_mask"]
    render_normal_and_color(res, sdf_net, nets, fn, is_training=True, max_num_pts=320000)
    _cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        render_edge_pixels(res, cam, sdf_net, tracer, nets, fn, is_training=True)
    finally:
        torch.Tensor.cuda = _cuda
    return nets, res, depth_edge_mask, cam


def grads_of(nets):
    out = {}
    for name in NETS:
        for pname, p in nets[name].named_parameters():
            out["%s/%s" % (name, pname)] = p.grad.reshape(-1).double().numpy().copy()
    return out


def floor_g15():
    """The reference's fp64 run of the G15 setting with the SAME loss weights and the SAME depth-edge-mask input; stored next
    to G15: per tensor the fp64 gradient norm / samples and the fp32-vs-fp64 discrepancy (norm error, worst sampled entry)."""
    g15 = dict(np.load(os.path.join(HERE, "g15_train_edges_S1.npz")))
    wt = torch.from_numpy(g15["loss_weights"])
    dem = torch.from_numpy(g15["depth_edge_mask_input"])
    nets64, res64, _, _ = edge_training_render(96, torch.float64, depth_edge_mask=dem)
    loss = (res64["color"] * wt.double()).sum() + 0.1 * (res64["normal"] * wt.double()).sum()
    loss.backward()
    g64 = grads_of(nets64)
    out = {"loss": np.float64(loss.item()), "edge_mask": npf(res64["edge_mask"]), "convergent_mask": npf(res64["convergent_mask"]),
           "color": npf(res64["color"])}
    worst_n = worst_s = 0.0
    for key, g in g64.items():
        idx = sample_idx(g.size)
        out["gnorm:" + key] = np.float64(np.linalg.norm(g))
        out["gsample:" + key] = g[idx]
        n32, s32 = float(g15["gnorm:" + key]), g15["gsample:" + key]
        en = abs(n32 - np.linalg.norm(g)) / max(np.linalg.norm(g), 1e-12)
        es = float(np.abs(s32 - g[idx]).max() / max(np.abs(g[idx]).max(), 1e-12))
        out["floor_n:" + key] = np.float64(en)
        out["floor_s:" + key] = np.float64(es)
        worst_n, worst_s = max(worst_n, en), max(worst_s, es)
    np.savez_compressed(os.path.join(HERE, "g15_floor_fp64.npz"), **out)
    meta_path = os.path.join(HERE, "meta.json")
    meta = json.load(open(meta_path))
    meta["g15_ref32_vs_ref64_worst_gnorm_err"] = worst_n
    meta["g15_ref32_vs_ref64_worst_gsample_err"] = worst_s
    meta["g15_edge_mask_flips_32_64"] = int((out["edge_mask"] != g15["edge_mask"]).sum())
    meta["g15_colour_rel_l2_ref32_ref64"] = float(np.linalg.norm(g15["color"] - out["color"]) / np.linalg.norm(out["color"]))
    json.dump(meta, open(meta_path, "w"), indent=1, sort_keys=True)
    print("G15 floor: ref32 vs ref64 worst gnorm err %.3e, worst sampled entry %.3e, edge-mask flips %d" % (
        worst_n, worst_s, meta["g15_edge_mask_flips_32_64"]))


def c3(size: int = 512, with64: bool = True):
    """G17: BASELINE config C3 at its real size -- S1, size x size, edge sampling in the graph, the real reference's gradients
    (fp32; with64: also fp64 + the per-tensor fp32-vs-fp64 floor).  The colour image is stored on the [::4, ::4] sub-lattice."""
    import time
    t0 = time.time()
    nets, res, dem, cam = edge_training_render(size, torch.float32)
    wt = torch.from_numpy(analytic_weights(size, size))
    loss = (res["color"] * wt).sum() + 0.1 * (res["normal"] * wt).sum()
    loss.backward()
    t32 = time.time() - t0
    g32 = grads_of(nets)
    out = {"K": npf(cam.K), "W2C": npf(cam.W2C), "W": np.int64(size), "H": np.int64(size), "loss": np.float64(loss.item()),
           "color_sub4": npf(res["color"])[::4, ::4].copy(), "convergent_mask_bits": np.packbits(npf(res["convergent_mask"])),
           "edge_mask_bits": np.packbits(npf(res["edge_mask"])), "depth_edge_mask_input_bits": np.packbits(npf(dem)),
           "n_hits": np.int64(int(res["convergent_mask"].sum())), "n_edge": np.int64(int(res["edge_mask"].sum())),
           "ref_seconds_fp32": np.float64(t32)}
    for key, g in g32.items():
        out["gnorm:" + key] = np.float64(np.linalg.norm(g))
        out["gsample:" + key] = g[sample_idx(g.size)]
    print("G17 fp32: %.0f s, loss %.6f, hits %d, edge pixels %d" % (t32, loss.item(), int(out["n_hits"]), int(out["n_edge"])), flush=True)
    if with64:
        t0 = time.time()
        nets64, res64, _, _ = edge_training_render(size, torch.float64, depth_edge_mask=dem)
        loss64 = (res64["color"] * wt.double()).sum() + 0.1 * (res64["normal"] * wt.double()).sum()
        loss64.backward()
        out["ref_seconds_fp64"] = np.float64(time.time() - t0)
        out["loss_fp64"] = np.float64(loss64.item())
        out["edge_mask_flips_32_64"] = np.int64(int((res64["edge_mask"] != res["edge_mask"]).sum()))
        out["mask_flips_32_64"] = np.int64(int((res64["convergent_mask"] != res["convergent_mask"]).sum()))
        worst_n = worst_s = 0.0
        for key, g in grads_of(nets64).items():
            idx = sample_idx(g.size)
            out["gnorm64:" + key] = np.float64(np.linalg.norm(g))
            out["gsample64:" + key] = g[idx]
            en = abs(float(out["gnorm:" + key]) - np.linalg.norm(g)) / max(np.linalg.norm(g), 1e-12)
            es = float(np.abs(out["gsample:" + key] - g[idx]).max() / max(np.abs(g[idx]).max(), 1e-12))
            out["floor_n:" + key] = np.float64(en)
            out["floor_s:" + key] = np.float64(es)
            worst_n, worst_s = max(worst_n, en), max(worst_s, es)
        print("G17 fp64: %.0f s; ref32 vs ref64 worst gnorm err %.3e, worst sampled entry %.3e, edge flips %d, mask flips %d" % (
            float(out["ref_seconds_fp64"]), worst_n, worst_s, int(out["edge_mask_flips_32_64"]), int(out["mask_flips_32_64"])), flush=True)
    np.savez_compressed(os.path.join(HERE, "g17_train_c3_S1_%d.npz" % size), **out)


if __name__ == "__main__" and "--floor" in sys.argv:
    floor_g15()
if __name__ == "__main__" and "--c3" in sys.argv:
    c3(int(sys.argv[sys.argv.index("--c3") + 1]) if len(sys.argv) > sys.argv.index("--c3") + 1 else 512)
