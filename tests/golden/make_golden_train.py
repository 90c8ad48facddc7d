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


if __name__ == "__main__" and not ({"--floor", "--floor14", "--c3", "--stable"} & set(sys.argv)):
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


def floor_g14():
    """Round 3: the reference's fp64 run of the G14 setting (same crop, same loss weights) and, per tensor, its own fp32-vs-fp64
    gradient discrepancy -- the conditioning floor of G14.  (The GGX lobe makes d loss / d roughness at a highlight pixel move by
    ~1e-2 per 5e-7 of normal direction: two fp32-accurate evaluations of the normal differ by more than G14's original flat
    5e-4 bound on the roughness net's tensors.)"""
    g14 = dict(np.load(os.path.join(HERE, "g14_train_S1_c32.npz")))
    nets = {k: v.double() for k, v in MG.build_reference_networks("S1").items()}
    cam512 = MG.fixture_camera(512, 512)
    cam, _, _ = cam512.crop_region(32, 32, ul_corner=(240, 240))
    cam = MG.Camera64(cam.W, cam.H, cam.K.double(), cam.W2C.double())
    fn = MG.make_render_fn(nets, GGXColocatedRenderer(use_cuda=False), torch.float64)
    res = render_camera(cam, nets["sdf_network"], RayTracer(), nets, fn, fill_holes=False, handle_edges=False, is_training=True)
    wt = torch.from_numpy(g14["loss_weights"]).double()
    loss = (res["color"] * wt).sum() + 0.1 * (res["normal"] * wt).sum()
    loss.backward()
    out = {"loss": np.float64(loss.item()), "convergent_mask": npf(res["convergent_mask"]), "color": npf(res["color"]), "normal": npf(res["normal"])}
    worst_n = worst_s = 0.0
    for key, g in grads_of(nets).items():
        idx = sample_idx(g.size)
        out["gnorm:" + key] = np.float64(np.linalg.norm(g))
        out["gsample:" + key] = g[idx]
        n32, s32 = float(g14["gnorm:" + key]), g14["gsample:" + key]
        en = abs(n32 - np.linalg.norm(g)) / max(np.linalg.norm(g), 1e-12)
        es = float(np.abs(s32 - g[idx]).max() / max(np.abs(g[idx]).max(), 1e-12))
        out["floor_n:" + key] = np.float64(en)
        out["floor_s:" + key] = np.float64(es)
        worst_n, worst_s = max(worst_n, en), max(worst_s, es)
        print("%-50s floor_n %.2e floor_s %.2e" % (key, en, es))
    np.savez_compressed(os.path.join(HERE, "g14_floor_fp64.npz"), **out)
    meta_path = os.path.join(HERE, "meta.json")
    meta = json.load(open(meta_path))
    meta["g14_ref32_vs_ref64_worst_gnorm_err"] = worst_n
    meta["g14_ref32_vs_ref64_worst_gsample_err"] = worst_s
    meta["g14_mask_flips_32_64"] = int((out["convergent_mask"] != g14["convergent_mask"]).sum())
    json.dump(meta, open(meta_path, "w"), indent=1, sort_keys=True)
    print("G14 floor: ref32 vs ref64 worst gnorm err %.3e, worst sampled entry %.3e, mask flips %d" % (worst_n, worst_s, meta["g14_mask_flips_32_64"]))


if __name__ == "__main__" and "--floor14" in sys.argv:
    floor_g14()
if __name__ == "__main__" and "--floor" in sys.argv:
    floor_g15()
if __name__ == "__main__" and "--c3" in sys.argv:
    c3(int(sys.argv[sys.argv.index("--c3") + 1]) if len(sys.argv) > sys.argv.index("--c3") + 1 else 512)


# ----------------------------------------------------------------------------------------------------------------------
# G15s: the G15 setting with the ill-conditioned edge pixels taken out of the loss.  reparam_points divides by
# clamp(n . (-d), 1e-4) (raytracer.py:17-24): a side ray that grazes the surface enters the gradient with a factor of up to
# 1e4 and its hit point is only defined to sdf_threshold / (n . d) along the ray, so a handful of such pixels decide how well
# ANY two runs agree (the reference's own fp32 and fp64 runs included).  The pixels whose side rays hit with
# |n . d| < GRAZING are found here, by the reference's own tracer on the reference's own side samples, and stored as an
# INPUT mask; fp32 and fp64 gradients of the masked loss are the golden values.
# ----------------------------------------------------------------------------------------------------------------------
GRAZING = 0.1
# ... and the blend weight 1 - (a - sin a) / 2pi, a = 2 acos(x), x = clamp(h / 0.707, 0, 1) (raytracer.py:696-698) has the
# derivative -2 / sqrt(1 - x^2) in x: unbounded as the edge point approaches the rim of the pixel disc, and with a relative
# sensitivity x / (1 - x^2) to the position itself -- 1e-5 px of walk rounding is 1e-3 of gradient at x = 0.995.  Pixels with
# x > RIM are taken out as well, and so are pixels whose edge point the reference's own fp32 and fp64 walks place more than
# WALK_TOL apart (the 16-step walk amplifies rounding; such a pixel is not defined to better than that by the reference).
RIM = 0.95
WALK_TOL = 1e-5
# ... and clamp(x, 0, 1) passes gradient for x >= 0 and none for x < 0 (torch.clamp's backward).  A candidate that already
# satisfies |n.v| <= 0.05 is "found" without walking, so its edge point is the pixel's own hit point and projects onto the
# pixel CENTRE: x = 0 up to the rounding of the projection (1e-6 px), and whether that pixel's blend weight has a gradient at
# all (dB/d sdf ~ 100 of a total norm ~500 in G15) is a coin flip of that rounding -- the reference's fp32 and fp64 runs can
# disagree on it, and so does every other arithmetic.  Pixels with |x| < CENTRE_TOL are taken out.
CENTRE_TOL = 1e-3


def side_ray_cosines(nets, res, cam):
    """|n . d| at the hit of each side ray of every edge pixel (NaN where the side ray misses): [n_edge, 2]."""
    from models.raytracer import raytrace_pixels
    sdf_net = nets["sdf_network"]
    pts = res["edge_points"].detach()
    uv = res["edge_uv"].detach()
    g = sdf_net.gradient(pts).detach().reshape(-1, 3)
    n3 = g / (g.norm(dim=-1, keepdim=True) + 1e-10)
    n2 = torch.matmul(n3, cam.W2C[:3, :3].transpose(1, 0))[:, :2]
    n2 = n2 / (n2.norm(dim=-1, keepdim=True) + 1e-10)
    centre = torch.floor(uv) + 0.5
    out = []
    for side_uv in (centre - 0.707 * n2, centre + 0.707 * n2):
        with torch.no_grad():
            r = raytrace_pixels(sdf_net, RayTracer(), side_uv, cam)
        hit = r["convergent_mask"]
        cosv = torch.full((uv.shape[0],), float("nan"), dtype=uv.dtype)
        if hit.any():
            gn = sdf_net.gradient(r["points"][hit]).detach().reshape(-1, 3)
            gn = gn / (gn.norm(dim=-1, keepdim=True) + 1e-10)
            cosv[hit] = (gn * r["ray_d"][hit]).sum(-1).abs()
        out.append(cosv)
    return torch.stack(out, dim=-1)


def stable_g15():
    g15 = dict(np.load(os.path.join(HERE, "g15_train_edges_S1.npz")))
    wt = torch.from_numpy(g15["loss_weights"])
    dem = torch.from_numpy(g15["depth_edge_mask_input"])
    # 1. which edge pixels are ill-conditioned: from the fp64 run (the better-defined one)
    nets64, res64, _, cam64 = edge_training_render(96, torch.float64, depth_edge_mask=dem)
    cos = side_ray_cosines(nets64, res64, cam64)
    grazing = (cos < GRAZING).any(dim=-1)
    # position of the edge point across its pixel disc, as render_edge_pixels computes it
    pts64, uv64 = res64["edge_points"].detach(), res64["edge_uv"].detach()
    g = nets64["sdf_network"].gradient(pts64).detach().reshape(-1, 3)
    n3 = g / (g.norm(dim=-1, keepdim=True) + 1e-10)
    n2 = torch.matmul(n3, cam64.W2C[:3, :3].transpose(1, 0))[:, :2]
    n2 = n2 / (n2.norm(dim=-1, keepdim=True) + 1e-10)
    x = (((uv64 - (torch.floor(uv64) + 0.5)) * n2).sum(-1) / 0.707)
    rim = (x > RIM) | (x.abs() < CENTRE_TOL)
    nets32, res32, _, _ = edge_training_render(96, torch.float32, depth_edge_mask=dem)
    assert torch.equal(res32["edge_pixel_idx"], res64["edge_pixel_idx"])
    walk_gap = (res32["edge_points"].detach().double() - pts64).norm(dim=-1)
    chaotic = walk_gap > WALK_TOL
    bad = grazing | rim | chaotic
    keep = torch.ones(96 * 96, dtype=torch.bool)
    keep[res64["edge_pixel_idx"][bad]] = False
    keep = keep.reshape(96, 96)
    # 1b. knife-edge pixels of ANY kind (tracer stop |sdf| <= 5e-5, sampler sign, bisection count, walk stop |n.v| <= 0.05):
    # the forward render is repeated with every SDF weight_v entry nudged by a relative N(0, 2^-23) -- one rounding error's worth --
    # and a pixel whose colour moves by more than 2e-5 of the image maximum in any of the realisations is not defined by the
    # algorithm to better than that (on the GPU box's EPYC host the bit-identical torch oracle moves such pixels by 5e-4
    # against this container's Xeon: tests/diag_g15_host.py).
    base = res32["color"].detach()
    spread = torch.zeros(96, 96)
    edge_flips = 0
    for k in range(6):
        netsk = MG.build_reference_networks("S1")
        gen = torch.Generator().manual_seed(100 + k)
        with torch.no_grad():
            for name, p in netsk["sdf_network"].named_parameters():
                if name.endswith("weight_v"):
                    p.mul_(1.0 + torch.randn(p.shape, generator=gen) * 2.0 ** -23)
        with torch.no_grad():
            pass
        _, resk, _, _ = edge_training_render(96, torch.float32, depth_edge_mask=dem, nets=netsk)
        edge_flips += int((resk["edge_mask"] != res32["edge_mask"]).sum())
        spread = torch.maximum(spread, (resk["color"].detach() - base).abs().max(dim=-1)[0])
        print("   realisation %d: colour max|d| %.2e, pixels over tolerance so far %d" % (k, float(spread.max()), int((spread > 2e-5 * float(base.max())).sum())), flush=True)
    knife = spread > 2e-5 * float(base.max())
    print("G15s: %d knife-edge pixels (%d of them edge pixels), edge-mask flips over the realisations %d" % (
        int(knife.sum()), int((knife & res32["edge_mask"]).sum()), edge_flips), flush=True)
    keep &= ~knife
    print("G15s: of %d edge pixels %d have a side ray with |n.d| < %.2f, %d sit at x > %.2f of the pixel disc or at |x| < 1e-3 (the clamp's corner), %d have fp32/fp64 walks more "
          "than %.0e apart (median gap %.1e, max %.1e): %d masked" % (cos.shape[0], int(grazing.sum()), GRAZING, int(rim.sum()), RIM,
                                                                      int(chaotic.sum()), WALK_TOL, float(walk_gap.median()), float(walk_gap.max()), int(bad.sum())), flush=True)
    wts = wt * keep[..., None].float()
    # 2. gradients of the masked loss, fp64 and fp32
    loss64 = (res64["color"] * wts.double()).sum() + 0.1 * (res64["normal"] * wts.double()).sum()
    loss64.backward()
    g64 = grads_of(nets64)
    loss32 = (res32["color"] * wts).sum() + 0.1 * (res32["normal"] * wts).sum()
    loss32.backward()
    g32 = grads_of(nets32)
    out = {"stable_pixel_mask": npf(keep), "grazing_threshold": np.float64(GRAZING), "n_masked_edge_pixels": np.int64(int(bad.sum())),
           "loss": np.float64(loss32.item()), "loss_fp64": np.float64(loss64.item()), "side_ray_cosines_fp64": npf(cos),
           "disc_position_fp64": npf(x), "walk_gap_32_64": npf(walk_gap), "edge_pixel_idx": npf(res64["edge_pixel_idx"]),
           "edge_points_fp64": npf(pts64), "colour_spread_over_realisations": npf(spread)}
    worst_n = worst_s = 0.0
    for key in g64:
        idx = sample_idx(g64[key].size)
        n32, n64 = np.linalg.norm(g32[key]), np.linalg.norm(g64[key])
        out["gnorm:" + key], out["gsample:" + key] = np.float64(n32), g32[key][idx]
        out["gnorm64:" + key], out["gsample64:" + key] = np.float64(n64), g64[key][idx]
        en = abs(n32 - n64) / max(n64, 1e-12)
        es = float(np.abs(g32[key][idx] - g64[key][idx]).max() / max(np.abs(g64[key][idx]).max(), 1e-12))
        out["floor_n:" + key], out["floor_s:" + key] = np.float64(en), np.float64(es)
        worst_n, worst_s = max(worst_n, en), max(worst_s, es)
    np.savez_compressed(os.path.join(HERE, "g15s_train_edges_stable_S1.npz"), **out)
    meta_path = os.path.join(HERE, "meta.json")
    meta = json.load(open(meta_path))
    meta["g15s_ref32_vs_ref64_worst_gnorm_err"] = worst_n
    meta["g15s_ref32_vs_ref64_worst_gsample_err"] = worst_s
    meta["g15s_masked_edge_pixels"] = int(bad.sum())
    json.dump(meta, open(meta_path, "w"), indent=1, sort_keys=True)
    print("G15s floor: ref32 vs ref64 worst gnorm err %.3e, worst sampled entry %.3e" % (worst_n, worst_s))


if __name__ == "__main__" and "--stable" in sys.argv:
    stable_g15()
