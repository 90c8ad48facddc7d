"""Golden-vector generator (BUILD CONTAINER ONLY -- needs the reference at /root/reference).

Imports the real reference (arthurlirui/IRON) on CPU, runs its own functions for the stage-2
forward path on seeded inputs and writes the outputs as small .npz fixtures next to this file.
The reference never travels: only these data files are committed.  tests/test_oracle_golden.py
pins oracle/iron_ref.py to them; the GPU parity tests then compare the HIP path with the oracle.

Third-party modules that the reference imports at module scope but that are absent from this image
(`turtle`, `kornia`, `cv2`, `icecream`) are registered as EMPTY placeholder modules so that
`models/raytracer.py` can be imported; nothing of those libraries is emulated, and the code paths
that would call them (fill_holes / detect_edges / Camera.resize(image=...)) are never run here
(=> kornia closing/sobel stay "parity unpinned").

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

sys.path.insert(0, REPO)
sys.path.insert(0, REF)
for _n in ("turtle", "kornia", "cv2", "icecream"):
    if _n not in sys.modules:
        sys.modules[_n] = types.ModuleType(_n)
sys.modules["turtle"].update = lambda *a, **k: None
sys.modules["icecream"].ic = lambda *a, **k: None

from models.embedder import get_embedder  # noqa: E402  (reference)
from models.fields import RenderingNetwork, SDFNetwork  # noqa: E402
from models.raytracer import (Camera, RayTracer, intersect_sphere, locate_edge_points, raytrace_camera, render_camera,  # noqa: E402
                              render_edge_pixels, render_normal_and_color)
from models.renderer_ggx import GGXColocatedRenderer  # noqa: E402
from models.rendering_func import get_materials  # noqa: E402

torch.set_num_threads(8)


def build_reference_networks(scene: str, seed: int = 0):
    """Same construction order / seeds as iron_amd.scenes.build_networks, with the reference classes."""
    torch.manual_seed(seed)
    nets = {
        "sdf_network": SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5,
                                  scale=1.0, geometric_init=True, weight_norm=True),
        "diffuse_albedo_network": RenderingNetwork(d_in=9, d_out=3, d_feature=256, d_hidden=256, n_layers=4,
                                                   multires_view=4, mode="idr", squeeze_out=True),
        "specular_albedo_network": RenderingNetwork(d_in=6, d_out=3, d_feature=256, d_hidden=256, n_layers=4,
                                                    multires=6, multires_view=-1, mode="no_view_dir",
                                                    squeeze_out=False, output_bias=0.4, output_scale=0.1),
        "specular_roughness_network": RenderingNetwork(d_in=6, d_out=1, d_feature=256, d_hidden=256, n_layers=4,
                                                       multires=6, multires_view=-1, mode="no_view_dir",
                                                       squeeze_out=False, output_bias=0.1, output_scale=0.1),
    }
    if scene == "S1":
        g = torch.Generator().manual_seed(1)
        v = nets["sdf_network"].lin0.weight_v
        with torch.no_grad():
            v[:, 3:] += 0.01 * torch.randn(v[:, 3:].shape, generator=g)
    return nets


def state_hash(nets) -> str:
    h = hashlib.sha256()
    for name in sorted(nets):
        sd = nets[name].state_dict()
        for k in sorted(sd):
            h.update(k.encode())
            h.update(sd[k].detach().cpu().numpy().tobytes())
    return h.hexdigest()


LIGHT = 32.0


def make_render_fn(nets, renderer, dtype):
    """The logic of render_surface.py:117-156 (that script cannot be imported: it parses argv and calls
    torch.cuda at import), with buffers in `dtype` so the fp64 run works too."""

    def render_fn(interior_mask, color_network_dict, ray_o, ray_d, points, normals, features):
        sh = list(interior_mask.shape)
        rgb = torch.zeros(sh + [3], dtype=dtype)
        out = {k: rgb.clone() for k in ("color", "diffuse_color", "specular_color", "diffuse_albedo",
                                        "specular_albedo", "normal")}
        out["specular_roughness"] = rgb[..., 0].clone()
        if interior_mask.any():
            normals = normals / (normals.norm(dim=-1, keepdim=True) + 1e-10)
            params = get_materials(network_dict=color_network_dict, points=points, normals=normals, features=features)
            res = renderer(torch.tensor(LIGHT, dtype=dtype), (points - ray_o).norm(dim=-1, keepdim=True), normals,
                           -ray_d, params=params)
            out["color"][interior_mask] = res["rgb"]
            out["diffuse_color"][interior_mask] = res["diffuse_rgb"]
            out["specular_color"][interior_mask] = res["specular_rgb"]
            out["diffuse_albedo"][interior_mask] = params["diffuse_albedo"]
            out["specular_albedo"][interior_mask] = params["specular_albedo"]
            out["specular_roughness"][interior_mask] = params["specular_roughness"].squeeze(-1)
            out["normal"][interior_mask] = normals
        return out

    return render_fn


def fixture_camera(width, height, dtype=torch.float32):
    cam = json.load(open(os.path.join(REF, "tests/data_singleview/cam_dict_norm.json")))["12.png"]
    K = torch.tensor(cam["K"], dtype=torch.float32).reshape(4, 4)
    W2C = torch.tensor(cam["W2C"], dtype=torch.float32).reshape(4, 4)
    K[0, :3] *= width / 512
    K[1, :3] *= height / 512
    return Camera(width, height, K.to(dtype), W2C.to(dtype))


class Camera64(Camera):
    """fp64 run: the reference's get_uv always makes fp32 pixel centres (raytracer.py:300-303)."""

    def get_uv(self):
        return super().get_uv().double()


def npf(t):
    return t.detach().cpu().numpy()


def main():
    meta = {"torch": torch.__version__, "numpy": np.__version__, "light": LIGHT}

    # ---- G0: constructor parity (hash of seeded state dicts; the build's own constructors must reproduce it)
    for scene in ("S0", "S1"):
        meta["state_sha256_" + scene] = state_hash(build_reference_networks(scene))

    # ---- G1: positional encoding
    g = torch.Generator().manual_seed(11)
    x = torch.rand(64, 3, generator=g) * 2 - 1
    g1 = {"x": npf(x)}
    for L in (4, 6, 10):
        fn, dim = get_embedder(L)
        g1["pe%d" % L] = npf(fn(x))
        assert g1["pe%d" % L].shape[1] == dim
    np.savez_compressed(os.path.join(HERE, "g1_pe.npz"), **g1)

    nets = build_reference_networks("S1")
    sdf_net = nets["sdf_network"]

    # ---- G2: SDFNetwork.forward + get_all
    g = torch.Generator().manual_seed(12)
    x = torch.rand(2048, 3, generator=g) * 2 - 1
    with torch.no_grad():
        full = sdf_net(x)
    y, feat, grad = sdf_net.get_all(x.clone(), is_training=False)
    np.savez_compressed(os.path.join(HERE, "g2_sdf.npz"), x=npf(x), sdf=npf(full[:, 0]), feature256=npf(full[:256, 1:]),
                        getall_sdf=npf(y[:, 0]), getall_grad=npf(grad), getall_feature256=npf(feat[:256]))

    # ---- G3: material nets + get_materials on surface-ish inputs
    g = torch.Generator().manual_seed(13)
    pts = torch.nn.functional.normalize(torch.randn(256, 3, generator=g), dim=-1) * 0.5
    _, feat, grad = sdf_net.get_all(pts.clone(), is_training=False)
    nrm = grad / (grad.norm(dim=-1, keepdim=True) + 1e-10)
    with torch.no_grad():
        raw_d = nets["diffuse_albedo_network"](pts, nrm, -nrm, feat)
        raw_s = nets["specular_albedo_network"](pts, nrm, None, feat)
        raw_r = nets["specular_roughness_network"](pts, nrm, None, feat)
        mats = get_materials(nets, pts, nrm, feat)
    np.savez_compressed(os.path.join(HERE, "g3_materials.npz"), points=npf(pts), normals=npf(nrm), features=npf(feat),
                        raw_diffuse=npf(raw_d), raw_specular=npf(raw_s), raw_roughness=npf(raw_r),
                        diffuse_albedo=npf(mats["diffuse_albedo"]), specular_albedo=npf(mats["specular_albedo"]),
                        specular_roughness=npf(mats["specular_roughness"]))

    # ---- G4: GGX on a (dot, alpha) grid incl. clamp edges
    renderer = GGXColocatedRenderer(use_cuda=False)
    dots = torch.cat([torch.linspace(-0.2, 1.0, 61), torch.tensor([1e-6, 1e-5, 0.5, 0.99999, 0.999995, 1.0])])
    alphas = torch.cat([torch.logspace(-5, 0.7, 40), torch.tensor([1e-4, 0.01, 0.11, 0.5, 4.0, 5.0])])
    dd, aa = torch.meshgrid(dots, alphas, indexing="ij")
    dd, aa = dd.reshape(-1, 1), aa.reshape(-1, 1)
    n = dd.shape[0]
    normal = torch.tensor([0.0, 0.0, 1.0]).expand(n, 3).contiguous()
    sin = torch.sqrt(torch.clamp(1 - dd * dd, min=0))
    view = torch.cat([sin, torch.zeros_like(sin), dd], dim=-1)
    g = torch.Generator().manual_seed(14)
    kd = torch.rand(n, 3, generator=g)
    ks = torch.rand(n, 3, generator=g)
    dist = 1.0 + torch.rand(n, 1, generator=g) * 2
    with torch.no_grad():
        res = renderer(torch.tensor(LIGHT), dist, normal, view,
                       params={"diffuse_albedo": kd, "specular_albedo": ks, "specular_roughness": aa})
    np.savez_compressed(os.path.join(HERE, "g4_ggx.npz"), light=np.float32(LIGHT), distance=npf(dist), normal=npf(normal),
                        viewdir=npf(view), diffuse_albedo=npf(kd), specular_albedo=npf(ks), specular_roughness=npf(aa),
                        diffuse_rgb=npf(res["diffuse_rgb"]), specular_rgb=npf(res["specular_rgb"]), rgb=npf(res["rgb"]))

    # ---- G5: camera rays + sphere intersection (fixture camera 512^2 subsampled, and the C0 crop)
    cam512 = fixture_camera(512, 512)
    uv = cam512.get_uv()[::8, ::8].contiguous()
    ro, rd, rn = cam512.get_rays(uv)
    m, near, far = intersect_sphere(ro.reshape(-1, 3), rd.reshape(-1, 3), r=1.0)
    crop, _, _ = cam512.crop_region(64, 64, ul_corner=(224, 224))
    uvc = crop.get_uv()
    roc, rdc, rnc = crop.get_rays(uvc)
    np.savez_compressed(os.path.join(HERE, "g5_rays.npz"), K=npf(cam512.K), W2C=npf(cam512.W2C), uv=npf(uv), ray_o=npf(ro),
                        ray_d=npf(rd), ray_d_norm=npf(rn), mask=npf(m), near=npf(near), far=npf(far),
                        crop_K=npf(crop.K), crop_uv=npf(uvc), crop_ray_d=npf(rdc), crop_ray_d_norm=npf(rnc))

    # ---- G6 / G7: tracer + full render, scenes S0 and S1, C0 crop (64x64) and a 128x128 full view
    tracer = RayTracer()
    for scene in ("S0", "S1"):
        nets = build_reference_networks(scene)
        sdf_net = nets["sdf_network"]
        counter = {"evals": 0}
        fwd = sdf_net.forward

        def counting_forward(x, _f=fwd, _c=counter):
            _c["evals"] += int(x.shape[0])
            return _f(x)

        for tag, cam in (("c0", fixture_camera(512, 512).crop_region(64, 64, ul_corner=(224, 224))[0]),
                         ("v128", fixture_camera(128, 128))):
            sdf_net.forward = counting_forward
            counter["evals"] = 0
            tr = raytrace_camera(cam, sdf_net, tracer, max_num_rays=50000, fill_holes=False, detect_edges=False)
            trace_evals = counter["evals"]
            sdf_net.forward = fwd
            res = render_camera(cam, sdf_net, tracer, nets, make_render_fn(nets, renderer, torch.float32),
                                fill_holes=False, handle_edges=False, is_training=False)
            out = {k: npf(v) for k, v in res.items() if isinstance(v, torch.Tensor)}
            out["trace_evals"] = np.int64(trace_evals)
            out["n_conv"] = np.int64(int(res["convergent_mask"].sum()))
            assert np.array_equal(out["convergent_mask"], npf(tr["convergent_mask"]))
            # fp64 run of the same reference code: the reference's own fp32 noise floor
            nets64 = {k: v.double() for k, v in build_reference_networks(scene).items()}
            r64 = GGXColocatedRenderer(use_cuda=False)
            cam64 = Camera64(cam.W, cam.H, cam.K.double(), cam.W2C.double())
            res64 = render_camera(cam64, nets64["sdf_network"], tracer, nets64, make_render_fn(nets64, r64, torch.float64),
                                  fill_holes=False, handle_edges=False, is_training=False)
            out["color_fp64"] = npf(res64["color"]).astype(np.float64)
            out["mask_fp64"] = npf(res64["convergent_mask"])
            out["distance_fp64"] = npf(res64["distance"]).astype(np.float64)
            out["normal_fp64"] = npf(res64["normal"]).astype(np.float64)
            for k in ("uv", "ray_o"):  # reproducible from the camera; keep fixtures small
                out.pop(k, None)
            np.savez_compressed(os.path.join(HERE, "g67_%s_%s.npz" % (scene, tag)), K=npf(cam.K), W2C=npf(cam.W2C),
                                W=np.int64(cam.W), H=np.int64(cam.H), **out)
            c32, c64 = out["color"].astype(np.float64), out["color_fp64"]
            meta["ref32_vs_ref64_rel_l2_%s_%s" % (scene, tag)] = float(
                np.linalg.norm(c32 - c64) / max(np.linalg.norm(c64), 1e-30))
            meta["mask_flips_32_64_%s_%s" % (scene, tag)] = int((out["convergent_mask"] != out["mask_fp64"]).sum())
            meta["trace_evals_%s_%s" % (scene, tag)] = int(trace_evals)
            meta["n_conv_%s_%s" % (scene, tag)] = int(out["n_conv"])
            print(scene, tag, "evals", trace_evals, "conv", int(out["n_conv"]),
                  "ref32~ref64 relL2", meta["ref32_vs_ref64_rel_l2_%s_%s" % (scene, tag)])

    # ---- G8: silhouette handling.  kornia (closing / sobel) is absent, so raytrace_camera's
    # fill_holes / detect_edges branches cannot run in the reference; what CAN be pinned is everything behind the
    # sobel mask: locate_edge_points + render_edge_pixels are called from the reference with a depth-edge mask that
    # the oracle's own (unpinned) sobel restatement produced and that is stored as an INPUT of the fixture.
    from oracle import iron_ref as R  # the build's restatement: only for the unpinned sobel mask
    for scene in ("S0", "S1"):
        nets = build_reference_networks(scene)
        sdf_net = nets["sdf_network"]
        cam = fixture_camera(96, 96)
        res = raytrace_camera(cam, sdf_net, tracer, max_num_rays=50000, fill_holes=False, detect_edges=False)
        depth_edge_mask = (R.sobel_magnitude(res["depth"]) > 1e-2) & res["convergent_mask"]
        res.update(locate_edge_points(cam, res["points"], sdf_net, max_step=16, step_size=1e-3, dot_threshold=5e-2,
                                      max_num_rays=50000, mask=depth_edge_mask))
        res["convergent_mask"] &= ~res["edge_mask"]
        fn = make_render_fn(nets, renderer, torch.float32)
        render_normal_and_color(res, sdf_net, nets, fn, is_training=False, max_num_pts=320000)
        before = {k: npf(res[k]).copy() for k in ("color", "normal", "points", "uv")}
        # The reference's render_normal_and_color builds EMPTY tensors with `.cuda()` when a chunk has no hit
        # (raytracer.py:627-633) -- the usual case for the outer side rays.  There is no GPU in this container, so for
        # this one call Tensor.cuda is made the identity (the tensors are empty; nothing of the computation changes).
        _cuda = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self
        try:
            render_edge_pixels(res, cam, sdf_net, tracer, nets, fn, is_training=False)
        finally:
            torch.Tensor.cuda = _cuda
        out = {k: npf(v) for k, v in res.items() if isinstance(v, torch.Tensor) and k not in ("ray_o",)}
        out["depth_edge_mask_input"] = npf(depth_edge_mask)
        for k, v in before.items():
            out["pre_edge_" + k] = v
        np.savez_compressed(os.path.join(HERE, "g8_edges_%s.npz" % scene), K=npf(cam.K), W2C=npf(cam.W2C), W=np.int64(cam.W),
                            H=np.int64(cam.H), **out)
        meta["n_edge_pixels_%s" % scene] = int(res["edge_mask"].sum())
        print(scene, "edge pixels", int(res["edge_mask"].sum()), "of sobel candidates", int(depth_edge_mask.sum()))

    json.dump(meta, open(os.path.join(HERE, "meta.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(meta, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
