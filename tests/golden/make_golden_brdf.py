"""Golden vectors for SURVEY 8 row f-4 (the fork's other co-located BRDF heads) -- BUILD CONTAINER ONLY.

Same rules as make_golden.py (whose helpers it reuses): the real reference is imported on CPU, run on seeded inputs,
and only the resulting data is written.  G9 = the pointwise heads of models/renderer_ggx.py (CompositeRenderer.forward
with and without env light, SmoothDielectric / ThinDielectric / SmoothConductorCoLoc / RoughConductorCoLoc);
G10 = get_materials_comp (models/rendering_func.py:19-49) and render_camera with the composite render_fn of
render_surface.py:159-234 on scene S2 (S0's SDF + the `comp2` material networks of models/network_conf.py:318-447);
G11 = the forward of a points_only RenderingNetwork (comp2's env_light_network).

RoughPlasticCoLocRenderer / CoLocRenderer are NOT recorded: the reference's RoughPlasticCoLocRenderer.forward passes a
Python float as `eta` to fresnel_dielectric, which indexes it (renderer_ggx.py:404,485) -> TypeError; there is no
output to pin.

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_brdf.py
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (placeholder modules + reference imports)

from models.fields import RenderingNetwork, SDFNetwork  # noqa: E402  (reference)
from models.raytracer import RayTracer, render_camera  # noqa: E402
from models.renderer_ggx import (CompositeRenderer, RoughConductorCoLocRenderer, RoughPlasticCoLocRenderer,  # noqa: E402
                                 SmoothConductorCoLocRenderer, SmoothDielectricRenderer, ThinDielectricRenderer)
from models.rendering_func import get_materials_comp  # noqa: E402

LIGHT = MG.LIGHT
npf = MG.npf

COMP_ORDER = ("diffuse_albedo_network", "specular_albedo_network", "specular_roughness_network", "metallic_network",
              "dielectric_network", "metallic_eta_network", "metallic_k_network", "dielectric_eta_network")


def build_comp_networks(seed: int = 0):
    """Scene S2: seed-0 SDF network, then the comp2 material networks in COMP_ORDER (fixes the RNG stream)."""
    torch.manual_seed(seed)
    nets = {"sdf_network": SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5,
                                      scale=1.0, geometric_init=True, weight_norm=True)}

    def ns(d_out, bias):
        return RenderingNetwork(d_in=6, d_out=d_out, d_feature=256, d_hidden=256, n_layers=4, multires=6, multires_view=-1,
                                mode="no_view_dir", squeeze_out=False, output_bias=bias, output_scale=1.0)

    for name in COMP_ORDER:
        if name == "diffuse_albedo_network":
            nets[name] = RenderingNetwork(d_in=9, d_out=3, d_feature=256, d_hidden=256, n_layers=4, multires_view=4,
                                          mode="idr", squeeze_out=True)
        elif name == "specular_albedo_network":
            nets[name] = ns(3, 0.0)
        else:
            nets[name] = ns(1, 0.1)
    return nets


def make_render_fn_comp(renderer):
    """The logic of render_surface.py:159-234."""

    def render_fn(interior_mask, color_network_dict, ray_o, ray_d, points, normals, features):
        sh = list(interior_mask.shape)
        rgb = torch.zeros(sh + [3], dtype=torch.float32)
        out = {k: rgb.clone() for k in ("color", "diffuse_color", "specular_color", "diffuse_albedo", "specular_albedo",
                                        "metallic_rgb", "dielectric_rgb", "normal")}
        for k in ("specular_roughness", "metallic_eta", "metallic_k", "dielectric_eta", "metallic", "dielectric"):
            out[k] = rgb[..., 0:1].clone()
        if interior_mask.any():
            normals = normals / (normals.norm(dim=-1, keepdim=True) + 1e-10)
            params = get_materials_comp(network_dict=color_network_dict, points=points, normals=normals, features=features)
            res = renderer(torch.tensor(LIGHT), (points - ray_o).norm(dim=-1, keepdim=True), normals, -ray_d, params=params)
            out["color"][interior_mask] = res["rgb"]
            out["diffuse_color"][interior_mask] = res["diffuse_rgb"]
            out["specular_color"][interior_mask] = res["specular_rgb"]
            out["metallic_rgb"][interior_mask] = res["metallic_rgb"]
            out["dielectric_rgb"][interior_mask] = res["dielectric_rgb"]
            for k in ("diffuse_albedo", "specular_albedo", "specular_roughness", "metallic_eta", "metallic_k", "dielectric_eta",
                      "metallic", "dielectric"):
                out[k][interior_mask] = params[k]
            out["normal"][interior_mask] = normals
        return out

    return render_fn


def main():
    meta_path = os.path.join(HERE, "meta.json")
    meta = json.load(open(meta_path))

    # ---- G9: pointwise heads on a (dot, roughness) grid incl. clamp edges, random material parameters
    dots = torch.cat([torch.linspace(-0.2, 1.0, 41), torch.tensor([1e-6, 1e-5, 0.5, 0.99999, 0.999995, 1.0])])
    alphas = torch.cat([torch.logspace(-6, 0.7, 24), torch.tensor([1e-5, 1e-4, 0.01, 0.11, 4.0, 5.0])])
    dd, aa = torch.meshgrid(dots, alphas, indexing="ij")
    dd, aa = dd.reshape(-1, 1), aa.reshape(-1, 1)
    n = dd.shape[0]
    normal = torch.tensor([0.0, 0.0, 1.0]).expand(n, 3).contiguous()
    sin = torch.sqrt(torch.clamp(1 - dd * dd, min=0))
    view = torch.cat([sin, torch.zeros_like(sin), dd], dim=-1)
    g = torch.Generator().manual_seed(19)
    rnd = lambda *s: torch.rand(*s, generator=g)  # noqa: E731
    inp = {"distance": 1.0 + rnd(n, 1) * 2, "normal": normal, "viewdir": view,
           "diffuse_albedo": rnd(n, 3) * 1.2 - 0.05, "specular_albedo": rnd(n, 3) * 1.2 - 0.05, "specular_roughness": aa,
           "metallic": rnd(n, 1) * 1.2 - 0.1, "dielectric": rnd(n, 1) * 1.2 - 0.1,
           "metallic_eta": rnd(n, 1) * 6.0, "metallic_k": rnd(n, 1) * 11.0, "dielectric_eta": 0.9 + rnd(n, 1) * 1.3,
           "env_light": rnd(n, 1) * 25.0 - 1.0}
    out = {k: npf(v) for k, v in inp.items()}
    out["light"] = np.float32(LIGHT)
    params = {k: inp[k] for k in ("diffuse_albedo", "specular_albedo", "specular_roughness", "metallic", "dielectric",
                                  "metallic_eta", "metallic_k", "dielectric_eta", "env_light")}
    comp = CompositeRenderer(use_cuda=False)
    with torch.no_grad():
        for tag, env in (("comp", False), ("compenv", True)):
            res = comp(torch.tensor(LIGHT), inp["distance"], normal, view, params={k: v.clone() for k, v in params.items()},
                       use_env_light=env)
            for k, v in res.items():
                out["%s_%s" % (tag, k)] = npf(v)
        heads = {"smooth_dielectric": SmoothDielectricRenderer(use_cuda=False), "thin_dielectric": ThinDielectricRenderer(use_cuda=False),
                 "smooth_conductor": SmoothConductorCoLocRenderer(ior_path="./resource/ior", use_cuda=False),
                 "rough_conductor": RoughConductorCoLocRenderer(ior_path="./resource/ior", use_cuda=False)}
        for tag, r in heads.items():
            res = r(torch.tensor(LIGHT), inp["distance"], normal, view, inp["diffuse_albedo"], inp["specular_albedo"], aa)
            for k, v in res.items():
                out["%s_%s" % (tag, k)] = npf(v)
        try:
            RoughPlasticCoLocRenderer(use_cuda=False)(torch.tensor(LIGHT), inp["distance"], normal, view, inp["diffuse_albedo"],
                                                      inp["specular_albedo"], aa)
            meta["rough_plastic_reference_error"] = None
        except Exception as e:  # noqa: BLE001  (recorded: the reference head has no output to pin)
            meta["rough_plastic_reference_error"] = type(e).__name__
    np.savez_compressed(os.path.join(HERE, "g9_brdf_heads.npz"), **out)

    # ---- G10: comp2 material networks (constructor parity hash), get_materials_comp, composite render_camera (C0 crop)
    nets = build_comp_networks()
    meta["state_sha256_S2"] = MG.state_hash(nets)
    sdf_net = nets["sdf_network"]
    g = torch.Generator().manual_seed(13)
    pts = torch.nn.functional.normalize(torch.randn(256, 3, generator=g), dim=-1) * 0.5
    _, feat, grad = sdf_net.get_all(pts.clone(), is_training=False)
    nrm = grad / (grad.norm(dim=-1, keepdim=True) + 1e-10)
    with torch.no_grad():
        mats = get_materials_comp(nets, pts, nrm, feat)
    np.savez_compressed(os.path.join(HERE, "g10_comp_materials.npz"), points=npf(pts), normals=npf(nrm), features=npf(feat),
                        **{k: npf(v) for k, v in mats.items()})
    cam512 = MG.fixture_camera(512, 512)
    cam, _, _ = cam512.crop_region(64, 64, ul_corner=(224, 224))
    with torch.no_grad():
        res = render_camera(cam, sdf_net, RayTracer(), nets, make_render_fn_comp(comp), fill_holes=False, handle_edges=False,
                            is_training=False)
    np.savez_compressed(os.path.join(HERE, "g10_comp_S2_c0.npz"), K=npf(cam.K), W2C=npf(cam.W2C), W=np.int64(cam.W), H=np.int64(cam.H),
                        **{k: npf(v) for k, v in res.items()})
    meta["n_conv_S2_c0"] = int(res["convergent_mask"].sum())

    # ---- G11: a points_only material head (comp2's env_light_network, models/network_conf.py:367-378), seed 7
    torch.manual_seed(7)
    env = RenderingNetwork(d_in=3, d_out=1, d_feature=256, d_hidden=256, n_layers=4, multires=6, multires_view=-1,
                           mode="points_only", squeeze_out=False, output_bias=0.0, output_scale=1.0)
    with torch.no_grad():
        env_out = env(pts, None, None, feat)
    meta["state_sha256_env_light"] = MG.state_hash({"env_light_network": env})
    np.savez_compressed(os.path.join(HERE, "g11_points_only.npz"), points=npf(pts), features=npf(feat), out=npf(env_out))
    json.dump(meta, open(meta_path, "w"), indent=1, sort_keys=True)
    print({k: meta[k] for k in ("state_sha256_S2", "state_sha256_env_light", "rough_plastic_reference_error", "n_conv_S2_c0")})


if __name__ == "__main__":
    main()
