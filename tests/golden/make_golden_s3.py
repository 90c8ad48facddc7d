"""Golden vectors for scene S3 ("trained-like" dynamic range, iron_amd/scenes.py) -- BUILD CONTAINER ONLY.

The reference's classes are built like S1 and then put through iron_amd.scenes.rescale_hidden_units (the SAME function the
product's scene builder uses, so both sides hold bit-identical parameters: state hash in meta.json); the reference's own
forward / get_all / material networks / render_camera are then run in fp32 AND in fp64.  The split-fp16 MLP core is held to
max(floor, 1.5 x ref32-vs-ref64) stage by stage and end to end (tests/test_gpu_s3.py): the stand-in for a trained checkpoint,
which does not exist offline.

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_s3.py
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

from models.raytracer import RayTracer, render_camera  # noqa: E402  (reference)
from models.renderer_ggx import GGXColocatedRenderer  # noqa: E402
from models.rendering_func import get_materials  # noqa: E402
from iron_amd.scenes import rescale_hidden_units  # noqa: E402  (parameter surgery only; no product compute)

npf = MG.npf


def build(dtype=torch.float32):
    nets = MG.build_reference_networks("S1")
    rescale_hidden_units(nets, seed=3)
    if dtype == torch.float64:
        nets = {k: v.double() for k, v in nets.items()}
    return nets


def main():
    n32, n64 = build(), build(torch.float64)
    out = {}
    meta_path = os.path.join(HERE, "meta.json")
    meta = json.load(open(meta_path))
    meta["state_sha256_S3"] = MG.state_hash(n32)
    # ---- stage 1: SDF forward + get_all
    g = torch.Generator().manual_seed(21)
    x = torch.rand(2048, 3, generator=g) * 1.4 - 0.7
    with torch.no_grad():
        f32, f64 = n32["sdf_network"](x), n64["sdf_network"](x.double())
    _, _, g32 = n32["sdf_network"].get_all(x.clone(), is_training=False)
    _, _, g64 = n64["sdf_network"].get_all(x.double().clone(), is_training=False)
    out.update(x=npf(x), sdf=npf(f32[:, 0]), sdf_fp64=npf(f64[:, 0]), grad=npf(g32), grad_fp64=npf(g64), feature256=npf(f32[:256, 1:]),
               feature256_fp64=npf(f64[:256, 1:]))
    # hidden activation statistics of the fp64 run (what "trained-like" means here)
    acts = []
    hooks = [getattr(n64["sdf_network"], "lin%d" % l).register_forward_hook(lambda m, i, o: acts.append(float(o.abs().max()))) for l in range(8)]
    with torch.no_grad():
        n64["sdf_network"](x.double())
    for h in hooks:
        h.remove()
    meta["S3_sdf_hidden_preactivation_max"] = acts
    meta["S3_weight_g_range"] = [float(min(p.min() for n, p in n32["sdf_network"].named_parameters() if n.endswith("weight_g"))),
                                 float(max(p.max() for n, p in n32["sdf_network"].named_parameters() if n.endswith("weight_g")))]
    # ---- stage 2: material networks on points near the surface, identical (fp32) inputs for both precisions
    g = torch.Generator().manual_seed(22)
    pts = torch.nn.functional.normalize(torch.randn(256, 3, generator=g), dim=-1) * 0.5
    _, feat, grad = n32["sdf_network"].get_all(pts.clone(), is_training=False)
    nrm = grad / (grad.norm(dim=-1, keepdim=True) + 1e-10)
    with torch.no_grad():
        m32 = get_materials(n32, pts, nrm, feat)
        m64 = get_materials(n64, pts.double(), nrm.double(), feat.double())
    out.update(m_points=npf(pts), m_normals=npf(nrm), m_features=npf(feat))
    for k in ("diffuse_albedo", "specular_albedo", "specular_roughness"):
        out["m_" + k], out["m_" + k + "_fp64"] = npf(m32[k]), npf(m64[k])
    # ---- end to end: 128x128 view
    renderer = GGXColocatedRenderer(use_cuda=False)
    cam = MG.fixture_camera(128, 128)
    with torch.no_grad():
        r32 = render_camera(cam, n32["sdf_network"], RayTracer(), n32, MG.make_render_fn(n32, renderer, torch.float32), fill_holes=False,
                            handle_edges=False, is_training=False)
        cam64 = MG.Camera64(cam.W, cam.H, cam.K.double(), cam.W2C.double())
        r64 = render_camera(cam64, n64["sdf_network"], RayTracer(), n64, MG.make_render_fn(n64, GGXColocatedRenderer(use_cuda=False), torch.float64),
                            fill_holes=False, handle_edges=False, is_training=False)
    for k in ("convergent_mask", "color", "distance", "normal"):
        out["r_" + k], out["r_" + k + "_fp64"] = npf(r32[k]), npf(r64[k])
    out.update(K=npf(cam.K), W2C=npf(cam.W2C))
    np.savez_compressed(os.path.join(HERE, "g18_S3_trained_like.npz"), **out)
    both = out["r_convergent_mask"] & out["r_convergent_mask_fp64"]
    rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
    meta["S3_mask_flips_32_64_v128"] = int((out["r_convergent_mask"] != out["r_convergent_mask_fp64"]).sum())
    meta["S3_colour_rel_l2_ref32_ref64_v128"] = rel(out["r_color"][both], out["r_color_fp64"][both])
    meta["S3_sdf_rel_l2_ref32_ref64"] = rel(out["sdf"], out["sdf_fp64"])
    meta["S3_grad_rel_l2_ref32_ref64"] = rel(out["grad"], out["grad_fp64"])
    json.dump(meta, open(meta_path, "w"), indent=1, sort_keys=True)
    print({k: v for k, v in meta.items() if k.startswith("S3") or k.endswith("_S3")})


if __name__ == "__main__":
    main()
