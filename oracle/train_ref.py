"""CPU restatement of the reference's stage-2 TRAINING forward (BASELINE config C3) -- TEST
INFRASTRUCTURE ONLY (same import rule as oracle/iron_ref.py).  No product code exists for this row yet: the restatement and
its goldens (tests/golden/make_golden_train.py; tests/test_oracle_train.py) fix what the HIP backward will be held to.

What differs from the inference path (oracle/iron_ref.py), line for line:
    models/raytracer.py:17-24     reparam_points: the (non-differentiable) hit point is made a function of the SDF
                                  parameters through x - d / max(n.d, 1e-4) * (sdf(x) - sdf(x).detach())
    models/fields.py:120-137      get_all(is_training=True): autograd.grad(..., create_graph=True) -> the normal carries
                                  second-order dependence on the SDF parameters
    models/raytracer.py:622-646   render_normal_and_color(is_training=True): render_fn under grad mode
Gradients are then whatever torch.autograd makes of this graph; the parameters are passed as leaf tensors that require
grad (a state_dict of such leaves), every op below is differentiable torch code.
"""
from __future__ import annotations

from typing import Dict

import torch
from torch import Tensor

from . import iron_ref as R


def leaf_state(sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """state_dict -> same keys, leaf tensors with requires_grad (floating-point entries only)."""
    return {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() else v.clone()) for k, v in sd.items()}


def reparam_points(points: Tensor, grads: Tensor, dirs: Tensor, sdf_vals: Tensor) -> Tensor:
    """raytracer.py:17-24."""
    dot = torch.clamp((grads * dirs).sum(dim=-1, keepdim=True), min=1e-4)
    return points - dirs / dot * (sdf_vals - sdf_vals.detach())


def sdf_get_all_train(sd: Dict[str, Tensor], spec: R.SDFSpec, x: Tensor):
    """fields.py:120-137 with is_training=True: (sdf, feature, gradient), all attached to the graph."""
    with torch.enable_grad():
        xg = x.detach().clone().requires_grad_(True)
        out = R.sdf_forward(sd, spec, xg)
        y, feat = out[..., :1], out[..., 1:]
        (grad,) = torch.autograd.grad(y, xg, torch.ones_like(y), create_graph=True, retain_graph=True, only_inputs=True)
    return y, feat, grad


def render_normal_and_color_train(scene: R.Scene, res: Dict[str, Tensor]) -> None:
    """render_normal_and_color(is_training=True), raytracer.py:593-662 (one chunk: the goldens are far below 320 000 points);
    mutates `res`."""
    sh = list(res["convergent_mask"].shape)
    m = res["convergent_mask"].reshape(-1)
    if bool(m.any()):
        p = res["points"].reshape(-1, 3)[m]
        d = res["ray_d"].reshape(-1, 3)[m]
        o = res["ray_o"].reshape(-1, 3)[m]
        sdf, feat, grad = sdf_get_all_train(scene.sdf_sd, scene.sdf_spec, p)
        p = reparam_points(p, grad.detach(), -d.detach(), sdf)
    else:
        p = d = o = grad = feat = torch.zeros(0, dtype=torch.float32)
    with torch.enable_grad():
        if scene.renderer == "comp":  # render_fn_comp, render_surface.py:159-234
            r = R.render_fn_comp(scene.nets, scene.light, scene.mts_trans, scene.mts_diff_trans, m, o, d, p, grad, feat)
        else:
            r = R.render_fn_ggx(scene, m, o, d, p, grad, feat)
    for k, v in r.items():
        v = v.reshape(sh + [-1])
        res[k] = v.squeeze(-1) if v.shape[-1] == 1 else v


def render_edge_pixels_train(scene: R.Scene, results: Dict[str, Tensor], cam: R.CameraSpec, prm: R.TracerParams = R.TracerParams()) -> None:
    """raytracer.py:665-729 with is_training=True: the edge point is re-parametrised along its normal, re-projected, and the
    blend weight of the two side colours becomes a function of the SDF parameters; the side rays are shaded under grad."""
    edge_points, edge_idx = results["edge_points"], results["edge_pixel_idx"]
    center = torch.floor(results["edge_uv"]) + 0.5
    edge_sdf, _, grads = sdf_get_all_train(scene.sdf_sd, scene.sdf_spec, edge_points)
    nrm = grads.detach() / (grads.detach().norm(dim=-1, keepdim=True) + 1e-10)
    edge_points = reparam_points(edge_points, grads.detach(), nrm, edge_sdf)
    edge_uv = R.project(cam, edge_points)
    n2d = torch.matmul(nrm, cam.W2C[:3, :3].transpose(1, 0))[:, :2]
    n2d = n2d / (n2d.norm(dim=-1, keepdim=True) + 1e-10)
    radius = 0.707
    pos_uv, neg_uv = center - radius * n2d, center + radius * n2d
    dot2d = torch.sum((edge_uv - center) * n2d, dim=-1)
    alpha = 2 * torch.arccos(torch.clamp(dot2d / radius, min=0.0, max=1.0))
    w_pos = 1.0 - (alpha - torch.sin(alpha)) / (2.0 * torch.pi)
    with torch.no_grad():
        pos = R.raytrace_pixels(scene, pos_uv, cam, prm=prm)
        neg = R.raytrace_pixels(scene, neg_uv, cam, prm=prm)
    render_normal_and_color_train(scene, pos)
    render_normal_and_color_train(scene, neg)
    color = pos["color"] * w_pos.unsqueeze(-1) + neg["color"] * (1.0 - w_pos.unsqueeze(-1))
    results["color"].view(-1, 3)[edge_idx] = color
    results["normal"].view(-1, 3)[edge_idx] = grads
    results["uv"].view(-1, 2)[edge_idx] = edge_uv.detach()
    results["points"].view(-1, 3)[edge_idx] = edge_points.detach()


def render_camera_edges_train(scene: R.Scene, cam: R.CameraSpec, depth_edge_mask: Tensor) -> Dict[str, Tensor]:
    """render_camera(fill_holes=False, handle_edges=True, is_training=True), raytracer.py:778-814, behind a given depth-edge
    mask (the sobel stencil is the unpinned part, see oracle/iron_ref.py)."""
    with torch.no_grad():
        res = R.raytrace_camera_full(scene, cam, max_num_rays=50000, fill_holes=False, detect_edges=True, depth_edge_mask=depth_edge_mask)
    render_normal_and_color_train(scene, res)
    if res["edge_mask"].sum() > 0:
        render_edge_pixels_train(scene, res, cam)
    return res


def render_camera_train(scene: R.Scene, cam: R.CameraSpec) -> Dict[str, Tensor]:
    """render_camera(fill_holes=False, handle_edges=False, is_training=True), raytracer.py:778-814: the trace runs without
    grad (raytracer.py:542 @torch.no_grad), shading with it.  `scene`'s state dicts must be leaf_state()s."""
    with torch.no_grad():
        res = R.raytrace_camera(scene, cam, max_num_rays=50000)
    assert bool(res["convergent_mask"].any()), "the golden crop has hits"
    render_normal_and_color_train(scene, res)
    return res
