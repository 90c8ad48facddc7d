"""CPU restatement of the reference's STAGE-1 volume renderer (BASELINE config C2) -- TEST
INFRASTRUCTURE ONLY, like oracle/iron_ref.py (same import rule).  No product code exists for this row yet: the
restatement and its goldens (tests/golden/make_golden_neus.py, pinned by tests/test_oracle_neus.py) are the target the
HIP build of the row will be held to.

Follows, op for op:
    models/fields.py:243-327   NeRF.forward (background field; use_viewdirs=True)
    models/fields.py:415-421   SingleVarianceNetwork.forward
    models/renderer.py:45-75   sample_pdf
    models/renderer.py:151-187 NeuSRenderer.render_core_outside
    models/renderer.py:189-232 NeuSRenderer.up_sample
    models/renderer.py:234-248 NeuSRenderer.cat_z_vals
    models/renderer.py:250-344 NeuSRenderer.render_core
    models/renderer.py:346-453 NeuSRenderer.render (perturb = 0: the reference draws torch.rand inside otherwise)
The SDF / colour networks are the ones of oracle/iron_ref.py (sdf_forward, sdf_get_all, rendering_forward with the
8-layer PE-10 skip-4 spec of confs/womask_iron.conf).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from . import iron_ref as R

# confs/womask_iron.conf: model.rendering_network / model.nerf / model.neus_renderer
COLOR_SPEC = R.RenderSpec(d_feature=256, mode="idr", d_in=9, d_out=3, d_hidden=256, n_layers=8, multires=10, multires_view=4,
                          squeeze_out=True, skip_in=(4,))


@dataclass
class NerfSpec:
    D: int = 8
    W: int = 256
    d_in: int = 4
    d_in_view: int = 3
    multires: int = 10
    multires_view: int = 4
    skips: Tuple[int, ...] = (4,)


def nerf_forward(sd: Dict[str, Tensor], spec: NerfSpec, pts: Tensor, views: Tensor) -> Tuple[Tensor, Tensor]:
    """fields.py:299-327 (use_viewdirs=True) -> (alpha [n,1], rgb [n,3]); plain nn.Linear layers, no weight norm."""
    x = R.positional_encoding(pts, spec.multires) if spec.multires > 0 else pts
    v = R.positional_encoding(views, spec.multires_view) if spec.multires_view > 0 else views
    h = x
    for i in range(spec.D):
        h = torch.relu(F.linear(h, sd["pts_linears.%d.weight" % i], sd["pts_linears.%d.bias" % i]))
        if i in spec.skips:
            h = torch.cat([x, h], -1)
    alpha = F.linear(h, sd["alpha_linear.weight"], sd["alpha_linear.bias"])
    feat = F.linear(h, sd["feature_linear.weight"], sd["feature_linear.bias"])
    h = torch.cat([feat, v], -1)
    h = torch.relu(F.linear(h, sd["views_linears.0.weight"], sd["views_linears.0.bias"]))
    rgb = F.linear(h, sd["rgb_linear.weight"], sd["rgb_linear.bias"])
    return alpha, rgb


def single_variance(variance: Tensor, n: int) -> Tensor:
    """fields.py:420-421."""
    return torch.ones([n, 1]) * torch.exp(variance * 10.0)


def sample_pdf(bins: Tensor, weights: Tensor, n_samples: int, det: bool = True) -> Tensor:
    """renderer.py:45-75 (det=True is what up_sample uses)."""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    assert det, "the reference's stochastic branch draws torch.rand; not restated"
    u = torch.linspace(0.0 + 0.5 / n_samples, 1.0 - 0.5 / n_samples, steps=n_samples)
    u = u.expand(list(cdf.shape[:-1]) + [n_samples]).contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.max(torch.zeros_like(inds - 1), inds - 1)
    above = torch.min((cdf.shape[-1] - 1) * torch.ones_like(inds), inds)
    inds_g = torch.stack([below, above], -1)
    shape = [inds_g.shape[0], inds_g.shape[1], cdf.shape[-1]]
    cdf_g = torch.gather(cdf.unsqueeze(1).expand(shape), 2, inds_g)
    bins_g = torch.gather(bins.unsqueeze(1).expand(shape), 2, inds_g)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_g[..., 0]) / denom
    return bins_g[..., 0] + t * (bins_g[..., 1] - bins_g[..., 0])


@dataclass
class NeusScene:
    sdf_sd: Dict[str, Tensor]
    sdf_spec: R.SDFSpec
    color_sd: Dict[str, Tensor]
    nerf_sd: Dict[str, Tensor]
    variance: Tensor
    color_spec: R.RenderSpec = COLOR_SPEC
    nerf_spec: NerfSpec = NerfSpec()
    n_samples: int = 64
    n_importance: int = 64
    n_outside: int = 32
    up_sample_steps: int = 4

    def sdf(self, x: Tensor) -> Tensor:
        return R.sdf_forward(self.sdf_sd, self.sdf_spec, x)[:, :1]


def up_sample(rays_o: Tensor, rays_d: Tensor, z_vals: Tensor, sdf: Tensor, n_importance: int, inv_s: float) -> Tensor:
    """renderer.py:189-232."""
    batch, n = z_vals.shape
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z_vals[..., :, None]
    radius = torch.linalg.norm(pts, ord=2, dim=-1, keepdim=False)
    inside = (radius[:, :-1] < 1.0) | (radius[:, 1:] < 1.0)
    sdf = sdf.reshape(batch, n)
    prev_sdf, next_sdf = sdf[:, :-1], sdf[:, 1:]
    prev_z, next_z = z_vals[:, :-1], z_vals[:, 1:]
    mid_sdf = (prev_sdf + next_sdf) * 0.5
    cos_val = (next_sdf - prev_sdf) / (next_z - prev_z + 1e-5)
    prev_cos = torch.cat([torch.zeros([batch, 1]), cos_val[:, :-1]], dim=-1)
    cos_val, _ = torch.min(torch.stack([prev_cos, cos_val], dim=-1), dim=-1, keepdim=False)
    cos_val = cos_val.clip(-1e3, 0.0) * inside
    dist = next_z - prev_z
    prev_esti = mid_sdf - cos_val * dist * 0.5
    next_esti = mid_sdf + cos_val * dist * 0.5
    prev_cdf = torch.sigmoid(prev_esti * inv_s)
    next_cdf = torch.sigmoid(next_esti * inv_s)
    alpha = (prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)
    weights = alpha * torch.cumprod(torch.cat([torch.ones([batch, 1]), 1.0 - alpha + 1e-7], -1), -1)[:, :-1]
    return sample_pdf(z_vals, weights, n_importance, det=True)


def cat_z_vals(sc: NeusScene, rays_o: Tensor, rays_d: Tensor, z_vals: Tensor, new_z: Tensor, sdf: Tensor,
               last: bool) -> Tuple[Tensor, Tensor]:
    """renderer.py:234-248."""
    batch, n = z_vals.shape
    _, n_imp = new_z.shape
    pts = rays_o[:, None, :] + rays_d[:, None, :] * new_z[..., :, None]
    z_vals, index = torch.sort(torch.cat([z_vals, new_z], dim=-1), dim=-1)
    if not last:
        new_sdf = sc.sdf(pts.reshape(-1, 3)).reshape(batch, n_imp)
        sdf = torch.cat([sdf, new_sdf], dim=-1)
        xx = torch.arange(batch)[:, None].expand(batch, n + n_imp).reshape(-1)
        sdf = sdf[(xx, index.reshape(-1))].reshape(batch, n + n_imp)
    return z_vals, sdf


def render_core_outside(sc: NeusScene, rays_o: Tensor, rays_d: Tensor, z_vals: Tensor, sample_dist: float) -> Dict[str, Tensor]:
    """renderer.py:151-187 (background_rgb=None)."""
    batch, n = z_vals.shape
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.Tensor([sample_dist]).expand(dists[..., :1].shape)], -1)
    mid = z_vals + dists * 0.5
    pts = rays_o[:, None, :] + rays_d[:, None, :] * mid[..., :, None]
    dis = torch.linalg.norm(pts, ord=2, dim=-1, keepdim=True).clip(1.0, 1e10)
    pts = torch.cat([pts / dis, 1.0 / dis], dim=-1).reshape(-1, 4)
    dirs = rays_d[:, None, :].expand(batch, n, 3).reshape(-1, 3)
    density, color = nerf_forward(sc.nerf_sd, sc.nerf_spec, pts, dirs)
    alpha = (1.0 - torch.exp(-F.softplus(density.reshape(batch, n)) * dists)).reshape(batch, n)
    weights = alpha * torch.cumprod(torch.cat([torch.ones([batch, 1]), 1.0 - alpha + 1e-7], -1), -1)[:, :-1]
    color = color.reshape(batch, n, 3)
    return {"color": (weights[:, :, None] * color).sum(dim=1), "sampled_color": color, "alpha": alpha, "weights": weights}


def render_core(sc: NeusScene, rays_o: Tensor, rays_d: Tensor, z_vals: Tensor, sample_dist: float,
                background_alpha: Optional[Tensor], background_sampled_color: Optional[Tensor],
                background_rgb: Optional[Tensor], cos_anneal_ratio: float, training: bool = False) -> Dict[str, Tensor]:
    """renderer.py:250-344.  training=True keeps the graph through sdf_network.gradient (create_graph=True, fields.py:106-118)."""
    batch, n = z_vals.shape
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.Tensor([sample_dist]).expand(dists[..., :1].shape)], -1)
    mid = z_vals + dists * 0.5
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * mid[..., :, None]).reshape(-1, 3)
    dirs = rays_d[:, None, :].expand(batch, n, 3).reshape(-1, 3)
    if training:
        from .train_ref import sdf_get_all_train
        sdf, feat, grads = sdf_get_all_train(sc.sdf_sd, sc.sdf_spec, pts)
    else:
        out, feat, grads = R.sdf_get_all(sc.sdf_sd, sc.sdf_spec, pts)
        sdf = out[:, :1]
    color = R.rendering_forward(sc.color_sd, sc.color_spec, pts, grads, dirs, feat).reshape(batch, n, 3)
    inv_s = single_variance(sc.variance, 1)[:, :1].clip(1e-6, 1e6)
    return composite(sdf, grads, color, dists, pts, dirs, inv_s, background_alpha, background_sampled_color, background_rgb, cos_anneal_ratio,
                     mid)


def composite(sdf: Tensor, grads: Tensor, color: Tensor, dists: Tensor, pts: Tensor, dirs: Tensor, inv_s: Tensor,
              background_alpha: Optional[Tensor], background_sampled_color: Optional[Tensor], background_rgb: Optional[Tensor],
              cos_anneal_ratio: float, mid: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """renderer.py:279-344: from the network outputs at the section mid points (sdf [b*n,1], grads [b*n,3], color [b,n,3]) to the
    composited colour, weights and statistics.  Pure tensor math (differentiable): the product's compositing kernel and its
    backward are tested against it."""
    batch, n = dists.shape
    inv_s = inv_s.reshape(1, 1).expand(batch * n, 1)
    true_cos = (dirs * grads).sum(-1, keepdim=True)
    iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal_ratio) + F.relu(-true_cos) * cos_anneal_ratio)
    est_next = sdf + iter_cos * dists.reshape(-1, 1) * 0.5
    est_prev = sdf - iter_cos * dists.reshape(-1, 1) * 0.5
    prev_cdf = torch.sigmoid(est_prev * inv_s)
    next_cdf = torch.sigmoid(est_next * inv_s)
    p, c = prev_cdf - next_cdf, prev_cdf
    alpha = ((p + 1e-5) / (c + 1e-5)).reshape(batch, n).clip(0.0, 1.0)
    pts_norm = torch.linalg.norm(pts, ord=2, dim=-1, keepdim=True).reshape(batch, n)
    inside = (pts_norm < 1.0).float()
    relax = (pts_norm < 1.2).float()
    if background_alpha is not None:
        alpha = alpha * inside + background_alpha[:, :n] * (1.0 - inside)
        alpha = torch.cat([alpha, background_alpha[:, n:]], dim=-1)
        color = color * inside[:, :, None] + background_sampled_color[:, :n] * (1.0 - inside)[:, :, None]
        color = torch.cat([color, background_sampled_color[:, n:]], dim=1)
    weights = alpha * torch.cumprod(torch.cat([torch.ones([batch, 1]), 1.0 - alpha + 1e-7], -1), -1)[:, :-1]
    weights_sum = weights.sum(dim=-1, keepdim=True)
    col = (color * weights[:, :, None]).sum(dim=1)
    if background_rgb is not None:
        col = col + background_rgb * (1.0 - weights_sum)
    g3 = grads.reshape(batch, n, 3)
    gerr = (torch.linalg.norm(g3, ord=2, dim=-1) - 1.0) ** 2
    gerr = (relax * gerr).sum() / (relax.sum() + 1e-5)
    return {"color": col, "sdf": sdf, "dists": dists, "gradients": g3, "s_val": 1.0 / inv_s, "mid_z_vals": mid,
            "weights": weights, "cdf": c.reshape(batch, n), "gradient_error": gerr, "inside_sphere": inside}


@torch.no_grad()
def render(sc: NeusScene, rays_o: Tensor, rays_d: Tensor, near: Tensor, far: Tensor, background_rgb: Optional[Tensor] = None,
           cos_anneal_ratio: float = 0.0) -> Dict[str, Tensor]:
    """renderer.py:346-453 with perturb = 0."""
    return _render(sc, rays_o, rays_d, near, far, background_rgb, cos_anneal_ratio, training=False)


def render_train(sc: NeusScene, rays_o: Tensor, rays_d: Tensor, near: Tensor, far: Tensor, background_rgb: Optional[Tensor] = None,
                 cos_anneal_ratio: float = 0.0) -> Dict[str, Tensor]:
    """The same under autograd, as render_volume.py:160-200 uses it (perturb = 0): the hierarchical sampling runs without grad
    (renderer.py:387-409), render_core_outside and render_core with it.  `sc`'s state dicts / variance must be leaves that require
    grad (oracle.train_ref.leaf_state)."""
    return _render(sc, rays_o, rays_d, near, far, background_rgb, cos_anneal_ratio, training=True)


def _render(sc: NeusScene, rays_o: Tensor, rays_d: Tensor, near: Tensor, far: Tensor, background_rgb: Optional[Tensor],
            cos_anneal_ratio: float, training: bool) -> Dict[str, Tensor]:
    batch = len(rays_o)
    sample_dist = 2.0 / sc.n_samples
    z_vals = near + (far - near) * torch.linspace(0.0, 1.0, sc.n_samples)[None, :]
    z_out = None
    if sc.n_outside > 0:
        z_out = torch.linspace(1e-3, 1.0 - 1.0 / (sc.n_outside + 1.0), sc.n_outside)
        z_out = far / torch.flip(z_out, dims=[-1]) + 1.0 / sc.n_samples
    n = sc.n_samples
    if sc.n_importance > 0:
        with torch.no_grad():  # renderer.py:387
            pts = rays_o[:, None, :] + rays_d[:, None, :] * z_vals[..., :, None]
            sdf = sc.sdf(pts.reshape(-1, 3)).reshape(batch, sc.n_samples)
            for i in range(sc.up_sample_steps):
                new_z = up_sample(rays_o, rays_d, z_vals, sdf, sc.n_importance // sc.up_sample_steps, 64 * 2 ** i)
                z_vals, sdf = cat_z_vals(sc, rays_o, rays_d, z_vals, new_z, sdf, last=(i + 1 == sc.up_sample_steps))
        n = sc.n_samples + sc.n_importance
    bg_alpha = bg_color = None
    if sc.n_outside > 0:
        z_feed, _ = torch.sort(torch.cat([z_vals, z_out], dim=-1), dim=-1)
        ro = render_core_outside(sc, rays_o, rays_d, z_feed, sample_dist)
        bg_color, bg_alpha = ro["sampled_color"], ro["alpha"]
    fine = render_core(sc, rays_o, rays_d, z_vals, sample_dist, bg_alpha, bg_color, background_rgb, cos_anneal_ratio, training=training)
    w = fine["weights"]
    return {"color_fine": fine["color"], "s_val": fine["s_val"].reshape(batch, n).mean(dim=-1, keepdim=True),
            "cdf_fine": fine["cdf"], "weight_sum": w.sum(dim=-1, keepdim=True), "weight_max": torch.max(w, dim=-1, keepdim=True)[0],
            "gradients": fine["gradients"], "weights": w, "gradient_error": fine["gradient_error"],
            "inside_sphere": fine["inside_sphere"], "z_vals": z_vals}


# ----------------------------------------------------------------------------
# seeded parameter containers (constructor parity with the reference is checked by hash in the tests)
# ----------------------------------------------------------------------------
class NerfParams(torch.nn.Module):
    """Parameter layout and construction order of models/fields.py:243-297 (use_viewdirs=True): same nn.Linear shapes in
    the same order => the same torch RNG stream => the same initial weights and state_dict keys."""

    def __init__(self, spec: NerfSpec = NerfSpec()):
        super().__init__()
        in_ch = R.pe_width(spec.multires, spec.d_in)
        in_view = R.pe_width(spec.multires_view, spec.d_in_view)
        W = spec.W
        self.pts_linears = torch.nn.ModuleList(
            [torch.nn.Linear(in_ch, W)]
            + [torch.nn.Linear(W, W) if i not in spec.skips else torch.nn.Linear(W + in_ch, W) for i in range(spec.D - 1)])
        self.views_linears = torch.nn.ModuleList([torch.nn.Linear(in_view + W, W // 2)])
        self.feature_linear = torch.nn.Linear(W, W)
        self.alpha_linear = torch.nn.Linear(W, 1)
        self.rgb_linear = torch.nn.Linear(W // 2, 3)


class VarianceParams(torch.nn.Module):
    """models/fields.py:415-418."""

    def __init__(self, init_val: float):
        super().__init__()
        self.register_parameter("variance", torch.nn.Parameter(torch.tensor(init_val)))
