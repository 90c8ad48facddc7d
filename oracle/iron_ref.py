"""CPU oracle for IRON's stage-2 forward render path (TEST INFRASTRUCTURE ONLY).

This file is a from-scratch torch-CPU restatement of the reference algorithm
for the hot path named in BASELINE.json: positional encoding, the weight-normed
SDF / material MLPs, unit-sphere intersection, camera ray generation, the
sphere tracer (sphere tracing -> dense sampler -> bisection), material query,
the co-located GGX BRDF and the render_camera assembly.

It is the checker, never the product: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.  The shipped path is
iron_amd/ (HIP kernels behind the C ABI) and never routes through here.

Parity pin: tests/golden/*.npz were generated in the build container by
tests/golden/make_golden.py, which imports the real reference from
/root/reference and records its outputs; tests/test_oracle_golden.py checks
every function below against those vectors.  The op sequence is kept
op-for-op equivalent to the reference (separate mul/add, same torch kernels),
so on the same torch build the outputs are bit-identical.

Each function cites the reference file:line it restates (paths relative to the
reference root).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------
# positional encoding -- models/embedder.py:6-54
# ----------------------------------------------------------------------------
def positional_encoding(x: Tensor, n_freqs: int) -> Tensor:
    """[..., d] -> [..., d + 2*d*n_freqs]: cat[x, sin(x*2^0), cos(x*2^0), ...].

    models/embedder.py:22-36: freq bands are 2**linspace(0, L-1, L) (fp32
    0-dim tensors), each term is p_fn(x * freq).  n_freqs <= 0 -> identity.
    """
    if n_freqs <= 0:
        return x
    bands = 2.0 ** torch.linspace(0.0, float(n_freqs - 1), n_freqs)
    parts = [x]
    for fr in bands:
        parts.append(torch.sin(x * fr))
        parts.append(torch.cos(x * fr))
    return torch.cat(parts, dim=-1)


def pe_width(n_freqs: int, d: int = 3) -> int:
    return d if n_freqs <= 0 else d + 2 * d * n_freqs


# ----------------------------------------------------------------------------
# network descriptions (plain data; weights are a state_dict of CPU tensors)
# ----------------------------------------------------------------------------
@dataclass
class SDFSpec:
    """models/fields.py:9-45 constructor arguments that shape the forward."""
    d_in: int = 3
    d_out: int = 257
    d_hidden: int = 256
    n_layers: int = 8
    skip_in: Tuple[int, ...] = (4,)
    multires: int = 6
    scale: float = 1.0

    @property
    def n_linear(self) -> int:
        return self.n_layers + 1


@dataclass
class RenderSpec:
    """models/fields.py:141-201 constructor arguments that shape the forward."""
    d_feature: int = 256
    mode: str = "idr"
    d_in: int = 9
    d_out: int = 3
    d_hidden: int = 256
    n_layers: int = 4
    multires: int = 0
    multires_view: int = 0
    squeeze_out: bool = True
    squeeze_out_scale: float = 1.0
    output_bias: float = 0.0
    output_scale: float = 1.0
    skip_in: Tuple[int, ...] = ()

    @property
    def n_linear(self) -> int:
        return self.n_layers + 1


# the `ggx` branch of models/network_conf.py:48-122
GGX_SPECS: Dict[str, RenderSpec] = {
    "diffuse_albedo_network": RenderSpec(d_in=9, d_out=3, n_layers=4, multires_view=4, mode="idr", squeeze_out=True),
    "specular_albedo_network": RenderSpec(
        d_in=6, d_out=3, n_layers=4, multires=6, multires_view=-1, mode="no_view_dir",
        squeeze_out=False, output_bias=0.4, output_scale=0.1),
    "specular_roughness_network": RenderSpec(
        d_in=6, d_out=1, n_layers=4, multires=6, multires_view=-1, mode="no_view_dir",
        squeeze_out=False, output_bias=0.1, output_scale=0.1),
}


def effective_weight(sd: Dict[str, Tensor], l: int) -> Tuple[Tensor, Tensor]:
    """Old-style weight_norm(dim=0) folding: W = v * (g / ||v||_row).

    models/fields.py:75-76 (nn.utils.weight_norm) -> torch._weight_norm(v, g, 0).
    Plain (un-normed) layers carry `lin{l}.weight`.
    """
    key = "lin%d." % l
    if key + "weight_v" in sd:
        w = torch._weight_norm(sd[key + "weight_v"], sd[key + "weight_g"], 0)
    else:
        w = sd[key + "weight"]
    return w, sd[key + "bias"]


def softplus100(x: Tensor) -> Tensor:
    """nn.Softplus(beta=100) with torch's default threshold 20 (fields.py:80)."""
    return F.softplus(x, beta=100.0, threshold=20.0)


# ----------------------------------------------------------------------------
# SDF network -- models/fields.py:82-137
# ----------------------------------------------------------------------------
class EvalCounter:
    """Counts SDF point evaluations (SURVEY 8d's E) made through `sdf_fn`."""

    def __init__(self):
        self.evals = 0


def sdf_forward(sd: Dict[str, Tensor], spec: SDFSpec, x: Tensor) -> Tensor:
    """[M,3] -> [M,d_out]; column 0 is the signed distance (fields.py:82-98)."""
    inputs = x * spec.scale
    if spec.multires > 0:
        inputs = positional_encoding(inputs, spec.multires)
    h = inputs
    for l in range(spec.n_linear):
        w, b = effective_weight(sd, l)
        if l in spec.skip_in:
            h = torch.cat([h, inputs], dim=-1) / np.sqrt(2)
        h = F.linear(h, w, b)
        if l < spec.n_linear - 1:
            h = softplus100(h)
    return torch.cat([h[..., :1] / spec.scale, h[..., 1:]], dim=-1)


def sdf_get_all(sd: Dict[str, Tensor], spec: SDFSpec, x: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """sdf [M,1], feature [M,d_out-1], d sdf/dx [M,3] (fields.py:120-137, is_training=False)."""
    with torch.enable_grad():
        xg = x.detach().clone().requires_grad_(True)
        out = sdf_forward(sd, spec, xg)
        y, feat = out[..., :1], out[..., 1:]
        (grad,) = torch.autograd.grad(y, xg, torch.ones_like(y), create_graph=False, retain_graph=False)
    return y.detach(), feat.detach(), grad.detach()


# ----------------------------------------------------------------------------
# material ("rendering") network -- models/fields.py:203-239
# ----------------------------------------------------------------------------
def rendering_forward(sd: Dict[str, Tensor], spec: RenderSpec, points: Tensor, normals: Tensor,
                      view_dirs: Optional[Tensor], feats: Tensor) -> Tensor:
    if spec.multires > 0:
        points = positional_encoding(points, spec.multires)
    if spec.multires_view > 0 and spec.mode not in ("no_view_dir", "points_only"):
        view_dirs = positional_encoding(view_dirs, spec.multires_view)
    if spec.mode == "idr":
        inp = torch.cat([points, view_dirs, normals, feats], dim=-1)
    elif spec.mode == "no_view_dir":
        inp = torch.cat([points, normals, feats], dim=-1)
    elif spec.mode == "no_normal":
        inp = torch.cat([points, view_dirs, feats], dim=-1)
    elif spec.mode == "points_only":
        inp = torch.cat([points, feats], dim=-1)
    else:
        raise ValueError(spec.mode)
    h = inp
    for l in range(spec.n_linear):
        w, b = effective_weight(sd, l)
        if l in spec.skip_in:
            h = torch.cat([h, inp], dim=-1) / np.sqrt(2)
        h = F.linear(h, w, b)
        if l < spec.n_linear - 1:
            h = torch.relu(h)
    h = spec.output_scale * (h + spec.output_bias)
    if spec.squeeze_out:
        h = spec.squeeze_out_scale * torch.sigmoid(h)
    return h


def get_materials(nets: Dict[str, Tuple[Dict[str, Tensor], RenderSpec]], points: Tensor, normals: Tensor,
                  feats: Tensor, is_metal: bool = False) -> Dict[str, Tensor]:
    """models/rendering_func.py:5-16."""
    sd, sp = nets["diffuse_albedo_network"]
    diffuse = rendering_forward(sd, sp, points, normals, -normals, feats).abs()
    sd, sp = nets["specular_albedo_network"]
    spec = rendering_forward(sd, sp, points, normals, None, feats).abs()
    if not is_metal:
        spec = torch.mean(spec, dim=-1, keepdim=True).expand_as(spec)
    sd, sp = nets["specular_roughness_network"]
    rough = rendering_forward(sd, sp, points, normals, None, feats).abs() + 0.01
    return {"diffuse_albedo": diffuse, "specular_albedo": spec, "specular_roughness": rough}


# ----------------------------------------------------------------------------
# co-located GGX -- models/renderer_ggx.py:12-16, 61-146
# ----------------------------------------------------------------------------
def smith_g1(cos_theta: Tensor, alpha: Tensor) -> Tensor:
    sin_theta = torch.sqrt(1.0 - cos_theta * cos_theta)
    tan_theta = sin_theta / (cos_theta + 1e-10)
    root = alpha * tan_theta
    return 2.0 / (1.0 + torch.hypot(root, torch.ones_like(root)))


def ggx_colocated(light, distance: Tensor, normal: Tensor, viewdir: Tensor, params: Dict[str, Tensor],
                  mts_trans: Tensor, mts_diff_trans: Tensor) -> Dict[str, Tensor]:
    """renderer_ggx.py:82-146.  mts_trans: 5000 floats, mts_diff_trans: 50 floats."""
    kd, ks, alpha = params["diffuse_albedo"], params["specular_albedo"], params["specular_roughness"]
    intensity = light / (distance * distance + 1e-10)
    dot = torch.sum(viewdir * normal, dim=-1, keepdim=True)
    dot = torch.clamp(dot, min=0.00001, max=0.99999)
    m_eta = 1.48958738
    m_inv_eta2 = 1.0 / (m_eta * m_eta)
    alpha = torch.clamp(alpha, min=0.0001)
    c2 = dot * dot
    root = c2 + (1.0 - c2) / (alpha * alpha + 1e-10)
    D = 1.0 / (np.pi * alpha * alpha * root * root + 1e-10)
    Fr = 0.03867
    G = smith_g1(dot, alpha) ** 2
    specular = intensity * ks * Fr * D * G / (4.0 * dot + 1e-10)

    n_theta, n_alpha = 100, 50
    warped_cos = dot ** 0.25
    warped_alpha = ((alpha - 0) / (4 - 0)) ** 0.25
    tx = torch.floor(warped_cos * n_theta).long()
    ty = torch.floor(warped_alpha * n_alpha).long()
    t_idx = torch.clamp(ty * n_theta + tx, min=0, max=mts_trans.numel() - 1)
    T12 = torch.clamp(mts_trans[t_idx.squeeze(-1)].unsqueeze(-1), min=0.0, max=1.0)
    a_idx = torch.clamp(ty, min=0, max=mts_diff_trans.numel() - 1)
    Fdr = torch.clamp(1.0 - mts_diff_trans[a_idx.squeeze(-1)].unsqueeze(-1), min=0.0, max=1.0)
    diffuse = intensity * (kd / (1.0 - Fdr + 1e-10) / np.pi) * dot * T12 * T12 * m_inv_eta2
    return {"diffuse_rgb": diffuse, "specular_rgb": specular, "rgb": diffuse + specular}


# ----------------------------------------------------------------------------
# the fork's other co-located BRDF heads -- models/renderer_ggx.py:149-517, 520-858
# ----------------------------------------------------------------------------
def fresnel_dielectric(cos_i: Tensor, eta: Tensor) -> Tensor:
    """renderer_ggx.py:398-416 as its callers use it (cosThetaT argument unused; eta a tensor like cos_i)."""
    scale = torch.ones_like(cos_i) * eta
    m = cos_i > 0
    scale[m] = 1.0 / eta[m]
    cos_t_sqr = 1 - (1 - cos_i ** 2) * (scale ** 2)
    cos_i = torch.abs(cos_i)
    cos_t = torch.sqrt(cos_t_sqr)
    rs = (cos_i - eta * cos_t) / (cos_i + eta * cos_t)
    rp = (eta * cos_i - cos_t) / (eta * cos_i + cos_t)
    return 0.5 * (rs * rs + rp * rp)


def fresnel_conductor_exact(cos_i: Tensor, eta, k) -> Tensor:
    """renderer_ggx.py:419-432 (= CompositeRenderer.fresnel_conductor_exact :592-605); eta, k scalars or tensors."""
    c2 = cos_i * cos_i
    s2 = 1 - c2
    s4 = s2 * s2
    temp1 = eta * eta - k * k - s2
    a2pb2 = torch.sqrt(temp1 * temp1 + 4 * k * k * eta * eta)
    a = torch.sqrt(0.5 * (a2pb2 + temp1))
    term1 = a2pb2 + c2
    term2 = 2 * a * cos_i
    rs2 = (term1 - term2) / (term1 + term2)
    term3 = a2pb2 * c2 + s4
    term4 = term2 * s2
    rp2 = rs2 * (term3 - term4) / (term3 + term4)
    return 0.5 * (rp2 + rs2)


def _coloc_common(light, distance: Tensor, normal: Tensor, viewdir: Tensor):
    intensity = light / (distance * distance + 1e-10)
    dot = torch.clamp(torch.sum(viewdir * normal, dim=-1, keepdim=True), min=0.00001, max=0.99999)
    return intensity, dot


def smooth_dielectric(light, distance, normal, viewdir, kd: Tensor, ks: Tensor, alpha=None) -> Dict[str, Tensor]:
    """SmoothDielectricRenderer.forward, renderer_ggx.py:171-204: constant F = 0.04."""
    intensity, _ = _coloc_common(light, distance, normal, viewdir)
    spec = intensity * ks * 0.04
    diff = intensity * kd * 0.0001
    return {"diffuse_rgb": diff, "specular_rgb": spec, "rgb": diff + spec}


def thin_dielectric(light, distance, normal, viewdir, kd: Tensor, ks: Tensor, alpha=None) -> Dict[str, Tensor]:
    """ThinDielectricRenderer.forward, renderer_ggx.py:229-267: R = 0.04 with the inter-reflection series."""
    intensity, _ = _coloc_common(light, distance, normal, viewdir)
    R = 0.04
    T = 1 - R
    if R < 1:
        R += T * T * R / (1 - R * R)
    spec = intensity * ks * R
    diff = intensity * kd * 0.0001
    return {"diffuse_rgb": diff, "specular_rgb": spec, "rgb": diff + spec}


def smooth_conductor(light, distance, normal, viewdir, kd: Tensor, ks: Tensor, alpha=None, eta: float = 2.58,
                     k: float = 8.21) -> Dict[str, Tensor]:
    """SmoothConductorCoLocRenderer.forward, renderer_ggx.py:299-319."""
    intensity, dot = _coloc_common(light, distance, normal, viewdir)
    spec = intensity * ks * fresnel_conductor_exact(dot, eta, k)
    diff = intensity * kd * 0.0001
    return {"diffuse_rgb": diff, "specular_rgb": spec, "rgb": diff + spec}


def rough_conductor(light, distance, normal, viewdir, kd: Tensor, ks: Tensor, alpha: Tensor, eta: float = 2.58,
                    k: float = 8.21) -> Dict[str, Tensor]:
    """RoughConductorCoLocRenderer.forward, renderer_ggx.py:351-395."""
    intensity, dot = _coloc_common(light, distance, normal, viewdir)
    alpha = torch.clamp(alpha, min=0.0001)
    c2 = dot * dot
    root = c2 + (1.0 - c2) / (alpha * alpha + 1e-10)
    D = 1.0 / (np.pi * alpha * alpha * root * root + 1e-10)
    Fr = fresnel_conductor_exact(dot, eta, k)
    G = smith_g1(dot, alpha) ** 2
    spec = intensity * ks * Fr * D * G / (4.0 * dot + 1e-10)
    diff = intensity * kd * 0.0001
    return {"diffuse_rgb": diff, "specular_rgb": spec, "rgb": diff + spec}


def _diffuse_ggx_tables(intensity: Tensor, cos_theta: Tensor, alpha: Tensor, kd: Tensor, mts_trans: Tensor,
                        mts_diff_trans: Tensor, eta: float = 1.48958738) -> Tensor:
    """CompositeRenderer.diffuse_reflection_ggx, renderer_ggx.py:654-681."""
    alpha = torch.clamp(alpha, min=0.0001)
    inv_eta2 = 1.0 / (eta * eta)
    n_theta, n_alpha = 100, 50
    warped_cos = cos_theta ** 0.25
    warped_alpha = ((alpha - 0) / (4 - 0)) ** 0.25
    tx = torch.floor(warped_cos * n_theta).long()
    ty = torch.floor(warped_alpha * n_alpha).long()
    t_idx = torch.clamp(ty * n_theta + tx, min=0, max=mts_trans.numel() - 1)
    T12 = torch.clamp(mts_trans[t_idx.squeeze(-1)].unsqueeze(-1), min=0.0, max=1.0)
    a_idx = torch.clamp(ty, min=0, max=mts_diff_trans.numel() - 1)
    Fdr = torch.clamp(1.0 - mts_diff_trans[a_idx.squeeze(-1)].unsqueeze(-1), min=0.0, max=1.0)
    return intensity * (kd / (1.0 - Fdr + 1e-10) / np.pi) * cos_theta * T12 * T12 * inv_eta2


def composite_forward(light, distance: Tensor, normal: Tensor, viewdir: Tensor, params: Dict[str, Tensor],
                      mts_trans: Tensor, mts_diff_trans: Tensor, use_env_light: bool = False) -> Dict[str, Tensor]:
    """CompositeRenderer.forward, renderer_ggx.py:781-858, quirks included: the GGX NDF is evaluated with
    alpha := 1.48958738 (`calc_D_specular(cos, eta)`, :806); the `metallic` / `dielectric` weights are clamped and
    then unused (the weighted sum :829 is overwritten by the plain sum :831); `rgb` aliases `diffuse_rgb` and is
    updated in place (:847-853), so the returned "diffuse_rgb" equals "rgb"."""
    rough = torch.clamp(params["specular_roughness"], min=0.00001)
    d_eta = torch.clamp(params["dielectric_eta"], min=1.000001, max=1.999999)
    m_eta = torch.clamp(params["metallic_eta"], min=0.099999, max=4.999999)
    m_k = torch.clamp(params["metallic_k"], min=0.099999, max=9.999999)
    ks = torch.clamp(params["specular_albedo"], min=0.00001)
    kd = torch.clamp(params["diffuse_albedo"], min=0.00001)
    eta = 1.48958738
    cos_i = torch.clamp(torch.sum(viewdir * normal, dim=-1, keepdim=True), min=0.00001, max=0.99999)
    c2 = cos_i * cos_i
    root = c2 + (1.0 - c2) / (eta * eta + 1e-10)
    D = 1.0 / (np.pi * eta * eta * root * root + 1e-10)
    G = smith_g1(cos_i, rough) * smith_g1(cos_i, rough)
    if use_env_light:
        intensity = torch.clamp(params["env_light"], min=0.000001, max=20.0)
    else:
        intensity = light / (distance * distance + 1e-10)
    metallic_rgb = ks * fresnel_conductor_exact(cos_i, m_eta, m_k)
    dielectric_rgb = ks * fresnel_dielectric(cos_i, d_eta) * D * G / (4.0 * torch.abs(cos_i))
    metallic_rgb = metallic_rgb * intensity
    dielectric_rgb = dielectric_rgb * intensity
    specular = dielectric_rgb + metallic_rgb
    rgb = _diffuse_ggx_tables(intensity, cos_i, rough, kd, mts_trans, mts_diff_trans) + specular
    ret = {"diffuse_rgb": rgb, "specular_rgb": specular, "metallic_rgb": metallic_rgb, "dielectric_rgb": dielectric_rgb,
           "rgb": rgb}
    if use_env_light:
        ret["env_light"] = intensity
    return ret


def _ns(**kw) -> RenderSpec:
    return RenderSpec(d_in=6, n_layers=4, multires=6, multires_view=-1, mode="no_view_dir", squeeze_out=False, **kw)


# the `comp2` branch of models/network_conf.py:318-447 (render_surface.py asks for 'comp', which the factory as
# shipped does not define; comp2 is the branch whose keys get_materials_comp consumes)
COMP_SPECS: Dict[str, RenderSpec] = {
    "diffuse_albedo_network": RenderSpec(d_in=9, d_out=3, n_layers=4, multires_view=4, mode="idr", squeeze_out=True),
    "specular_albedo_network": _ns(d_out=3, output_bias=0.0, output_scale=1.0),
    "specular_roughness_network": _ns(d_out=1, output_bias=0.1, output_scale=1.0),
    "metallic_network": _ns(d_out=1, output_bias=0.1, output_scale=1.0),
    "dielectric_network": _ns(d_out=1, output_bias=0.1, output_scale=1.0),
    "metallic_eta_network": _ns(d_out=1, output_bias=0.1, output_scale=1.0),
    "metallic_k_network": _ns(d_out=1, output_bias=0.1, output_scale=1.0),
    "dielectric_eta_network": _ns(d_out=1, output_bias=0.1, output_scale=1.0),
}


def get_materials_comp(nets: Dict[str, Tuple[Dict[str, Tensor], RenderSpec]], points: Tensor, normals: Tensor,
                       feats: Tensor) -> Dict[str, Tensor]:
    """models/rendering_func.py:19-49."""
    def run(name, view):
        sd, sp = nets[name]
        return rendering_forward(sd, sp, points, normals, view, feats).abs()
    return {"diffuse_albedo": run("diffuse_albedo_network", -normals),
            "specular_albedo": run("specular_albedo_network", None),
            "metallic": run("metallic_network", None),
            "dielectric": run("dielectric_network", None),
            "specular_roughness": run("specular_roughness_network", None),
            "metallic_eta": run("metallic_eta_network", None),
            "metallic_k": run("metallic_k_network", None),
            "dielectric_eta": run("dielectric_eta_network", None)}


COMP_RENDER_KEYS = (("color", 3), ("diffuse_color", 3), ("specular_color", 3), ("diffuse_albedo", 3), ("specular_albedo", 3),
                    ("specular_roughness", 1), ("metallic_eta", 1), ("metallic_k", 1), ("dielectric_eta", 1), ("normal", 3),
                    ("metallic_rgb", 3), ("metallic", 1), ("dielectric_rgb", 3), ("dielectric", 1))


def render_fn_comp(nets, light: float, mts_trans: Tensor, mts_diff_trans: Tensor, interior_mask: Tensor, ray_o: Tensor,
                   ray_d: Tensor, points: Tensor, normals: Tensor, feats: Tensor) -> Dict[str, Tensor]:
    """render_surface.py:159-234 with the composite renderer (scalar buffers keep their trailing [.., 1])."""
    sh = list(interior_mask.shape)
    out = {k: torch.zeros(sh + [w], dtype=torch.float32) for k, w in COMP_RENDER_KEYS}
    if interior_mask.any():
        n = normals / (normals.norm(dim=-1, keepdim=True) + 1e-10)
        prm = get_materials_comp(nets, points, n, feats)
        res = composite_forward(light if torch.is_tensor(light) else torch.tensor(light, dtype=torch.float32), (points - ray_o).norm(dim=-1, keepdim=True), n,
                                -ray_d, prm, mts_trans, mts_diff_trans)
        out["color"][interior_mask] = res["rgb"]
        out["diffuse_color"][interior_mask] = res["diffuse_rgb"]
        out["specular_color"][interior_mask] = res["specular_rgb"]
        out["metallic_rgb"][interior_mask] = res["metallic_rgb"]
        out["dielectric_rgb"][interior_mask] = res["dielectric_rgb"]
        for k in ("diffuse_albedo", "specular_albedo", "specular_roughness", "metallic_eta", "metallic_k", "dielectric_eta",
                  "metallic", "dielectric"):
            out[k][interior_mask] = prm[k]
        out["normal"][interior_mask] = n
    return out


# ----------------------------------------------------------------------------
# geometry helpers -- models/raytracer.py:223-303
# ----------------------------------------------------------------------------
def intersect_sphere(ray_o: Tensor, ray_d: Tensor, r: float = 1.0) -> Tuple[Tensor, Tensor, Tensor]:
    """raytracer.py:223-237."""
    d1 = -torch.sum(ray_d * ray_o, dim=-1) / torch.sum(ray_d * ray_d, dim=-1)
    p = ray_o + d1.unsqueeze(-1) * ray_d
    tmp = r * r - torch.sum(p * p, dim=-1)
    mask = tmp > 0.0
    d2 = torch.sqrt(torch.clamp(tmp, min=0.0)) / torch.norm(ray_d, dim=-1)
    return mask, torch.clamp(d1 - d2, min=0.0), d1 + d2


@dataclass
class CameraSpec:
    W: int
    H: int
    K: Tensor  # [4,4]
    W2C: Tensor  # [4,4]

    def __post_init__(self):
        self.K_inv = torch.inverse(self.K)   # raytracer.py:250
        self.C2W = torch.inverse(self.W2C)   # raytracer.py:251

    def get_uv(self) -> Tensor:
        """raytracer.py:300-303: pixel centres, [H,W,2]."""
        u, v = np.meshgrid(np.arange(self.W), np.arange(self.H))
        return torch.from_numpy(np.stack((u, v), axis=-1).astype(np.float32)) + 0.5

    def get_rays(self, uv: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        """raytracer.py:254-286."""
        sh = list(uv.shape[:-1])
        uv = uv.reshape(-1, 2)
        uv1 = torch.cat((uv, torch.ones_like(uv[..., 0:1])), dim=-1)
        d = torch.matmul(torch.matmul(uv1, self.K_inv[:3, :3].transpose(1, 0)),
                         self.C2W[:3, :3].transpose(1, 0)).reshape(sh + [3])
        dn = d.norm(dim=-1)
        d = d / dn.unsqueeze(-1)
        o = self.C2W[:3, 3].unsqueeze(0).expand(uv1.shape[0], -1).reshape(sh + [3])
        return o, d, dn

    def crop(self, tw: int, th: int, ul: Tuple[int, int]) -> "CameraSpec":
        """raytracer.py:327-351 with an explicit ul_corner=(col,row)."""
        K = self.K.clone()
        K[0, 2] -= ul[0]
        K[1, 2] -= ul[1]
        return CameraSpec(tw, th, K, self.W2C.clone())

    def scaled(self, W: int, H: int) -> "CameraSpec":
        """raytracer.py:353-364 (Camera.resize without the image)."""
        K = self.K.clone()
        K[0, :3] *= W / self.W
        K[1, :3] *= H / self.H
        return CameraSpec(W, H, K, self.W2C.clone())


# ----------------------------------------------------------------------------
# ray tracer -- models/raytracer.py:27-220
# ----------------------------------------------------------------------------
@dataclass
class TracerParams:
    sdf_threshold: float = 5.0e-5
    sphere_tracing_iters: int = 16
    n_steps: int = 128
    max_num_pts: int = 200000


def sphere_tracing(sdf_fn: Callable[[Tensor], Tensor], ray_o, ray_d, min_dis, max_dis, work_mask,
                   prm: TracerParams):
    """raytracer.py:105-140."""
    iters = 0
    unfinished = work_mask.clone()
    t = min_dis.clone()
    p = ray_o + ray_d * t.unsqueeze(-1)
    s = sdf_fn(p)
    while True:
        unfinished = unfinished & (s.abs() > prm.sdf_threshold) & (t < max_dis)
        if iters == prm.sphere_tracing_iters or unfinished.sum() == 0:
            break
        iters += 1
        step = s[unfinished]
        t[unfinished] += step
        p[unfinished] += ray_d[unfinished] * step.unsqueeze(-1)
        s[unfinished] = sdf_fn(p[unfinished])
    conv = work_mask & ~unfinished & (s.abs() <= prm.sdf_threshold) & (t < max_dis)
    return conv, unfinished, p, s, t


def rootfind(sdf_fn, f_low, f_high, d_low, d_high, ray_o, ray_d, prm: TracerParams):
    """raytracer.py:199-220.  NB every ray is updated each iteration; the loop runs while ANY
    initially bracketed ray is wider than 2*threshold, so the count is batch-global."""
    work = (f_low > 0) & (f_high < 0)
    d_mid = (d_low + d_high) / 2.0
    n_iter = 0
    while work.any():
        f_mid = sdf_fn(ray_o + ray_d * d_mid.unsqueeze(-1))
        lo = f_mid > 0
        hi = f_mid <= 0
        if lo.sum() > 0:
            d_low[lo] = d_mid[lo]
            f_low[lo] = f_mid[lo]
        if hi.sum() > 0:
            d_high[hi] = d_mid[hi]
            f_high[hi] = f_mid[hi]
        d_mid = (d_low + d_high) / 2.0
        work &= (d_high - d_low) > 2 * prm.sdf_threshold
        n_iter += 1
    p_mid = ray_o + ray_d * d_mid.unsqueeze(-1)
    f_mid = sdf_fn(p_mid)
    return p_mid, d_mid, f_mid, n_iter


def ray_sampler(sdf_fn, ray_o, ray_d, min_dis, max_dis, prm: TracerParams):
    """raytracer.py:142-197."""
    n = prm.n_steps
    lin = torch.linspace(0, 1, steps=n).float().view(1, n)
    z = min_dis.unsqueeze(-1) + lin * (max_dis.unsqueeze(-1) - min_dis.unsqueeze(-1))
    pts = ray_o.unsqueeze(-2) + ray_d.unsqueeze(-2) * z.unsqueeze(-1)
    vals = [sdf_fn(c) for c in torch.split(pts.reshape(-1, 3), prm.max_num_pts, dim=0)]
    val = torch.cat(vals, dim=0).reshape(-1, n)

    out_p = torch.zeros_like(ray_d)
    out_s = torch.zeros_like(min_dis)
    out_t = torch.zeros_like(min_dis)
    weights = torch.arange(n, 0, -1).float().reshape(1, n)
    mn, idx = torch.min(torch.sign(val) * weights, dim=-1)
    root = (mn < 0.0) & (idx >= 1)
    n_iter = 0
    if root.sum() > 0:
        i1 = idx[root].unsqueeze(-1)
        z_lo = torch.gather(z[root], -1, i1 - 1).squeeze(-1)
        f_lo = torch.gather(val[root], -1, i1 - 1).squeeze(-1)
        z_hi = torch.gather(z[root], -1, i1).squeeze(-1)
        f_hi = torch.gather(val[root], -1, i1).squeeze(-1)
        p, t, s, n_iter = rootfind(sdf_fn, f_lo, f_hi, z_lo, z_hi, ray_o[root], ray_d[root], prm)
        out_p[root] = p
        out_s[root] = s
        out_t[root] = t
    return root, out_p, out_s, out_t, n_iter


def raytracer_forward(sdf_fn, ray_o, ray_d, min_dis, max_dis, work_mask,
                      prm: TracerParams = TracerParams(), stats: Optional[dict] = None) -> Dict[str, Tensor]:
    """raytracer.py:45-103 (non-verbose).  `stats` (optional) receives n_sampler / n_bisect_iters."""
    conv, unfinished, p, s, t = sphere_tracing(sdf_fn, ray_o, ray_d, min_dis, max_dis, work_mask, prm)
    smask = unfinished
    n_iter = 0
    if smask.sum() > 0:
        pos = (s[smask] > 0.0).float()
        s_min = pos * t[smask] + (1.0 - pos) * min_dis[smask]
        s_max = pos * max_dis[smask] + (1.0 - pos) * t[smask]
        sc, sp, ss, st, n_iter = ray_sampler(sdf_fn, ray_o[smask], ray_d[smask], s_min, s_max, prm)
        conv[smask] = sc
        p[smask] = sp
        s[smask] = ss
        t[smask] = st
    if stats is not None:
        stats["n_sampler"] = stats.get("n_sampler", 0) + int(smask.sum())
        stats.setdefault("bisect_iters", []).append(n_iter)
    return {"convergent_mask": conv, "points": p, "sdf": s, "distance": t}


# ----------------------------------------------------------------------------
# stage-2 assembly -- models/raytracer.py:367-409, 542-552, 593-662, 778-814
#                     render_surface.py:117-156 (render_fn)
# ----------------------------------------------------------------------------
@dataclass
class Scene:
    """Everything a render needs, as plain CPU tensors."""
    sdf_sd: Dict[str, Tensor]
    sdf_spec: SDFSpec
    nets: Dict[str, Tuple[Dict[str, Tensor], RenderSpec]]
    light: float
    mts_trans: Tensor
    mts_diff_trans: Tensor
    counter: EvalCounter = field(default_factory=EvalCounter)
    renderer: str = "ggx"  # "ggx": render_surface.py:117-156; "comp": the composite render_fn, :159-234 (row f-4)

    def sdf_fn(self, x: Tensor) -> Tensor:
        self.counter.evals += int(x.shape[0])
        return sdf_forward(self.sdf_sd, self.sdf_spec, x)[..., 0]


@torch.no_grad()
def raytrace_pixels(scene: Scene, uv: Tensor, cam: CameraSpec, mask: Optional[Tensor] = None,
                    max_num_rays: int = 200000, prm: TracerParams = TracerParams(),
                    stats: Optional[dict] = None) -> Dict[str, Tensor]:
    """raytracer.py:367-409."""
    if mask is None:
        mask = torch.ones_like(uv[..., 0]).bool()
    sh = list(uv.shape[:-1])
    ray_o, ray_d, ray_d_norm = cam.get_rays(uv)
    merged: Dict[str, List[Tensor]] = {}
    for o_c, d_c, n_c, m_c in zip(torch.split(ray_o.reshape(-1, 3), max_num_rays, dim=0),
                                  torch.split(ray_d.reshape(-1, 3), max_num_rays, dim=0),
                                  torch.split(ray_d_norm.reshape(-1), max_num_rays, dim=0),
                                  torch.split(mask.reshape(-1), max_num_rays, dim=0)):
        hit, near, far = intersect_sphere(o_c, d_c, 1.0)
        res = raytracer_forward(scene.sdf_fn, o_c, d_c, near, far, hit & m_c, prm, stats)
        res["depth"] = res["distance"] / n_c
        for k, v in res.items():
            merged.setdefault(k, []).append(v)
    out: Dict[str, Tensor] = {}
    for k, parts in merged.items():
        v = torch.cat(parts, dim=0).reshape(sh + [-1])
        out[k] = v[..., 0] if v.shape[-1] == 1 else v
    out.update({"uv": uv, "ray_o": ray_o, "ray_d": ray_d, "ray_d_norm": ray_d_norm})
    return out


@torch.no_grad()
def raytrace_camera(scene: Scene, cam: CameraSpec, max_num_rays: int = 200000,
                    prm: TracerParams = TracerParams(), stats: Optional[dict] = None) -> Dict[str, Tensor]:
    """raytracer.py:542-552 (fill_holes=False, detect_edges=False)."""
    res = raytrace_pixels(scene, cam.get_uv(), cam, max_num_rays=max_num_rays, prm=prm, stats=stats)
    res["depth"] *= res["convergent_mask"].float()
    return res


def render_fn_ggx(scene: Scene, interior_mask: Tensor, ray_o: Tensor, ray_d: Tensor, points: Tensor,
                  normals: Tensor, feats: Tensor) -> Dict[str, Tensor]:
    """render_surface.py:117-156 with the GGX renderer."""
    sh = list(interior_mask.shape)
    rgb = torch.zeros(sh + [3], dtype=torch.float32)
    out = {k: rgb.clone() for k in ("color", "diffuse_color", "specular_color", "diffuse_albedo",
                                    "specular_albedo", "normal")}
    out["specular_roughness"] = rgb[..., 0].clone()
    if interior_mask.any():
        n = normals / (normals.norm(dim=-1, keepdim=True) + 1e-10)
        prm = get_materials(scene.nets, points, n, feats)
        res = ggx_colocated(scene.light if torch.is_tensor(scene.light) else torch.tensor(scene.light, dtype=torch.float32),
                            (points - ray_o).norm(dim=-1, keepdim=True), n, -ray_d, prm,
                            scene.mts_trans, scene.mts_diff_trans)
        out["color"][interior_mask] = res["rgb"]
        out["diffuse_color"][interior_mask] = res["diffuse_rgb"]
        out["specular_color"][interior_mask] = res["specular_rgb"]
        out["diffuse_albedo"][interior_mask] = prm["diffuse_albedo"]
        out["specular_albedo"][interior_mask] = prm["specular_albedo"]
        out["specular_roughness"][interior_mask] = prm["specular_roughness"].squeeze(-1)
        out["normal"][interior_mask] = n
    return out


def render_normal_and_color(scene: Scene, results: Dict[str, Tensor], max_num_pts: int = 320000) -> None:
    """raytracer.py:593-662 (is_training=False); mutates `results`."""
    sh = list(results["convergent_mask"].shape)
    merged: Dict[str, List[Tensor]] = {}
    for p_c, d_c, o_c, m_c in zip(torch.split(results["points"].reshape(-1, 3), max_num_pts, dim=0),
                                  torch.split(results["ray_d"].reshape(-1, 3), max_num_pts, dim=0),
                                  torch.split(results["ray_o"].reshape(-1, 3), max_num_pts, dim=0),
                                  torch.split(results["convergent_mask"].reshape(-1), max_num_pts, dim=0)):
        if m_c.any():
            p_h, d_h, o_h = p_c[m_c], d_c[m_c], o_c[m_c]
            _, feat, grad = sdf_get_all(scene.sdf_sd, scene.sdf_spec, p_h)
        else:
            p_h = d_h = o_h = grad = feat = torch.zeros(0, dtype=torch.float32)
        with torch.no_grad():
            if scene.renderer == "comp":
                r = render_fn_comp(scene.nets, scene.light, scene.mts_trans, scene.mts_diff_trans, m_c, o_h, d_h, p_h, grad, feat)
            else:
                r = render_fn_ggx(scene, m_c, o_h, d_h, p_h, grad, feat)
        for k, v in r.items():
            merged.setdefault(k, []).append(v)
    for k, parts in merged.items():
        v = torch.cat(parts, dim=0).reshape(sh + [-1])
        results[k] = v.squeeze(-1) if v.shape[-1] == 1 else v


def render_camera(scene: Scene, cam: CameraSpec, prm: TracerParams = TracerParams(),
                  stats: Optional[dict] = None) -> Dict[str, Tensor]:
    """raytracer.py:778-814 with fill_holes=False, handle_edges=False, is_training=False."""
    res = raytrace_camera(scene, cam, max_num_rays=50000, prm=prm, stats=stats)
    render_normal_and_color(scene, res, max_num_pts=320000)
    return res


# ----------------------------------------------------------------------------
# silhouette handling -- models/raytracer.py:412-539, 554-585, 665-729
# ----------------------------------------------------------------------------
def morph_closing3x3(depth: Tensor) -> Tensor:
    """kornia.morphology.closing(depth[None,None], ones(3,3)) as raytracer.py:554-557 calls it: erosion(dilation(x))
    with kornia's default 'geodesic' border (the border never wins: -/+ 1e4 padding) => min-pool3(max-pool3(x)).
    PARITY UNPINNED: kornia is not installed in the build container; restated from kornia's documented semantics."""
    x = depth[None, None]
    big = 1e4
    d = F.max_pool2d(F.pad(x, (1, 1, 1, 1), value=-big), 3, stride=1)
    e = -F.max_pool2d(F.pad(-d, (1, 1, 1, 1), value=-big), 3, stride=1)
    return e[0, 0]


def sobel_magnitude(depth: Tensor) -> Tensor:
    """kornia.filters.sobel(depth[None,None]) (normalized=True, eps=1e-6) as raytracer.py:569 calls it: 3x3 Sobel
    kernels divided by 8, replicate padding, sqrt(gx^2 + gy^2 + 1e-6).  PARITY UNPINNED (see morph_closing3x3)."""
    x = F.pad(depth[None, None], (1, 1, 1, 1), mode="replicate")
    kx = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]]) / 8.0
    gx = F.conv2d(x, kx[None, None])
    gy = F.conv2d(x, kx.t().contiguous()[None, None])
    return torch.sqrt(gx * gx + gy * gy + 1e-6)[0, 0]


def unique_first(x: Tensor):
    """raytracer.py:412-419: unique values and, for each, the index of its FIRST occurrence in x."""
    uniq, inverse = torch.unique(x, return_inverse=True, dim=0)
    perm = torch.arange(inverse.size(0), dtype=inverse.dtype)
    inverse, perm = inverse.flip([0]), perm.flip([0])
    return uniq, inverse.new_empty(uniq.size(0)).scatter_(0, inverse, perm)


def project(cam: CameraSpec, points: Tensor) -> Tensor:
    """raytracer.py:305-325."""
    p = torch.cat([points, torch.ones_like(points[:, :1])], dim=1)
    uv = torch.matmul(torch.matmul(p, cam.W2C.transpose(1, 0)), cam.K.transpose(1, 0))
    return uv[:, :2] / uv[:, 2:3]


@torch.no_grad()
def locate_edge_points(scene: Scene, cam: CameraSpec, walk_start_points: Tensor, mask: Tensor, max_step: int = 16,
                       step_size: float = 1e-3, dot_threshold: float = 5e-2, max_num_rays: int = 200000) -> Dict[str, Tensor]:
    """raytracer.py:421-506: walk on the surface towards the silhouette (|n.v| <= dot_threshold)."""
    finish = walk_start_points.clone()
    found = mask.clone()
    if mask.sum() > 0:
        sh = list(walk_start_points.shape[:-1])
        fin_parts, found_parts = [], []
        cam_o = cam.C2W[:3, 3]
        for cur in torch.split(walk_start_points[mask].clone().view(-1, 3), max_num_rays, dim=0):
            f = torch.zeros_like(cur[..., 0]).bool()
            nf = ~f
            ray_o = cam_o.view(1, 3).expand(cur.shape[0], 3)
            i = 0
            while True:
                view = ray_o[nf] - cur[nf]
                view = view / (view.norm(dim=-1, keepdim=True) + 1e-10)
                sdf, _, nrm = sdf_get_all(scene.sdf_sd, scene.sdf_spec, cur[nf].view(-1, 3))
                nrm = nrm / (nrm.norm(dim=-1, keepdim=True) + 1e-10)
                dot = (nrm * view).sum(dim=-1)
                tmp_nf = dot.abs() > dot_threshold
                f[nf] = ~tmp_nf
                nf = ~f
                if i >= max_step or nf.sum() == 0:
                    break
                walk = nrm - view / dot.unsqueeze(-1)
                walk = walk / (walk.norm(dim=-1, keepdim=True) + 1e-10)
                walk = walk - sdf * nrm
                cur[nf] += (step_size * walk)[tmp_nf]
                i += 1
            fin_parts.append(cur)
            found_parts.append(f)
        finish[mask] = torch.cat(fin_parts, dim=0)
        found[mask] = torch.cat(found_parts, dim=0)
        finish = finish.reshape(sh + [3])
        found = found.reshape(sh)
    edge_points = finish[found]
    edge_mask = torch.zeros(cam.H, cam.W).bool()
    edge_uv = torch.zeros_like(edge_points[..., :2])
    upd = torch.zeros(0, dtype=torch.long)
    if found.any():
        edge_uv = project(cam, edge_points)
        upd = torch.floor(edge_uv).long()
        upd = upd[:, 1] * cam.W + upd[:, 0]
        ok = (upd < cam.H * cam.W) & (upd >= 0)
        upd, edge_points, edge_uv = upd[ok], edge_points[ok], edge_uv[ok]
        if ok.any():
            cnt = upd.shape[0]
            upd, uidx = unique_first(upd)
            uidx = torch.arange(cnt)[uidx]
            edge_points = edge_points[uidx]
            edge_uv = edge_uv[uidx]
            edge_mask.view(-1)[upd] = True
    return {"edge_mask": edge_mask, "edge_points": edge_points, "edge_uv": edge_uv, "edge_pixel_idx": upd}


@torch.no_grad()
def raytrace_camera_full(scene: Scene, cam: CameraSpec, max_num_rays: int = 200000, fill_holes: bool = False,
                         detect_edges: bool = False, prm: TracerParams = TracerParams(),
                         depth_edge_mask: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """raytracer.py:542-590.  `depth_edge_mask` (optional) replaces the sobel-derived mask (used to pin the rest of the
    chain to reference goldens, since kornia's sobel cannot run in the build container)."""
    res = raytrace_camera(scene, cam, max_num_rays=max_num_rays, prm=prm)
    if fill_holes:
        depth = morph_closing3x3(res["depth"])
        new_conv = depth > 1e-2
        upd = new_conv & (~res["convergent_mask"])
        if upd.any():
            res["depth"][upd] = depth[upd]
            res["convergent_mask"] = new_conv
            res["distance"] = res["depth"] * res["ray_d_norm"]
            res["points"] = res["ray_o"] + res["ray_d"] * res["distance"].unsqueeze(-1)
    if detect_edges:
        if depth_edge_mask is None:
            depth_edge_mask = (sobel_magnitude(res["depth"]) > 1e-2) & res["convergent_mask"]
        res.update(locate_edge_points(scene, cam, res["points"], depth_edge_mask, max_step=16, step_size=1e-3,
                                      dot_threshold=5e-2, max_num_rays=max_num_rays))
        res["convergent_mask"] = res["convergent_mask"] & ~res["edge_mask"]
    return res


def render_edge_pixels(scene: Scene, results: Dict[str, Tensor], cam: CameraSpec, prm: TracerParams = TracerParams()) -> None:
    """raytracer.py:665-729 (is_training=False): two side rays per edge pixel, area-weighted blend; mutates results."""
    edge_points, edge_uv, edge_idx = results["edge_points"], results["edge_uv"], results["edge_pixel_idx"]
    center = torch.floor(edge_uv) + 0.5
    _, _, grads = sdf_get_all(scene.sdf_sd, scene.sdf_spec, edge_points)
    nrm = grads / (grads.norm(dim=-1, keepdim=True) + 1e-10)
    n2d = torch.matmul(nrm, cam.W2C[:3, :3].transpose(1, 0))[:, :2]
    n2d = n2d / (n2d.norm(dim=-1, keepdim=True) + 1e-10)
    radius = 0.707
    pos_uv = center - radius * n2d
    neg_uv = center + radius * n2d
    dot2d = torch.sum((edge_uv - center) * n2d, dim=-1)
    alpha = 2 * torch.arccos(torch.clamp(dot2d / radius, min=0.0, max=1.0))
    w_pos = 1.0 - (alpha - torch.sin(alpha)) / (2.0 * np.pi)
    pos = raytrace_pixels(scene, pos_uv, cam, prm=prm)
    neg = raytrace_pixels(scene, neg_uv, cam, prm=prm)
    render_normal_and_color(scene, pos)
    render_normal_and_color(scene, neg)
    color = pos["color"] * w_pos.unsqueeze(-1) + neg["color"] * (1.0 - w_pos.unsqueeze(-1))
    results["color"].view(-1, 3)[edge_idx] = color
    results["normal"].view(-1, 3)[edge_idx] = grads
    results["edge_pos_neg_normal"] = torch.cat([pos["normal"][pos["convergent_mask"]], neg["normal"][neg["convergent_mask"]]], dim=0)
    results["uv"].view(-1, 2)[edge_idx] = edge_uv
    results["points"].view(-1, 3)[edge_idx] = edge_points


def render_camera_full(scene: Scene, cam: CameraSpec, fill_holes: bool = False, handle_edges: bool = True,
                       prm: TracerParams = TracerParams(), depth_edge_mask: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """raytracer.py:778-814 with is_training=False."""
    res = raytrace_camera_full(scene, cam, max_num_rays=50000, fill_holes=fill_holes, detect_edges=handle_edges, prm=prm,
                               depth_edge_mask=depth_edge_mask)
    render_normal_and_color(scene, res, max_num_pts=320000)
    if handle_edges and res["edge_mask"].sum() > 0:
        render_edge_pixels(scene, res, cam, prm=prm)
    return res


# ----------------------------------------------------------------------------
# algorithmic work
# ----------------------------------------------------------------------------
FLOP_PER_SDF_EVAL = 2 * (39 * 256 + 2 * 256 * 256 + 256 * 217 + 4 * 256 * 256 + 256 * 1)  # 918 016
FLOP_PER_HIT = 2 * (524544 + 459008 + 818176)  # 3 603 456


def algorithmic_flop(n_evals: int, n_hits: int) -> int:
    return FLOP_PER_SDF_EVAL * n_evals + FLOP_PER_HIT * n_hits
