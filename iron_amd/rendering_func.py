"""Material query + the stage-2 render_fn, with the reference's surface:
    get_materials            <- models/rendering_func.py:5-16
    get_materials_comp       <- models/rendering_func.py:19-49
    get_materials_multi      <- models/rendering_func.py:50-63
    make_render_fn(renderer) <- the driver's render_fn closure, render_surface.py:117-156
    make_render_fn_comp(renderer) <- render_fn_comp, render_surface.py:159-234

`make_render_fn` returns a callable with the reference render_fn signature
(interior_mask, color_network_dict, ray_o, ray_d, points, normals, features) -> dict.  When
iron_amd.raytracer.render_normal_and_color sees it, it uses the attached fused path instead
(one `iron_shade_ggx` launch: get_all + normalise + 3 material MLPs + GGX + scatter); called
directly it runs the same steps through the per-operator HIP entry points.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from . import _lib


def get_materials(network_dict, points, normals, features, is_metal=False):
    """models/rendering_func.py:5-16."""
    diffuse_albedo = network_dict["diffuse_albedo_network"](points, normals, -normals, features).abs()
    specular_albedo = network_dict["specular_albedo_network"](points, normals, None, features).abs()
    if not is_metal:
        specular_albedo = torch.mean(specular_albedo, dim=-1, keepdim=True).expand_as(specular_albedo)
    specular_roughness = network_dict["specular_roughness_network"](points, normals, None, features).abs() + 0.01
    return {"diffuse_albedo": diffuse_albedo, "specular_albedo": specular_albedo,
            "specular_roughness": specular_roughness}


def get_materials_comp(network_dict, points, normals, features):
    """models/rendering_func.py:19-49."""
    def run(name, view):
        return network_dict[name](points, normals, view, features).abs()
    return {"diffuse_albedo": run("diffuse_albedo_network", -normals),
            "specular_albedo": run("specular_albedo_network", None),
            "metallic": run("metallic_network", None),
            "dielectric": run("dielectric_network", None),
            "specular_roughness": run("specular_roughness_network", None),
            "metallic_eta": run("metallic_eta_network", None),
            "metallic_k": run("metallic_k_network", None),
            "dielectric_eta": run("dielectric_eta_network", None)}


def get_materials_multi(color_network_dict, points, normals, features, is_metal=False):
    """models/rendering_func.py:50-63 (the `multi` network set: no channel-mean on the specular albedo, plus the 4-wide
    points-only material vector)."""
    return {"diffuse_albedo": color_network_dict["diffuse_albedo_network"](points, normals, -normals, features).abs(),
            "specular_albedo": color_network_dict["specular_albedo_network"](points, normals, None, features).abs(),
            "specular_roughness": color_network_dict["specular_roughness_network"](points, normals, None, features).abs() + 0.01,
            "material_vector": color_network_dict["material_network"](points, None, None, features).abs()}


class MaterialPredictor(torch.nn.Module):
    """render_surface.py:434-450: surface points -> (diffuse_albedo [n,3], specular_albedo [n,3], specular_roughness [n,1]),
    the callable export_materials (models/export_materials.py:165-203) samples 25 M times to splat its 2048^2 textures.
    get_all (value + analytic normal, one launch) -> normalise -> the material networks of the dictionary at hand: the
    composite set (get_materials_comp, what the reference's class hard-codes) when the dictionary has it, the ggx set
    (get_materials) otherwise."""

    def __init__(self, sdf_network, color_network_dict):
        super().__init__()
        self.sdf_network = sdf_network
        self.color_network_dict = color_network_dict

    @torch.no_grad()
    def forward(self, points):
        _, features, normals = self.sdf_network.get_all(points, is_training=False)
        normals = normals / (normals.norm(dim=-1, keepdim=True) + 1e-10)
        fn = get_materials_comp if "metallic_network" in self.color_network_dict else get_materials
        res = fn(self.color_network_dict, points, normals, features)
        return res["diffuse_albedo"], res["specular_albedo"], res["specular_roughness"]


@torch.no_grad()
def query_materials(material_predictor, points, max_num_pts=320000):
    """The bulk query loop of export_materials (models/export_materials.py:181-191): `points` [N,3] in splits of max_num_pts
    -> [N,7] = cat(diffuse_albedo, specular_albedo, specular_roughness), kept on the device (the reference moves every split
    to the host; the splat that follows is host-side numpy there and is not part of this build)."""
    pts = _lib.require_cuda_f32(points, "points").reshape(-1, 3)
    out = torch.empty((pts.shape[0], 7), dtype=torch.float32, device=pts.device)
    for start in range(0, pts.shape[0], int(max_num_pts)):
        kd, ks, rough = material_predictor(pts[start:start + int(max_num_pts)])
        rows = out[start:start + kd.shape[0]]
        rows[:, 0:3], rows[:, 3:6], rows[:, 6:7] = kd, ks, rough.reshape(-1, 1)
    return out


class CompRenderFn:
    """render_surface.py:159-234 (render_fn_comp) as a callable: get_materials_comp -> CompositeRenderer -> scatter.
    Every step runs through its HIP operator (8 material-network launches + one composite kernel); scalar maps keep
    their trailing [...,1] here and are squeezed by render_normal_and_color like in the reference."""

    _VEC = ("color", "diffuse_color", "specular_color", "diffuse_albedo", "specular_albedo", "metallic_rgb", "dielectric_rgb", "normal")
    _SCALAR = ("specular_roughness", "metallic_eta", "metallic_k", "dielectric_eta", "metallic", "dielectric")

    def __init__(self, renderer):
        self.renderer = renderer

    iron_takes_hit_index = True   # see GGXRenderFn.__call__

    def __call__(self, interior_mask, color_network_dict, ray_o, ray_d, points, normals, features, hit_index=None):
        dots_sh = list(interior_mask.shape)
        dev = interior_mask.device
        rgb = torch.zeros(dots_sh + [3], dtype=torch.float32, device=dev)
        out = {k: rgb.clone() for k in self._VEC}
        for k in self._SCALAR:
            out[k] = rgb[..., 0:1].clone()
        if points.shape[0] > 0:   # the rows handed over are the mask's hits (render_surface.py:159-170)
            idx = hit_index if hit_index is not None else interior_mask.reshape(-1).nonzero(as_tuple=True)[0]
            normals = normals / (normals.norm(dim=-1, keepdim=True) + 1e-10)
            params = get_materials_comp(color_network_dict, points, normals, features)
            light = color_network_dict["point_light_network"]()  # a Parameter: trainable under is_training
            res = self.renderer(light, (points - ray_o).norm(dim=-1, keepdim=True), normals, -ray_d, params=params)
            values = {"color": res["rgb"], "diffuse_color": res["diffuse_rgb"], "specular_color": res["specular_rgb"],
                      "metallic_rgb": res["metallic_rgb"], "dielectric_rgb": res["dielectric_rgb"], "normal": normals}
            for k in ("diffuse_albedo", "specular_albedo") + self._SCALAR:
                values[k] = params[k]
            for k, v in values.items():   # one list of hit positions for all fourteen scatters
                out[k].view(-1, out[k].shape[-1])[idx] = v
        return out


    # -- fused path used by render_normal_and_color (same hook name as the GGX render_fn) ------------------------
    def iron_fused_ggx(self, results: Dict[str, torch.Tensor], sdf_network, color_network_dict) -> Dict[str, torch.Tensor]:
        pts = _lib.require_cuda_f32(results["points"], "points").reshape(-1, 3)
        ray_o = _lib.require_cuda_f32(results["ray_o"], "ray_o").reshape(-1, 3)
        ray_d = _lib.require_cuda_f32(results["ray_d"], "ray_d").reshape(-1, 3)
        conv = results["convergent_mask"].reshape(-1).contiguous()
        n = pts.shape[0]
        dev = pts.device
        lib = _lib.load()
        nets = _lib.iron_shade_comp_nets()
        keep = [sdf_network.hip_net()]
        nets.sdf = keep[0].handle
        for k in ("diffuse_albedo", "specular_albedo", "specular_roughness", "metallic", "dielectric", "metallic_eta", "metallic_k",
                  "dielectric_eta"):
            keep.append(color_network_dict[k + "_network"].hip_net())
            setattr(nets, k, keep[-1].handle)
        bufs = {k: torch.empty((n, 3) if k in self._VEC else (n, 1), dtype=torch.float32, device=dev) for k in _lib.COMP_OUT_FIELDS}
        so = _lib.iron_shade_comp_out()
        for k in _lib.COMP_OUT_FIELDS:
            setattr(so, k, bufs[k].data_ptr())
        t1, t2 = self.renderer._tables_on(dev)
        ws_bytes = lib.iron_shade_composite_workspace_bytes(n)
        ws = _lib.workspace(ws_bytes, dev, "shade")
        light = _host_light(color_network_dict["point_light_network"])
        with torch.cuda.device(dev):
            _lib.check(lib.iron_shade_composite(C.byref(nets), light, t1.data_ptr(), t2.data_ptr(), ray_o.data_ptr(), ray_d.data_ptr(),
                                                pts.data_ptr(), conv.data_ptr(), n, C.byref(so), ws.data_ptr(), ws_bytes,
                                                _lib.stream_ptr(dev)))
        return bufs


def make_render_fn_comp(renderer) -> CompRenderFn:
    return CompRenderFn(renderer)


_OUT_KEYS = ("color", "diffuse_color", "specular_color", "diffuse_albedo", "specular_albedo", "specular_roughness",
             "normal")


class GGXRenderFn:
    """render_surface.py:117-156 as a callable object."""

    def __init__(self, renderer, is_metal: bool = False):
        self.renderer = renderer
        self.is_metal = is_metal

    # -- generic path: same steps as the reference closure, each through its HIP operator -----------
    iron_takes_hit_index = True   # render_normal_and_color hands over the hit positions it has already listed

    def __call__(self, interior_mask, color_network_dict, ray_o, ray_d, points, normals, features, hit_index=None):
        """`hit_index` (optional, beyond the reference's signature): int64 positions of the True entries of the flattened mask.  Every
        `x[mask] = v` lists the mask again (a nonzero + a host sync each); the seven scatters below share one list instead."""
        dots_sh = list(interior_mask.shape)
        dev = interior_mask.device
        rgb = torch.zeros(dots_sh + [3], dtype=torch.float32, device=dev)
        out = {k: rgb.clone() for k in ("color", "diffuse_color", "specular_color", "diffuse_albedo",
                                        "specular_albedo", "normal")}
        out["specular_roughness"] = rgb[..., 0].clone()
        if points.shape[0] > 0:   # the rows handed over ARE the mask's hits (render_surface.py:117-126): no device query needed
            idx = hit_index if hit_index is not None else interior_mask.reshape(-1).nonzero(as_tuple=True)[0]
            normals = normals / (normals.norm(dim=-1, keepdim=True) + 1e-10)
            params = get_materials(color_network_dict, points, normals, features, is_metal=self.is_metal)
            light = color_network_dict["point_light_network"]()  # a Parameter: trainable under is_training
            res = self.renderer(light, (points - ray_o).norm(dim=-1, keepdim=True), normals, -ray_d, params=params)
            for key, val in (("color", res["rgb"]), ("diffuse_color", res["diffuse_rgb"]), ("specular_color", res["specular_rgb"]),
                             ("diffuse_albedo", params["diffuse_albedo"]), ("specular_albedo", params["specular_albedo"]),
                             ("normal", normals)):
                out[key].view(-1, 3)[idx] = val
            out["specular_roughness"].view(-1)[idx] = params["specular_roughness"].squeeze(-1)
        return out

    # -- fused path used by render_normal_and_color ---------------------------------------------------
    def iron_fused_ggx(self, results: Dict[str, torch.Tensor], sdf_network, color_network_dict) -> Dict[str, torch.Tensor]:
        pts = _lib.require_cuda_f32(results["points"], "points").reshape(-1, 3)
        ray_o = _lib.require_cuda_f32(results["ray_o"], "ray_o").reshape(-1, 3)
        ray_d = _lib.require_cuda_f32(results["ray_d"], "ray_d").reshape(-1, 3)
        conv = results["convergent_mask"].reshape(-1).contiguous()
        n = pts.shape[0]
        dev = pts.device
        lib = _lib.load()
        nets = _lib.iron_shade_nets()
        keep = [sdf_network.hip_net(), color_network_dict["diffuse_albedo_network"].hip_net(),
                color_network_dict["specular_albedo_network"].hip_net(),
                color_network_dict["specular_roughness_network"].hip_net()]
        nets.sdf, nets.diffuse_albedo, nets.specular_albedo, nets.specular_roughness = [k.handle for k in keep]
        bufs = {k: torch.empty((n,) if k == "specular_roughness" else (n, 3), dtype=torch.float32, device=dev)
                for k in _OUT_KEYS}
        so = _lib.iron_shade_out()
        for k in _OUT_KEYS:
            setattr(so, k, bufs[k].data_ptr())
        t1, t2 = self.renderer._tables_on(dev)
        ws_bytes = lib.iron_shade_workspace_bytes(n)
        ws = _lib.workspace(ws_bytes, dev, "shade")
        light = _host_light(color_network_dict["point_light_network"])
        with torch.cuda.device(dev):
            _lib.check(lib.iron_shade_ggx(C.byref(nets), light, 1 if self.is_metal else 0, t1.data_ptr(), t2.data_ptr(),
                                          ray_o.data_ptr(), ray_d.data_ptr(), pts.data_ptr(), conv.data_ptr(), n,
                                          C.byref(so), ws.data_ptr(), ws_bytes, _lib.stream_ptr(dev)))
        return bufs


def _host_light(module) -> float:
    """The light scalar by value: cached per parameter version when the module offers it (network_conf.PointLightNetwork), read from
    the device otherwise (any callable returning a 0-dim tensor, as the reference's)."""
    cached = getattr(module, "host_light", None)
    return cached() if cached is not None else float(module().detach())


def make_render_fn(renderer, is_metal: bool = False) -> GGXRenderFn:
    return GGXRenderFn(renderer, is_metal=is_metal)
