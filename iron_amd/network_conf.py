"""The `ggx` and `comp2` branches of the reference's network factory (models/network_conf.py:16-44, 72-122, 318-447,
748-764), restated for this path: network shapes are the spec the HIP kernels are built for.
"""
from __future__ import annotations

import torch
from torch import nn

from .fields import RenderingNetwork, SDFNetwork


class PointLightNetwork(nn.Module):
    """models/network_conf.py:16-28: one learnable scalar light intensity."""

    def __init__(self):
        super().__init__()
        self.register_parameter("light", nn.Parameter(torch.tensor(5.0)))

    def forward(self):
        return self.light

    def set_light(self, light):
        with torch.no_grad():
            self.light.fill_(light)   # in place on the parameter (not through .data): the version counter moves, host_light() sees it

    def host_light(self) -> float:
        """The scalar as a host float, read from the device once per (storage, version) of the parameter: the fused inference
        shading passes it by value, and reading it every frame is the frame's one host sync (it stalls the launch queue behind the
        whole trace).  A write through `.data` does not move the version counter: call invalidate() after one."""
        key = (self.light.data_ptr(), self.light._version, str(self.light.device))
        if getattr(self, "_host_light_key", None) != key:
            self._host_light_value = float(self.light.detach())
            self._host_light_key = key
        return self._host_light_value

    def invalidate(self) -> None:
        self._host_light_key = None

    def get_light(self):
        return self.light.data.clone().detach()


def init_sdf_network_dict(device="cuda"):
    """models/network_conf.py:31-44."""
    return SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5, scale=1.0,
                      geometric_init=True, weight_norm=True).to(device)


COMP_NETWORKS = ("diffuse_albedo_network", "specular_albedo_network", "specular_roughness_network", "metallic_network",
                 "dielectric_network", "metallic_eta_network", "metallic_k_network", "dielectric_eta_network")


def comp_material_network(name: str) -> RenderingNetwork:
    """One material network of the `comp2` branch (models/network_conf.py:330-445): the diffuse head is the IDR-style
    net of the ggx branch; every other head is a no_view_dir PE-6 net with output_scale 1 and bias 0.1 (albedo: 0)."""
    if name == "diffuse_albedo_network":
        return RenderingNetwork(d_in=9, d_out=3, d_feature=256, d_hidden=256, n_layers=4, multires_view=4, mode="idr",
                                squeeze_out=True)
    if name not in COMP_NETWORKS:
        raise KeyError(name)
    d_out, bias = (3, 0.0) if name == "specular_albedo_network" else (1, 0.1)
    return RenderingNetwork(d_in=6, d_out=d_out, d_feature=256, d_hidden=256, n_layers=4, multires=6, multires_view=-1,
                            mode="no_view_dir", squeeze_out=False, output_bias=bias, output_scale=1.0)


def comp_env_light_network() -> RenderingNetwork:
    """comp2's env_light_network (models/network_conf.py:367-378): a points_only PE-6 head with one output."""
    return RenderingNetwork(d_in=3, d_out=1, d_feature=256, d_hidden=256, n_layers=4, multires=6, multires_view=-1,
                            mode="points_only", squeeze_out=False, output_bias=0.0, output_scale=1.0)


def init_rendering_network_dict(renderer_name="comp", device="cuda"):
    """models/network_conf.py:47-122 (`ggx`) and :318-447 (`comp2`; `comp`, which render_surface.py:107 asks for, is not
    defined by the reference's factory and is served by the same shapes).  color_network of the comp2 dict (the
    stage-1 colour net, unused by the render path) is not built."""
    if renderer_name in ("comp", "comp2"):
        d = {name: comp_material_network(name).to(device) for name in COMP_NETWORKS}
        d["env_light_network"] = comp_env_light_network().to(device)
        d["point_light_network"] = PointLightNetwork().to(device)
        return d
    if renderer_name != "ggx":
        raise NotImplementedError("renderer %r: only 'ggx' and 'comp'/'comp2' are built" % renderer_name)
    return {
        "diffuse_albedo_network": RenderingNetwork(d_in=9, d_out=3, d_feature=256, d_hidden=256, n_layers=4,
                                                   multires_view=4, mode="idr", squeeze_out=True).to(device),
        "specular_albedo_network": RenderingNetwork(d_in=6, d_out=3, d_feature=256, d_hidden=256, n_layers=4,
                                                    multires=6, multires_view=-1, mode="no_view_dir",
                                                    squeeze_out=False, output_bias=0.4, output_scale=0.1).to(device),
        "specular_roughness_network": RenderingNetwork(d_in=6, d_out=1, d_feature=256, d_hidden=256, n_layers=4,
                                                       multires=6, multires_view=-1, mode="no_view_dir",
                                                       squeeze_out=False, output_bias=0.1, output_scale=0.1).to(device),
        "point_light_network": PointLightNetwork().to(device),
    }


def choose_renderer(renderer_name="comp"):
    """models/network_conf.py:748-764."""
    from .renderer_ggx import CompositeRenderer, GGXColocatedRenderer

    if renderer_name in ("comp", "comp2"):
        return CompositeRenderer(use_cuda=True)
    if renderer_name != "ggx":
        raise NotImplementedError("renderer %r: only 'ggx' and 'comp'/'comp2' are built" % renderer_name)
    return GGXColocatedRenderer(use_cuda=True)
