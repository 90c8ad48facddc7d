"""The `ggx` branch of the reference's network factory (models/network_conf.py:16-44, 72-122,
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
        self.light.data.fill_(light)

    def get_light(self):
        return self.light.data.clone().detach()


def init_sdf_network_dict(device="cuda"):
    """models/network_conf.py:31-44."""
    return SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5, scale=1.0,
                      geometric_init=True, weight_norm=True).to(device)


def init_rendering_network_dict(renderer_name="ggx", device="cuda"):
    """models/network_conf.py:47-122, `ggx` branch (the fork's other branches are out of scope)."""
    if renderer_name != "ggx":
        raise NotImplementedError("only the 'ggx' renderer is built (SURVEY 8 row f-4 covers the others)")
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


def choose_renderer(renderer_name="ggx"):
    """models/network_conf.py:748-764."""
    from .renderer_ggx import GGXColocatedRenderer

    if renderer_name != "ggx":
        raise NotImplementedError("only the 'ggx' renderer is built")
    return GGXColocatedRenderer(use_cuda=True)
