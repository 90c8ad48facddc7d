"""Synthetic scenes / cameras for tests and bench: no dataset or checkpoint exists
offline, so the "drv/dragon" configs are restated as seeded networks.

S0 "sphere": seed 0, geometric init (SDF ~ sphere r 0.5), `ggx` material nets, light 8*2^2 = 32
              (render_surface.py:353-355 rule at camera distance 2).
S1 "bumpy" : S0 + N(0, 0.01^2) on lin0.weight_v[:, 3:] (seed 1) so PE channels shape the surface
              (SURVEY proposed sigma 0.05, which leaves no zero level set; 0.01 gives ~34 % hits).
S3 "trained-like": S1 with every hidden unit of every network rescaled (unit i of layer l by r_i, the next layer's column i
              by 1/r_i; r log-uniform in [1/8, 32]) and the 256 feature outputs of the SDF net by s_j in [0.1, 10] (undone in
              the material nets' feature columns): hidden activations reach the tens, weight_g rows span 250:1, folded
              weights span 1e-4 ... 30 -- the dynamic range of a trained checkpoint, which none exists of offline -- while
              the material nets' functions stay S1's exactly (ReLU is homogeneous); the SDF becomes a bumpy blob of radius
              ~0.5 with gradient norm ~0.5 (softplus(beta=100) is not homogeneous at its knee; the output bias is re-centred).
Cameras    : the reference fixture camera (tests/data_singleview/cam_dict_norm.json) rescaled.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

from .network_conf import PointLightNetwork
from .fields import RenderingNetwork, SDFNetwork

# tests/data_singleview/cam_dict_norm.json of the reference (512x512)
FIXTURE_K = [[811.9282694049824, 0.0, 256.0, 0.0], [0.0, 811.9282694049824, 256.0, 0.0],
             [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]]
FIXTURE_W2C = [[0.998867339183008, 0.0, -0.04758191582374219, 1.5416074755814572e-17],
               [-0.013163727354886733, -0.9609695958324571, -0.27634064516051604, 1.1553154537250536e-16],
               [-0.045724774418075535, 0.27665400030652737, -0.959881143224937, 2.0],
               [0.0, 0.0, 0.0, 1.0]]
FIXTURE_SIZE = 512


def build_networks(scene: str = "S0", seed: int = 0) -> Dict[str, torch.nn.Module]:
    """CPU-resident networks; construction order fixes the RNG stream (sdf, diffuse, specular, roughness)."""
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
        "point_light_network": PointLightNetwork(),
    }
    nets["point_light_network"].set_light(8.0 * 2.0 * 2.0)
    if scene == "S1":
        g = torch.Generator().manual_seed(1)
        v = nets["sdf_network"].lin0.weight_v
        with torch.no_grad():
            v[:, 3:] += 0.01 * torch.randn(v[:, 3:].shape, generator=g)
    elif scene == "S3":
        g = torch.Generator().manual_seed(1)
        v = nets["sdf_network"].lin0.weight_v
        with torch.no_grad():
            v[:, 3:] += 0.01 * torch.randn(v[:, 3:].shape, generator=g)
        rescale_hidden_units(nets, seed=3)
    elif scene != "S0":
        raise ValueError(scene)
    return nets


def _effective(lin) -> torch.Tensor:
    v, g = lin.weight_v.detach().double(), lin.weight_g.detach().double()
    return g * v / v.norm(dim=1, keepdim=True)


def _set_effective(lin, W: torch.Tensor, bias: torch.Tensor) -> None:
    """weight_v := W, weight_g := its row norms (so that g v / |v| = W), bias := bias."""
    with torch.no_grad():
        lin.weight_v.copy_(W.to(lin.weight_v.dtype))
        lin.weight_g.copy_(W.norm(dim=1, keepdim=True).to(lin.weight_g.dtype))
        lin.bias.copy_(bias.to(lin.bias.dtype))


def rescale_hidden_units(nets, seed: int = 3, lo: float = 1.0 / 8.0, hi: float = 32.0) -> None:
    """Scene S3's transformation, in place, on any networks with the reference's parameter names (lin{l}.weight_g / weight_v /
    bias): works on iron_amd's classes and on the reference's alike (tests/golden/make_golden_s3.py applies it to the latter).
    Arithmetic in float64 on the CPU so that both sides end with bit-identical parameters."""
    gen = torch.Generator().manual_seed(seed)

    def draw(n, a, b):
        return torch.exp(torch.rand(n, generator=gen, dtype=torch.float64) * (math.log(b) - math.log(a)) + math.log(a))

    sdf = nets["sdf_network"]
    n_lin = sdf.num_layers - 1
    W = [_effective(getattr(sdf, "lin%d" % l)) for l in range(n_lin)]
    B = [getattr(sdf, "lin%d" % l).bias.detach().double() for l in range(n_lin)]
    for l in range(n_lin - 1):                       # hidden layers: outputs of lin_l feed lin_{l+1}
        r = draw(W[l].shape[0], lo, hi)
        W[l], B[l] = W[l] * r[:, None], B[l] * r
        W[l + 1][:, : r.shape[0]] = W[l + 1][:, : r.shape[0]] / r[None, :]   # (the skip layer's extra 39 input columns stay)
    s = draw(W[-1].shape[0] - 1, 0.1, 10.0)          # feature outputs 1..256 of the last layer; row 0 is the distance
    W[-1][1:], B[-1][1:] = W[-1][1:] * s[:, None], B[-1][1:] * s
    # softplus(beta=100) is not homogeneous around its knee, so the rescaled network is not S1's function any more: its
    # distance still grows with the radius but is offset (0.36 on the r = 0.5 sphere); the output bias puts the zero level set back
    B[-1][0] = B[-1][0] - 0.36
    for l in range(n_lin):
        _set_effective(getattr(sdf, "lin%d" % l), W[l], B[l])
    for name, net in nets.items():
        if name == "sdf_network" or not hasattr(net, "num_layers") or not hasattr(net, "lin0"):
            continue
        n_lin = net.num_layers - 1
        W = [_effective(getattr(net, "lin%d" % l)) for l in range(n_lin)]
        B = [getattr(net, "lin%d" % l).bias.detach().double() for l in range(n_lin)]
        W[0][:, -s.shape[0]:] = W[0][:, -s.shape[0]:] / s[None, :]            # the feature columns are the last 256 inputs
        for l in range(n_lin - 1):
            r = draw(W[l].shape[0], lo, hi)
            W[l], B[l] = W[l] * r[:, None], B[l] * r
            W[l + 1] = W[l + 1] / r[None, :]
        for l in range(n_lin):
            _set_effective(getattr(net, "lin%d" % l), W[l], B[l])


COMP_ORDER = ("diffuse_albedo_network", "specular_albedo_network", "specular_roughness_network", "metallic_network",
              "dielectric_network", "metallic_eta_network", "metallic_k_network", "dielectric_eta_network")


def build_comp_networks(seed: int = 0) -> Dict[str, torch.nn.Module]:
    """Scene S2: the seed-0 SDF network of S0, then the `comp2` material networks of
    models/network_conf.py:318-447 in COMP_ORDER (the construction order fixes the RNG stream)."""
    from .network_conf import comp_material_network
    torch.manual_seed(seed)
    nets = {"sdf_network": SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5,
                                      scale=1.0, geometric_init=True, weight_norm=True)}
    for name in COMP_ORDER:
        nets[name] = comp_material_network(name)
    nets["point_light_network"] = PointLightNetwork()
    nets["point_light_network"].set_light(8.0 * 2.0 * 2.0)
    return nets


def fixture_camera_matrices(width: int, height: int, yaw_deg: float = 0.0):
    """K, W2C (4x4 fp32) of the fixture camera rescaled to width x height, optionally orbited about
    world Y by yaw_deg (C4's 8 views = k*45 degrees)."""
    K = torch.tensor(FIXTURE_K, dtype=torch.float32)
    K[0, :3] *= width / FIXTURE_SIZE
    K[1, :3] *= height / FIXTURE_SIZE
    W2C = torch.tensor(FIXTURE_W2C, dtype=torch.float64)
    if yaw_deg != 0.0:
        a = math.radians(yaw_deg)
        R = torch.tensor([[math.cos(a), 0, math.sin(a), 0], [0, 1, 0, 0], [-math.sin(a), 0, math.cos(a), 0],
                          [0, 0, 0, 1]], dtype=torch.float64)
        W2C = W2C @ R
    return K, W2C.float()
