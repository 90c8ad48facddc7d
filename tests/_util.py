"""Shared helpers for the tests: golden loading, product-nets -> oracle Scene conversion, metrics."""
from __future__ import annotations

import hashlib
import json
import os

import numpy as np
import torch

from oracle import iron_ref as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name: str):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def golden_meta():
    return json.load(open(os.path.join(GOLDEN, "meta.json")))


def t(a) -> torch.Tensor:
    return torch.from_numpy(np.asarray(a).copy())


def state_hash(nets) -> str:
    h = hashlib.sha256()
    for name in sorted(nets):
        if name == "point_light_network":
            continue
        sd = nets[name].state_dict()
        for k in sorted(sd):
            h.update(k.encode())
            h.update(sd[k].detach().cpu().numpy().tobytes())
    return h.hexdigest()


def cpu_sd(module) -> dict:
    return {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}


def tables():
    from iron_amd.renderer_ggx import load_mts_tables
    a, b = load_mts_tables()
    return a, b


def oracle_scene(nets, light: float = 32.0) -> R.Scene:
    """Oracle Scene (plain CPU tensors) from the product's nn.Modules."""
    mt, md = tables()
    rn = {k: (cpu_sd(nets[k]), R.GGX_SPECS[k]) for k in R.GGX_SPECS}
    return R.Scene(cpu_sd(nets["sdf_network"]), R.SDFSpec(), rn, light, mt, md)


def rel_l2(a, b) -> float:
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
