"""Golden vectors for SURVEY 8 row f-2 (stage-2 training forward + backward, BASELINE config C3) -- BUILD CONTAINER ONLY.

The real reference's render_camera(is_training=True, fill_holes=False, handle_edges=False) on the 32x32 centre crop of
the fixture camera (scene S1), a fixed linear functional of the rendered colour as loss, torch.autograd through the
reference's graph (reparam_points, double-backward through the SDF MLP, material nets, GGX).  Stored: the training-mode
colour image, the loss, and for EVERY parameter tensor of the four networks its gradient norm and 48 sampled entries.

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_train.py
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

npf = MG.npf
NETS = ("sdf_network", "diffuse_albedo_network", "specular_albedo_network", "specular_roughness_network")


def sample_idx(n: int) -> np.ndarray:
    """48 fixed positions of a flattened tensor of n elements: the first 16, then 32 on an even stride."""
    head = np.arange(min(16, n))
    rest = np.linspace(0, n - 1, 32).astype(np.int64)
    return np.concatenate([head, rest])


def main():
    nets = MG.build_reference_networks("S1")
    cam512 = MG.fixture_camera(512, 512)
    cam, _, _ = cam512.crop_region(32, 32, ul_corner=(240, 240))
    fn = MG.make_render_fn(nets, GGXColocatedRenderer(use_cuda=False), torch.float32)
    res = render_camera(cam, nets["sdf_network"], RayTracer(), nets, fn, fill_holes=False, handle_edges=False, is_training=True)
    gen = torch.Generator().manual_seed(31)
    wt = torch.rand(32, 32, 3, generator=gen) - 0.3
    loss = (res["color"] * wt).sum() + 0.1 * (res["normal"] * wt).sum()
    loss.backward()
    out = {"K": npf(cam.K), "W2C": npf(cam.W2C), "W": np.int64(32), "H": np.int64(32), "loss_weights": npf(wt),
           "loss": np.float64(loss.item()), "color": npf(res["color"]), "normal": npf(res["normal"]),
           "convergent_mask": npf(res["convergent_mask"])}
    n_params = 0
    for name in NETS:
        for pname, p in nets[name].named_parameters():
            assert p.grad is not None, (name, pname)
            g = p.grad.reshape(-1).double().numpy()
            key = "%s/%s" % (name, pname)
            out["gnorm:" + key] = np.float64(np.linalg.norm(g))
            out["gsample:" + key] = g[sample_idx(g.size)]
            n_params += 1
    np.savez_compressed(os.path.join(HERE, "g14_train_S1_c32.npz"), **out)
    meta_path = os.path.join(HERE, "meta.json")
    meta = json.load(open(meta_path))
    meta["n_param_tensors_train_golden"] = n_params
    json.dump(meta, open(meta_path, "w"), indent=1, sort_keys=True)
    print("loss", loss.item(), "hits", int(res["convergent_mask"].sum()), "param tensors", n_params,
          "sdf lin0.weight_v grad norm", float(nets["sdf_network"].lin0.weight_v.grad.norm()))


if __name__ == "__main__":
    main()
