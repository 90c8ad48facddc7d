"""Golden vectors for SURVEY 8 row f-3 (stage-1 NeuS volume renderer, BASELINE config C2) -- BUILD CONTAINER ONLY.

The real reference's NeuSRenderer.render (models/renderer.py:346-453) with the networks of confs/womask_iron.conf
(8x256 SDF net, 8-layer PE-10 skip-4 colour net, NeRF background field with n_outside = 32, variance 0.3), perturb = 0,
on 96 fixture-camera rays; plus NeRF.forward and sample_pdf on their own.  `mcubes` (only used by extract_geometry) gets
an empty placeholder module like the others.  No product code exists for this row yet; the goldens pin
oracle/neus_ref.py (tests/test_oracle_neus.py).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_neus.py
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.modules.setdefault("mcubes", types.ModuleType("mcubes"))
import make_golden as MG  # noqa: E402

from models.fields import NeRF, RenderingNetwork, SDFNetwork, SingleVarianceNetwork  # noqa: E402  (reference)
from models.renderer import NeuSRenderer, sample_pdf  # noqa: E402

npf = MG.npf


def build_stage1(seed: int = 0):
    torch.manual_seed(seed)
    nets = {
        "sdf_network": SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5, scale=1.0,
                                  geometric_init=True, weight_norm=True),
        "color_network": RenderingNetwork(d_feature=256, mode="idr", d_in=9, d_out=3, d_hidden=256, n_layers=8, skip_in=[4],
                                          weight_norm=True, multires=10, multires_view=4, squeeze_out=True),
        "nerf": NeRF(D=8, d_in=4, d_in_view=3, W=256, multires=10, multires_view=4, output_ch=4, skips=[4], use_viewdirs=True),
        "deviation_network": SingleVarianceNetwork(0.3),
    }
    return nets


def train():
    """G16: the reference's NeuSRenderer.render under autograd (perturb = 0, as render_volume.py:160-200 forms its loss): colour,
    eikonal statistic and weight_sum in a fixed loss; gradient norm + 48 sampled entries of EVERY parameter tensor of the four
    stage-1 networks."""
    import make_golden_train as MT
    nets = build_stage1()
    cam = MG.fixture_camera(512, 512)
    uv = cam.get_uv()[16::44, 30::64].reshape(-1, 2).contiguous()
    rays_o, rays_d, _ = cam.get_rays(uv)
    a = torch.sum(rays_d ** 2, dim=-1, keepdim=True)
    b = 2.0 * torch.sum(rays_o * rays_d, dim=-1, keepdim=True)
    mid = 0.5 * (-b) / a
    near, far = mid - 1.0, mid + 1.0
    renderer = NeuSRenderer(nets["nerf"], nets["sdf_network"], nets["deviation_network"], nets["color_network"],
                            n_samples=64, n_importance=64, n_outside=32, up_sample_steps=4, perturb=0.0)
    out = renderer.render(rays_o, rays_d, near, far, perturb_overwrite=0, background_rgb=None, cos_anneal_ratio=0.3)
    gen = torch.Generator().manual_seed(35)
    wc, ww = torch.rand(96, 3, generator=gen) - 0.3, torch.rand(96, 1, generator=gen) - 0.5
    loss = (out["color_fine"] * wc).sum() + 0.1 * out["gradient_error"] + (out["weight_sum"] * ww).sum()
    loss.backward()
    g = {"loss_wc": npf(wc), "loss_ww": npf(ww), "loss": np.float64(loss.item()), "color_fine": npf(out["color_fine"].detach()),
         "weight_sum": npf(out["weight_sum"].detach()), "gradient_error": npf(out["gradient_error"].detach())}
    n_params = 0
    for name in sorted(nets):
        for pname, p in nets[name].named_parameters():
            assert p.grad is not None, (name, pname)
            gr = p.grad.reshape(-1).double().numpy()
            key = "%s/%s" % (name, pname)
            g["gnorm:" + key] = np.float64(np.linalg.norm(gr))
            g["gsample:" + key] = gr[MT.sample_idx(gr.size)]
            n_params += 1
    np.savez_compressed(os.path.join(HERE, "g16_neus_train.npz"), **g)
    meta_path = os.path.join(HERE, "meta.json")
    meta = json.load(open(meta_path))
    meta["n_param_tensors_neus_train_golden"] = n_params
    json.dump(meta, open(meta_path, "w"), indent=1, sort_keys=True)
    print("G16 loss", loss.item(), "param tensors", n_params)


def main():
    meta_path = os.path.join(HERE, "meta.json")
    meta = json.load(open(meta_path))
    nets = build_stage1()
    meta["state_sha256_stage1"] = MG.state_hash(nets)
    # rays: a 12 x 8 lattice over the fixture camera's 512^2 image (centre hits the object, the rim misses it)
    cam = MG.fixture_camera(512, 512)
    uv = cam.get_uv()[16::44, 30::64].reshape(-1, 2).contiguous()
    rays_o, rays_d, _ = cam.get_rays(uv)
    a = torch.sum(rays_d ** 2, dim=-1, keepdim=True)
    b = 2.0 * torch.sum(rays_o * rays_d, dim=-1, keepdim=True)
    mid = 0.5 * (-b) / a
    near, far = mid - 1.0, mid + 1.0   # models/dataset.py:335-343
    renderer = NeuSRenderer(nets["nerf"], nets["sdf_network"], nets["deviation_network"], nets["color_network"],
                            n_samples=64, n_importance=64, n_outside=32, up_sample_steps=4, perturb=0.0)
    out = renderer.render(rays_o, rays_d, near, far, perturb_overwrite=0, background_rgb=None, cos_anneal_ratio=0.3)
    g = {"rays_o": npf(rays_o), "rays_d": npf(rays_d), "near": npf(near), "far": npf(far), "cos_anneal_ratio": np.float32(0.3)}
    for k, v in out.items():
        g[k] = npf(v)
    # NeRF.forward and sample_pdf on their own
    gen = torch.Generator().manual_seed(21)
    p4 = torch.rand(128, 4, generator=gen) * 2 - 1
    vd = torch.nn.functional.normalize(torch.randn(128, 3, generator=gen), dim=-1)
    with torch.no_grad():
        alpha, rgb = nets["nerf"](p4, vd)
    g.update({"nerf_pts": npf(p4), "nerf_views": npf(vd), "nerf_alpha": npf(alpha), "nerf_rgb": npf(rgb)})
    # the colour net on its own (8 layers, PE-10 points, PE-4 views, skip at layer 4; models/fields.py:203-239)
    cp = torch.rand(160, 3, generator=gen) * 1.2 - 0.6
    cn = torch.randn(160, 3, generator=gen)
    cv = torch.nn.functional.normalize(torch.randn(160, 3, generator=gen), dim=-1)
    cf = torch.randn(160, 256, generator=gen) * 0.3
    with torch.no_grad():
        cout = nets["color_network"](cp, cn, cv, cf)
    g.update({"color_pts": npf(cp), "color_nrm": npf(cn), "color_view": npf(cv), "color_feat": npf(cf), "color_out": npf(cout)})
    bins = torch.sort(torch.rand(16, 65, generator=gen), dim=-1)[0]
    w = torch.rand(16, 64, generator=gen) ** 4
    g.update({"pdf_bins": npf(bins), "pdf_weights": npf(w), "pdf_samples": npf(sample_pdf(bins, w, 16, det=True))})
    np.savez_compressed(os.path.join(HERE, "g13_neus.npz"), **g)
    json.dump(meta, open(meta_path, "w"), indent=1, sort_keys=True)
    print("state", meta["state_sha256_stage1"][:16], "rays", rays_o.shape[0], "weight_sum range",
          float(out["weight_sum"].min()), float(out["weight_sum"].max()))


if __name__ == "__main__":
    if "--train" in sys.argv:
        train()
    else:
        main()
