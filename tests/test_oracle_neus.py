"""Pins oracle/neus_ref.py (the stage-1 NeuS volume renderer, BASELINE config C2) to goldens recorded
from the real reference (tests/golden/make_golden_neus.py).  CPU only.  There is no product code for this row yet: these
tests fix the target the HIP build of the row will be held to."""
import numpy as np
import torch

from oracle import iron_ref as R
from oracle import neus_ref as N
from iron_amd.fields import RenderingNetwork, SDFNetwork

from _util import cpu_sd, golden, golden_meta, state_hash, t


def _stage1():
    """Same construction order / seed as make_golden_neus.build_stage1, with the build's own constructors."""
    torch.manual_seed(0)
    nets = {
        "sdf_network": SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5, scale=1.0,
                                  geometric_init=True, weight_norm=True),
        "color_network": RenderingNetwork(d_feature=256, mode="idr", d_in=9, d_out=3, d_hidden=256, n_layers=8, skip_in=[4],
                                          weight_norm=True, multires=10, multires_view=4, squeeze_out=True),
        "nerf": N.NerfParams(),
        "deviation_network": N.VarianceParams(0.3),
    }
    return nets


def test_stage1_constructors_reproduce_reference_state():
    assert state_hash(_stage1()) == golden_meta()["state_sha256_stage1"]


def test_g13_nerf_forward_and_sample_pdf():
    g = golden("g13_neus.npz")
    nets = _stage1()
    alpha, rgb = N.nerf_forward(cpu_sd(nets["nerf"]), N.NerfSpec(), t(g["nerf_pts"]), t(g["nerf_views"]))
    np.testing.assert_allclose(alpha.numpy(), g["nerf_alpha"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rgb.numpy(), g["nerf_rgb"], rtol=1e-5, atol=1e-6)
    c = R.rendering_forward(cpu_sd(nets["color_network"]), N.COLOR_SPEC, t(g["color_pts"]), t(g["color_nrm"]), t(g["color_view"]),
                            t(g["color_feat"]))
    np.testing.assert_allclose(c.numpy(), g["color_out"], rtol=1e-5, atol=1e-6)
    s = N.sample_pdf(t(g["pdf_bins"]), t(g["pdf_weights"]), 16, det=True)
    np.testing.assert_allclose(s.numpy(), g["pdf_samples"], rtol=1e-6, atol=1e-7)


def test_g13_neus_render():
    """NeuSRenderer.render (renderer.py:346-453): 64 + 4 x 16 hierarchical samples, NeRF background with 32 outside
    samples, logistic-CDF alpha, compositing -- on 96 rays."""
    g = golden("g13_neus.npz")
    nets = _stage1()
    sc = N.NeusScene(cpu_sd(nets["sdf_network"]), R.SDFSpec(), cpu_sd(nets["color_network"]), cpu_sd(nets["nerf"]),
                     nets["deviation_network"].variance.detach().clone())
    torch.set_num_threads(8)
    out = N.render(sc, t(g["rays_o"]), t(g["rays_d"]), t(g["near"]), t(g["far"]), background_rgb=None,
                   cos_anneal_ratio=float(g["cos_anneal_ratio"]))
    for k in ("color_fine", "s_val", "cdf_fine", "weight_sum", "weight_max", "gradients", "weights", "gradient_error",
              "inside_sphere"):
        assert tuple(out[k].shape) == g[k].shape, k
        np.testing.assert_allclose(out[k].numpy(), g[k], rtol=2e-5, atol=2e-6, err_msg=k)
    assert out["weights"].shape[1] == 64 + 64 + 32


def test_g16_neus_training_gradients():
    """render under autograd (render_volume.py:160-200): the oracle's render_train + torch.autograd vs the real reference's
    gradients of all 79 parameter tensors of the four stage-1 networks."""
    from oracle import train_ref as T
    g = golden("g16_neus_train.npz")
    g13 = golden("g13_neus.npz")
    nets = _stage1()
    sd = {k: T.leaf_state(cpu_sd(nets[k])) for k in ("sdf_network", "color_network", "nerf")}
    var = nets["deviation_network"].variance.detach().clone().requires_grad_(True)
    sc = N.NeusScene(sd["sdf_network"], R.SDFSpec(), sd["color_network"], sd["nerf"], var)
    torch.set_num_threads(8)
    out = N.render_train(sc, t(g13["rays_o"]), t(g13["rays_d"]), t(g13["near"]), t(g13["far"]), background_rgb=None, cos_anneal_ratio=0.3)
    np.testing.assert_allclose(out["color_fine"].detach().numpy(), g["color_fine"], rtol=2e-5, atol=2e-6)
    loss = (out["color_fine"] * t(g["loss_wc"])).sum() + 0.1 * out["gradient_error"] + (out["weight_sum"] * t(g["loss_ww"])).sum()
    assert abs(loss.item() - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    loss.backward()
    n = 0
    grads = {("deviation_network", "variance"): var.grad}
    for name in sd:
        for pname, p in sd[name].items():
            if p.is_floating_point():
                grads[(name, pname)] = p.grad
    for (name, pname), gr in grads.items():
        key = "%s/%s" % (name, pname)
        assert gr is not None, key
        gr = gr.reshape(-1).double().numpy()
        ref_n = float(g["gnorm:" + key])
        assert abs(np.linalg.norm(gr) - ref_n) <= 5e-4 * max(ref_n, 1e-12), (key, np.linalg.norm(gr), ref_n)
        idx = np.concatenate([np.arange(min(16, gr.size)), np.linspace(0, gr.size - 1, 32).astype(np.int64)])
        ref_s = g["gsample:" + key]
        np.testing.assert_allclose(gr[idx], ref_s, rtol=5e-3, atol=5e-5 * max(np.abs(ref_s).max(), 1e-12), err_msg=key)
        n += 1
    assert n == golden_meta()["n_param_tensors_neus_train_golden"]
