"""GPU: the numeric envelope of the default (split-fp16, "h2") core (csrc/envelope.hip, include/iron_hip.h).

The reference is plain fp32 (models/fields.py:82-98, 203-239): hidden activations or features above fp16's range (65 504) are
ordinary numbers to it.  The h2 core splits every operand into two fp16 pieces, so such a value overflows; the guard must (a) make
the overflowing call LOUD (non-finite values, never plausible ones), (b) raise the network's flag so that every later call runs on
the exact-fp32 MFMA core and matches the fp32 reference, (c) with IRON_H2_OVERFLOW=rerun semantics return the right values at once."""
import copy

import numpy as np
import pytest
import torch

from iron_amd import scenes

from _util import cpu_sd, oracle_scene

pytestmark = pytest.mark.gpu


BLOW_UP = 1.8e5   # folded |W_0| max 0.306 -> 5.5e4 (still an fp16 number: the h2 stream is built), z_0 max 0.477 -> 8.6e4 (not one)


def _big_activation_sdf():
    """S1's SDF net with layer 0 scaled by 1.8e5 and layer 1's weights scaled back: the largest hidden activations of layer 0
    reach 8.6e4 > 65 504 (softplus is the identity there) while every folded weight stays inside fp16's range, every later layer
    is in its usual range, and the reference's fp32 arithmetic has no trouble with any of it."""
    nets = scenes.build_networks("S1")
    net = nets["sdf_network"]
    with torch.no_grad():
        net.lin0.weight_g.mul_(BLOW_UP)
        net.lin0.bias.mul_(BLOW_UP)
        net.lin1.weight_g.mul_(1.0 / BLOW_UP)   # weight_norm: W = g v / |v| is scale-free in v, the scale goes on g
    return net


@torch.no_grad()
def test_hidden_activations_above_fp16_range_fall_back_to_the_exact_core():
    from oracle import iron_ref as R
    net = _big_activation_sdf()
    ref_sd = cpu_sd(net)
    g = torch.Generator().manual_seed(3)
    x = torch.rand(4096, 3, generator=g) * 1.6 - 0.8
    want = R.sdf_forward(ref_sd, R.SDFSpec(), x)[:, 0]          # the reference's fp32 arithmetic (torch CPU)
    assert torch.isfinite(want).all()
    net = net.cuda()
    xc = x.cuda()
    st0 = net.numeric_status()
    assert st0 == {"overflow_seen": False, "exact_core": False, "pending": False}
    first = net.sdf(xc)[:, 0]
    st1 = net.numeric_status()
    assert st1["pending"], "activations of 8.6e4 went through the fp16 split unnoticed"
    assert not torch.isfinite(first).all(), "the overflowing call must be loud"
    second = net.sdf(xc)[:, 0]
    st2 = net.numeric_status()
    assert st2["overflow_seen"] and st2["exact_core"] and not st2["pending"]
    err = float((second.cpu() - want).abs().max())
    scale = float(want.abs().max())
    print("   exact core after the overflow: max|d| vs the fp32 reference %.2e (scale %.2e)" % (err, scale))
    assert torch.isfinite(second).all() and err <= 2e-5 * max(1.0, scale)
    # everything on this handle is exact now: get_all (forward-mode kernels of the f32 core) and the tracer's evaluation
    sdf3, feat3, grad3 = net.get_all(xc, is_training=False)
    assert torch.isfinite(sdf3).all() and torch.isfinite(grad3).all() and torch.isfinite(feat3).all()
    assert float((sdf3[:, 0].cpu() - want).abs().max()) <= 2e-5 * max(1.0, scale)
    # re-packing (a parameter write) starts from the default core again; force_exact() survives it
    net.force_exact(True)
    net.lin8.bias.add_(0.0)
    assert net.numeric_status()["exact_core"]
    assert torch.isfinite(net.sdf(xc)).all()


@torch.no_grad()
def test_rerun_mode_returns_the_right_values_at_once(monkeypatch):
    from oracle import iron_ref as R
    import iron_amd.fields as F
    net = _big_activation_sdf()
    ref_sd = cpu_sd(net)
    x = torch.rand(1000, 3, generator=torch.Generator().manual_seed(4)) * 1.6 - 0.8
    want = R.sdf_forward(ref_sd, R.SDFSpec(), x)[:, 0]
    monkeypatch.setattr(F, "_OVERFLOW_RERUN", True)
    net = net.cuda()
    got = net.sdf(x.cuda())[:, 0].cpu()
    assert torch.isfinite(got).all()
    assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
    assert net.numeric_status()["exact_core"]


@torch.no_grad()
def test_features_of_1e5_into_a_material_net():
    """RenderingNetwork.forward with feature vectors of magnitude ~1e5 (fields.py:203-239 takes any fp32 feature): the h2 kernel's
    split of the features overflows; second call == the exact core == a torch fp32 evaluation."""
    from oracle import iron_ref as R
    nets = scenes.build_networks("S1")
    mat = nets["specular_albedo_network"]          # no_view_dir, PE-6 points, linear output head (no sigmoid hiding an inf)
    g = torch.Generator().manual_seed(5)
    n = 3000
    pts = torch.rand(n, 3, generator=g) - 0.5
    nrm = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    feat = torch.randn(n, 256, generator=g) * 1.0e5
    spec = R.GGX_SPECS["specular_albedo_network"]
    want = R.rendering_forward(cpu_sd(mat), spec, pts, nrm, None, feat)
    assert torch.isfinite(want).all()
    mat = mat.cuda()
    args = (pts.cuda(), nrm.cuda(), None, feat.cuda())
    first = mat(*args)
    assert mat.numeric_status()["pending"] and not torch.isfinite(first).all()
    second = mat(*args).cpu()
    assert mat.numeric_status()["exact_core"]
    rel = float((second - want).norm() / want.norm())
    print("   material net on 1e5 features, exact core vs torch fp32: rel-L2 %.2e" % rel)
    assert rel <= 1e-5
    # in-range features on a fresh handle stay on the default core
    mat2 = scenes.build_networks("S1")["specular_albedo_network"].cuda()
    mat2(pts.cuda(), nrm.cuda(), None, (feat * 1e-5).cuda())
    assert mat2.numeric_status() == {"overflow_seen": False, "exact_core": False, "pending": False}


@torch.no_grad()
def test_a_normal_frame_raises_no_flag():
    from iron_amd.raytracer import Camera, RayTracer, render_camera
    from iron_amd.renderer_ggx import GGXColocatedRenderer
    from iron_amd.rendering_func import make_render_fn
    nets = {k: v.cuda() for k, v in scenes.build_networks("S3").items()}   # the trained-like dynamic range
    K, W2C = scenes.fixture_camera_matrices(128, 128)
    cam = Camera(128, 128, K.cuda(), W2C.cuda())
    render_camera(cam, nets["sdf_network"], RayTracer(), nets, make_render_fn(GGXColocatedRenderer(use_cuda=True)), fill_holes=True,
                  handle_edges=True)
    for k in ("sdf_network", "diffuse_albedo_network", "specular_albedo_network", "specular_roughness_network"):
        assert nets[k].numeric_status() == {"overflow_seen": False, "exact_core": False, "pending": False}, k
