"""Pins oracle/train_ref.py (stage-2 training forward + autograd backward, BASELINE config C3) to
goldens recorded from the real reference (tests/golden/make_golden_train.py).  CPU only; no product code for this row yet."""
import numpy as np
import torch

from oracle import iron_ref as R
from oracle import train_ref as T
from iron_amd import scenes

from _util import cpu_sd, golden, golden_meta, t, tables

NETS = ("sdf_network", "diffuse_albedo_network", "specular_albedo_network", "specular_roughness_network")


def _sample_idx(n):
    return np.concatenate([np.arange(min(16, n)), np.linspace(0, n - 1, 32).astype(np.int64)])


def _check_grads(sd, g):
    n = 0
    for name in NETS:
        for pname, p in sd[name].items():
            if not p.is_floating_point():
                continue
            key = "%s/%s" % (name, pname)
            assert p.grad is not None, key
            gr = p.grad.reshape(-1).double().numpy()
            ref_n = float(g["gnorm:" + key])
            assert abs(np.linalg.norm(gr) - ref_n) <= 2e-4 * max(ref_n, 1e-12), (key, np.linalg.norm(gr), ref_n)
            ref_s = g["gsample:" + key]
            np.testing.assert_allclose(gr[_sample_idx(gr.size)], ref_s, rtol=2e-3, atol=2e-5 * max(np.abs(ref_s).max(), 1e-12), err_msg=key)
            n += 1
    return n


def test_g15_training_with_edge_sampling():
    """render_camera(handle_edges=True, is_training=True) behind the fixture's depth-edge mask: 122 edge pixels whose blend
    weights and side colours are in the graph."""
    g = golden("g15_train_edges_S1.npz")
    nets = scenes.build_networks("S1")
    mt, md = tables()
    sd = {k: T.leaf_state(cpu_sd(nets[k])) for k in NETS}
    sc = R.Scene(sd["sdf_network"], R.SDFSpec(), {k: (sd[k], R.GGX_SPECS[k]) for k in R.GGX_SPECS}, golden_meta()["light"], mt, md)
    cam = R.CameraSpec(int(g["W"]), int(g["H"]), t(g["K"]), t(g["W2C"]))
    torch.set_num_threads(8)
    res = T.render_camera_edges_train(sc, cam, t(g["depth_edge_mask_input"]))
    assert np.array_equal(res["edge_mask"].numpy(), g["edge_mask"]) and np.array_equal(res["convergent_mask"].numpy(), g["convergent_mask"])
    np.testing.assert_allclose(res["color"].detach().numpy(), g["color"], rtol=2e-5, atol=2e-6)
    wt = t(g["loss_weights"])
    loss = (res["color"] * wt).sum() + 0.1 * (res["normal"] * wt).sum()
    assert abs(loss.item() - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    loss.backward()
    assert _check_grads(sd, g) == golden_meta()["n_param_tensors_train_golden"]


def test_g14_training_forward_and_parameter_gradients():
    g = golden("g14_train_S1_c32.npz")
    nets = scenes.build_networks("S1")
    mt, md = tables()
    sd = {k: T.leaf_state(cpu_sd(nets[k])) for k in NETS}
    sc = R.Scene(sd["sdf_network"], R.SDFSpec(), {k: (sd[k], R.GGX_SPECS[k]) for k in R.GGX_SPECS}, golden_meta()["light"], mt, md)
    cam = R.CameraSpec(int(g["W"]), int(g["H"]), t(g["K"]), t(g["W2C"]))
    torch.set_num_threads(8)
    res = T.render_camera_train(sc, cam)
    assert np.array_equal(res["convergent_mask"].numpy(), g["convergent_mask"])
    np.testing.assert_allclose(res["color"].detach().numpy(), g["color"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(res["normal"].detach().numpy(), g["normal"], rtol=2e-5, atol=2e-6)
    wt = t(g["loss_weights"])
    loss = (res["color"] * wt).sum() + 0.1 * (res["normal"] * wt).sum()
    assert abs(loss.item() - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    loss.backward()
    n = 0
    for name in NETS:
        for pname, p in sd[name].items():
            if not p.is_floating_point():
                continue
            key = "%s/%s" % (name, pname)
            assert p.grad is not None, key
            gr = p.grad.reshape(-1).double().numpy()
            ref_n = float(g["gnorm:" + key])
            assert abs(np.linalg.norm(gr) - ref_n) <= 2e-4 * max(ref_n, 1e-12), (key, np.linalg.norm(gr), ref_n)
            ref_s = g["gsample:" + key]
            np.testing.assert_allclose(gr[_sample_idx(gr.size)], ref_s, rtol=2e-3, atol=2e-5 * max(np.abs(ref_s).max(), 1e-12), err_msg=key)
            n += 1
    assert n == golden_meta()["n_param_tensors_train_golden"]
