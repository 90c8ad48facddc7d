"""Pins oracle/iron_ref.py (and the build's own constructors) to golden vectors that
tests/golden/make_golden.py recorded from the real reference.  CPU only.

The oracle repeats the reference's torch op sequence, so on the torch build the fixtures were made
with the results are bit-identical; the tolerances below only leave room for a different torch /
MKL build on the GPU box (they are ~1 ulp-level, far below any parity tolerance used elsewhere).
"""
import numpy as np
import pytest
import torch

from oracle import iron_ref as R
from iron_amd import scenes

from _util import golden, golden_meta, oracle_scene, state_hash, t, tables

TIGHT = dict(rtol=2e-6, atol=2e-7)


@pytest.fixture(scope="module")
def nets_s1():
    return scenes.build_networks("S1")


def test_constructors_reproduce_reference_state():
    """The build's SDFNetwork / RenderingNetwork constructors consume the RNG like the reference's
    (models/fields.py:47-76): seeded state dicts hash to the value recorded from the reference."""
    meta = golden_meta()
    for scene in ("S0", "S1", "S3"):
        assert state_hash(scenes.build_networks(scene)) == meta["state_sha256_" + scene]


def test_g1_positional_encoding():
    g = golden("g1_pe.npz")
    x = t(g["x"])
    for L in (4, 6, 10):
        out = R.positional_encoding(x, L).numpy()
        assert out.shape[1] == R.pe_width(L)
        np.testing.assert_allclose(out, g["pe%d" % L], **TIGHT)


def test_g2_sdf_forward_and_get_all(nets_s1):
    g = golden("g2_sdf.npz")
    sc = oracle_scene(nets_s1)
    x = t(g["x"])
    full = R.sdf_forward(sc.sdf_sd, sc.sdf_spec, x).numpy()
    np.testing.assert_allclose(full[:, 0], g["sdf"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(full[:256, 1:], g["feature256"], rtol=1e-5, atol=1e-6)
    y, feat, grad = R.sdf_get_all(sc.sdf_sd, sc.sdf_spec, x)
    np.testing.assert_allclose(y[:, 0].numpy(), g["getall_sdf"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(grad.numpy(), g["getall_grad"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(feat[:256].numpy(), g["getall_feature256"], rtol=1e-5, atol=1e-6)


def test_g3_material_networks(nets_s1):
    g = golden("g3_materials.npz")
    sc = oracle_scene(nets_s1)
    p, n, f = t(g["points"]), t(g["normals"]), t(g["features"])
    for key, name, view in (("raw_diffuse", "diffuse_albedo_network", -n), ("raw_specular", "specular_albedo_network", None),
                            ("raw_roughness", "specular_roughness_network", None)):
        sd, spec = sc.nets[name]
        out = R.rendering_forward(sd, spec, p, n, view, f).numpy()
        np.testing.assert_allclose(out, g[key], rtol=1e-5, atol=1e-6)
    m = R.get_materials(sc.nets, p, n, f)
    for k in ("diffuse_albedo", "specular_albedo", "specular_roughness"):
        np.testing.assert_allclose(m[k].numpy(), g[k], rtol=1e-5, atol=1e-6)


def test_g4_ggx():
    g = golden("g4_ggx.npz")
    mt, md = tables()
    prm = {"diffuse_albedo": t(g["diffuse_albedo"]), "specular_albedo": t(g["specular_albedo"]),
           "specular_roughness": t(g["specular_roughness"])}
    res = R.ggx_colocated(torch.tensor(float(g["light"])), t(g["distance"]), t(g["normal"]), t(g["viewdir"]), prm, mt, md)
    for k in ("diffuse_rgb", "specular_rgb", "rgb"):
        np.testing.assert_allclose(res[k].numpy(), g[k], rtol=1e-6, atol=1e-30)


def test_g5_rays_and_sphere():
    g = golden("g5_rays.npz")
    cam = R.CameraSpec(512, 512, t(g["K"]), t(g["W2C"]))
    o, d, dn = cam.get_rays(t(g["uv"]))
    np.testing.assert_allclose(d.numpy(), g["ray_d"], **TIGHT)
    np.testing.assert_allclose(dn.numpy(), g["ray_d_norm"], **TIGHT)
    np.testing.assert_allclose(o.numpy(), g["ray_o"], **TIGHT)
    m, near, far = R.intersect_sphere(o.reshape(-1, 3), d.reshape(-1, 3), 1.0)
    assert np.array_equal(m.numpy(), g["mask"])
    np.testing.assert_allclose(near.numpy(), g["near"], **TIGHT)
    np.testing.assert_allclose(far.numpy(), g["far"], **TIGHT)
    crop = cam.crop(64, 64, (224, 224))
    np.testing.assert_allclose(crop.K.numpy(), g["crop_K"], rtol=0, atol=0)
    np.testing.assert_allclose(crop.get_uv().numpy(), g["crop_uv"], rtol=0, atol=0)
    _, dc, dnc = crop.get_rays(crop.get_uv())
    np.testing.assert_allclose(dc.numpy(), g["crop_ray_d"], **TIGHT)
    np.testing.assert_allclose(dnc.numpy(), g["crop_ray_d_norm"], **TIGHT)


@pytest.mark.parametrize("scene,tag", [("S0", "c0"), ("S1", "c0"), ("S0", "v128"), ("S1", "v128")])
def test_g6_g7_trace_and_render(scene, tag):
    """RayTracer.forward chain + render_camera end to end, incl. the reference's eval count E."""
    g = golden("g67_%s_%s.npz" % (scene, tag))
    sc = oracle_scene(scenes.build_networks(scene), light=golden_meta()["light"])
    cam = R.CameraSpec(int(g["W"]), int(g["H"]), t(g["K"]), t(g["W2C"]))
    torch.set_num_threads(8)
    tr = R.raytrace_camera(sc, cam, max_num_rays=50000)
    assert sc.counter.evals == int(g["trace_evals"])
    flips = int((tr["convergent_mask"].numpy() != g["convergent_mask"]).sum())
    assert flips == 0
    for k in ("distance", "sdf", "depth", "ray_d_norm"):
        np.testing.assert_allclose(tr[k].numpy(), g[k], rtol=1e-5, atol=1e-6, err_msg=k)
    for k in ("points", "ray_d"):
        np.testing.assert_allclose(tr[k].numpy(), g[k], rtol=1e-5, atol=1e-6, err_msg=k)
    R.render_normal_and_color(sc, tr)
    for k in ("color", "diffuse_color", "specular_color", "diffuse_albedo", "specular_albedo", "specular_roughness",
              "normal"):
        assert tr[k].shape == g[k].shape, k
        np.testing.assert_allclose(tr[k].numpy(), g[k], rtol=2e-5, atol=2e-6, err_msg=k)
    assert int(tr["convergent_mask"].sum()) == int(g["n_conv"])


def test_result_dict_contract():
    """Key set / shapes / dtypes of render_camera."""
    sc = oracle_scene(scenes.build_networks("S0"))
    K, W2C = scenes.fixture_camera_matrices(16, 16)
    res = R.render_camera(sc, R.CameraSpec(16, 16, K, W2C))
    want = {"convergent_mask": (16, 16), "points": (16, 16, 3), "sdf": (16, 16), "distance": (16, 16), "depth": (16, 16),
            "uv": (16, 16, 2), "ray_o": (16, 16, 3), "ray_d": (16, 16, 3), "ray_d_norm": (16, 16), "color": (16, 16, 3),
            "diffuse_color": (16, 16, 3), "specular_color": (16, 16, 3), "diffuse_albedo": (16, 16, 3),
            "specular_albedo": (16, 16, 3), "specular_roughness": (16, 16), "normal": (16, 16, 3)}
    assert set(res.keys()) == set(want.keys())
    for k, sh in want.items():
        assert tuple(res[k].shape) == sh, k
    assert res["convergent_mask"].dtype == torch.bool


@pytest.mark.parametrize("scene", ["S0", "S1"])
def test_g8_edge_walk_and_edge_blend(scene):
    """Row f-1 behind the sobel mask: locate_edge_points + render_edge_pixels (raytracer.py:421-506, 665-729) vs the
    reference, fed with the same depth-edge mask.  (closing / sobel themselves are parity-unpinned: kornia is absent.)"""
    g = golden("g8_edges_%s.npz" % scene)
    sc = oracle_scene(scenes.build_networks(scene), light=golden_meta()["light"])
    cam = R.CameraSpec(int(g["W"]), int(g["H"]), t(g["K"]), t(g["W2C"]))
    res = R.render_camera_full(sc, cam, fill_holes=False, handle_edges=True, depth_edge_mask=t(g["depth_edge_mask_input"]))
    assert np.array_equal(res["edge_mask"].numpy(), g["edge_mask"])
    assert np.array_equal(res["edge_pixel_idx"].numpy(), g["edge_pixel_idx"])
    assert np.array_equal(res["convergent_mask"].numpy(), g["convergent_mask"])
    np.testing.assert_allclose(res["edge_points"].numpy(), g["edge_points"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(res["edge_uv"].numpy(), g["edge_uv"], rtol=1e-5, atol=1e-4)
    for k in ("color", "normal", "points", "uv"):
        np.testing.assert_allclose(res[k].numpy(), g[k], rtol=2e-5, atol=2e-6, err_msg=k)
    assert res["edge_pos_neg_normal"].shape == g["edge_pos_neg_normal"].shape
    # the edge pixels really were re-coloured
    idx = g["edge_pixel_idx"]
    assert np.abs(g["color"].reshape(-1, 3)[idx] - g["pre_edge_color"].reshape(-1, 3)[idx]).max() > 0


def test_closing_and_sobel_restatements_selfcheck():
    """Unpinned kornia restatements: internal consistency only (closing is extensive + idempotent, fills 1-px holes;
    sobel is zero on constants, |grad| on ramps, replicate border)."""
    torch.manual_seed(0)
    x = (torch.rand(20, 24) > 0.3).float() * (1.0 + torch.rand(20, 24))
    c = R.morph_closing3x3(x)
    assert torch.all(c >= x)
    assert torch.equal(R.morph_closing3x3(c), c)
    h = torch.ones(9, 9); h[4, 4] = 0
    assert torch.equal(R.morph_closing3x3(h), torch.ones(9, 9))
    assert float(R.sobel_magnitude(torch.full((8, 8), 3.0)).max()) == pytest.approx(1e-3, rel=1e-3)
    ramp = torch.arange(10.0).view(1, 10).expand(6, 10).contiguous() * 0.5
    s = R.sobel_magnitude(ramp)
    assert torch.allclose(s[:, 1:-1], torch.full((6, 8), (0.25 + 1e-6) ** 0.5))


# ---------------------------------------------------------------------------------------------------------------
# the fork's other co-located BRDF heads
# ---------------------------------------------------------------------------------------------------------------
def _g9_inputs():
    g = golden("g9_brdf_heads.npz")
    prm = {k: t(g[k]) for k in ("diffuse_albedo", "specular_albedo", "specular_roughness", "metallic", "dielectric",
                                "metallic_eta", "metallic_k", "dielectric_eta", "env_light")}
    return g, prm, torch.tensor(float(g["light"])), t(g["distance"]), t(g["normal"]), t(g["viewdir"])


def test_g9_composite_renderer():
    """CompositeRenderer.forward (renderer_ggx.py:781-858) incl. the env-light branch and the diffuse_rgb == rgb alias."""
    g, prm, light, dist, nrm, view = _g9_inputs()
    mt, md = tables()
    for tag, env in (("comp", False), ("compenv", True)):
        res = R.composite_forward(light, dist, nrm, view, prm, mt, md, use_env_light=env)
        keys = ["diffuse_rgb", "specular_rgb", "metallic_rgb", "dielectric_rgb", "rgb"] + (["env_light"] if env else [])
        assert set(res.keys()) == set(keys)
        for k in keys:
            np.testing.assert_allclose(res[k].numpy(), g["%s_%s" % (tag, k)], rtol=2e-6, atol=1e-30, err_msg=tag + k)
        assert np.array_equal(g[tag + "_diffuse_rgb"], g[tag + "_rgb"])  # the reference's in-place alias


def test_g9_simple_heads():
    g, prm, light, dist, nrm, view = _g9_inputs()
    kd, ks, a = prm["diffuse_albedo"], prm["specular_albedo"], prm["specular_roughness"]
    for tag, fn in (("smooth_dielectric", R.smooth_dielectric), ("thin_dielectric", R.thin_dielectric),
                    ("smooth_conductor", R.smooth_conductor), ("rough_conductor", R.rough_conductor)):
        res = fn(light, dist, nrm, view, kd, ks, a)
        for k in ("diffuse_rgb", "specular_rgb", "rgb"):
            np.testing.assert_allclose(res[k].numpy(), g["%s_%s" % (tag, k)], rtol=2e-6, atol=1e-30, err_msg=tag + k)
    assert golden_meta()["rough_plastic_reference_error"] == "TypeError"


def _comp_scene():
    nets = scenes.build_comp_networks()
    mt, md = tables()
    from _util import cpu_sd
    rn = {k: (cpu_sd(nets[k]), R.COMP_SPECS[k]) for k in R.COMP_SPECS}
    return nets, R.Scene(cpu_sd(nets["sdf_network"]), R.SDFSpec(), rn, golden_meta()["light"], mt, md, renderer="comp")


def test_g10_comp_materials_and_render():
    """comp2 constructors reproduce the reference's seeded state; get_materials_comp (rendering_func.py:19-49); and
    render_camera with the composite render_fn (render_surface.py:159-234) on the C0 crop."""
    nets, sc = _comp_scene()
    assert state_hash(nets) == golden_meta()["state_sha256_S2"]
    g = golden("g10_comp_materials.npz")
    m = R.get_materials_comp(sc.nets, t(g["points"]), t(g["normals"]), t(g["features"]))
    assert set(m.keys()) == {"diffuse_albedo", "specular_albedo", "metallic", "dielectric", "specular_roughness",
                             "metallic_eta", "metallic_k", "dielectric_eta"}
    for k, v in m.items():
        np.testing.assert_allclose(v.numpy(), g[k], rtol=1e-5, atol=1e-6, err_msg=k)
    g = golden("g10_comp_S2_c0.npz")
    cam = R.CameraSpec(int(g["W"]), int(g["H"]), t(g["K"]), t(g["W2C"]))
    torch.set_num_threads(8)
    res = R.render_camera(sc, cam)
    assert np.array_equal(res["convergent_mask"].numpy(), g["convergent_mask"])
    for k, w in R.COMP_RENDER_KEYS:
        assert tuple(res[k].shape) == g[k].shape, k
        np.testing.assert_allclose(res[k].numpy(), g[k], rtol=3e-5, atol=3e-6, err_msg=k)


def test_g11_points_only_head():
    """comp2's env_light_network: constructor parity (seed 7) and the points_only forward (fields.py:203-239)."""
    from iron_amd.network_conf import comp_env_light_network
    from _util import cpu_sd
    torch.manual_seed(7)
    env = comp_env_light_network()
    assert state_hash({"env_light_network": env}) == golden_meta()["state_sha256_env_light"]
    g = golden("g11_points_only.npz")
    spec = R.RenderSpec(d_in=3, d_out=1, n_layers=4, multires=6, multires_view=-1, mode="points_only", squeeze_out=False)
    out = R.rendering_forward(cpu_sd(env), spec, t(g["points"]), None, None, t(g["features"]))
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=1e-5, atol=1e-6)
