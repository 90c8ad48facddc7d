"""GPU: BASELINE.json's full size (800x800, config C1) through size-independent properties, plus the tracer's
edge cases.  The oracle needs minutes at this size, so the checks are invariants of the algorithm and the
oracle's committed work counts (tests/golden/work_counts.json, made by tools/count_work.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import iron_ref as R
from iron_amd import scenes
from iron_amd.raytracer import Camera, RayTracer, SDFHandle, intersect_sphere, raytrace_camera, render_camera
from iron_amd.renderer_ggx import GGXColocatedRenderer
from iron_amd.rendering_func import make_render_fn

from _util import GOLDEN, oracle_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def frame800():
    nets = {k: v.cuda() for k, v in scenes.build_networks("S0").items()}
    K, W2C = scenes.fixture_camera_matrices(800, 800)
    cam = Camera(800, 800, K.cuda(), W2C.cuda())
    fn = make_render_fn(GGXColocatedRenderer(use_cuda=True))
    import iron_amd.raytracer as rt
    tracer = RayTracer()
    rt.VERBOSE_MODE = True
    try:
        res = render_camera(cam, nets["sdf_network"], tracer, nets, fn, fill_holes=False, handle_edges=False)
    finally:
        rt.VERBOSE_MODE = False
    torch.cuda.synchronize()
    return nets, cam, fn, res, tracer.last_stats


def test_fullsize_work_counts_match_oracle(frame800):
    """H (hits) and the reference-equivalent evaluation count E of the 800x800 frame vs the CPU oracle."""
    _, _, _, res, st = frame800
    wc = json.load(open(os.path.join(GOLDEN, "work_counts.json")))["S0_800"]
    H = int(res["convergent_mask"].sum())
    assert abs(H - wc["H"]) <= 8, (H, wc["H"])                       # a handful of silhouette flips at most
    assert abs(st["n_evals_ref"] - wc["E"]) <= wc["E"] * 1e-4         # same algorithm, same work
    assert st["n_evals"] <= st["n_evals_ref"]                          # early exit only removes work
    assert st["reserved"] == 0                                       # no k_sampler workgroup gave up polling its work queue
    assert abs(st["n_sampler"] - wc["n_sampler"]) <= 16


def test_fullsize_invariants(frame800):
    nets, cam, _, res, st = frame800
    conv = res["convergent_mask"]
    assert conv.shape == (800, 800) and conv.dtype == torch.bool
    # every convergent ray sits on the zero level set to the tracer's tolerance: sphere-traced rays meet
    # |sdf| <= 5e-5 (raytracer.py:128-133); bisected rays are inside a 1e-4 bracket (|sdf| <~ 1e-4 * |grad|)
    s = res["sdf"][conv].abs()
    assert float(s.max()) <= 2e-4
    assert float((s <= 5e-5).float().mean()) >= 0.6
    # re-evaluating the network at the returned points reproduces the returned sdf (same kernel: bitwise)
    re = nets["sdf_network"].sdf(res["points"][conv])[:, 0]
    assert torch.equal(re, res["sdf"][conv])
    # points lie on their rays at the returned distance, inside the unit sphere
    p = res["ray_o"] + res["ray_d"] * res["distance"].unsqueeze(-1)
    assert float((p - res["points"])[conv].abs().max()) <= 2e-6
    assert float(res["points"][conv].norm(dim=-1).max()) <= 1.0 + 1e-5
    # depth = distance / |d| on hits, 0 elsewhere (raytracer.py:393,552)
    assert torch.all(res["depth"][~conv] == 0)
    np.testing.assert_allclose(res["depth"][conv].cpu().numpy(), (res["distance"] / res["ray_d_norm"])[conv].cpu().numpy(), rtol=1e-6)
    # shading outputs: zero off the mask, finite and in range on it
    for k in ("color", "diffuse_color", "specular_color", "diffuse_albedo", "specular_albedo", "normal"):
        assert torch.all(res[k][~conv] == 0), k
        assert torch.isfinite(res[k]).all(), k
    assert torch.all(res["specular_roughness"][~conv] == 0)
    n = res["normal"][conv]
    assert float((n.norm(dim=-1) - 1).abs().max()) <= 1e-5                       # render_surface.py:127
    assert float(res["diffuse_albedo"][conv].min()) >= 0 and float(res["diffuse_albedo"][conv].max()) <= 1.0   # sigmoid
    assert float(res["specular_roughness"][conv].min()) >= 0.01                  # rendering_func.py:11
    sa = res["specular_albedo"][conv]
    assert torch.equal(sa[:, 0], sa[:, 1]) and torch.equal(sa[:, 1], sa[:, 2])   # channel mean (is_metal=False)
    col = res["color"][conv]
    np.testing.assert_allclose(col.cpu().numpy(), (res["diffuse_color"] + res["specular_color"])[conv].cpu().numpy(), rtol=1e-6, atol=1e-8)
    assert float(col.min()) >= 0


def test_fullsize_deterministic_and_linear_in_light(frame800):
    nets, cam, fn, res, _ = frame800
    res2 = render_camera(cam, nets["sdf_network"], RayTracer(), nets, fn, fill_holes=False, handle_edges=False)
    for k in ("convergent_mask", "distance", "points", "color", "normal"):
        assert torch.equal(res[k], res2[k]), k                                    # idempotent, bitwise
    # GGX is linear in the light intensity (renderer_ggx.py:94): doubling the light doubles the colour exactly
    old = float(nets["point_light_network"].get_light())
    try:
        nets["point_light_network"].set_light(2.0 * old)
        res3 = render_camera(cam, nets["sdf_network"], RayTracer(), nets, fn, fill_holes=False, handle_edges=False)
    finally:
        nets["point_light_network"].set_light(old)
    assert torch.equal(res3["convergent_mask"], res["convergent_mask"])
    np.testing.assert_allclose(res3["color"].cpu().numpy(), 2.0 * res["color"].cpu().numpy(), rtol=2e-6, atol=0)


def test_chunk_global_bisection_semantics():
    """Several reference chunks in one launch: rays of a chunk share the bisection count (raytracer.py:204-217).
    Compared ray by ray with the oracle run chunk by chunk."""
    nets = scenes.build_networks("S1")
    sc = oracle_scene(nets)
    sdf = nets["sdf_network"].cuda()
    K, W2C = scenes.fixture_camera_matrices(56, 56)
    cam = Camera(56, 56, K.cuda(), W2C.cuda())
    for chunk in (500, 1000, 3136):
        res = raytrace_camera(cam, sdf, RayTracer(), max_num_rays=chunk)
        ref = R.raytrace_camera(sc, R.CameraSpec(56, 56, K, W2C), max_num_rays=chunk)
        conv = res["convergent_mask"].cpu().numpy()
        assert int((conv != ref["convergent_mask"].numpy()).sum()) <= 2
        both = conv & ref["convergent_mask"].numpy()
        d = np.abs(res["distance"].cpu().numpy() - ref["distance"].numpy())[both]
        assert d.max() <= 2e-4
        # bisected rays end inside the reference's final bracket width: the extra chunk-wide iterations are applied
        assert np.percentile(d, 99) <= 5e-5


def test_tracer_edge_cases():
    nets = scenes.build_networks("S0")
    sc = oracle_scene(nets)
    sdf = nets["sdf_network"].cuda()
    h = SDFHandle(sdf)
    dev = "cuda"
    # all rays miss the unit sphere: nothing converges, state = the single initial evaluation
    o = torch.tensor([[0.0, 0.0, 3.0]], device=dev).repeat(70, 1)
    d = torch.nn.functional.normalize(torch.tensor([[1.0, 0.2, 0.0]], device=dev).repeat(70, 1), dim=-1)
    m, near, far = intersect_sphere(o, d, 1.0)
    assert not bool(m.any())
    out = RayTracer()(h, o, d, near, far, m)
    assert not bool(out["convergent_mask"].any())
    ref = R.raytracer_forward(sc.sdf_fn, o.cpu(), d.cpu(), near.cpu(), far.cpu(), m.cpu())
    np.testing.assert_allclose(out["sdf"].cpu().numpy(), ref["sdf"].numpy(), atol=5e-6)
    np.testing.assert_allclose(out["distance"].cpu().numpy(), ref["distance"].numpy(), atol=1e-6)
    # work mask all False although rays cross the object
    o2 = torch.tensor([[0.0, 0.0, 2.0]], device=dev).repeat(33, 1)
    d2 = torch.tensor([[0.0, 0.0, -1.0]], device=dev).repeat(33, 1)
    m2, n2, f2 = intersect_sphere(o2, d2, 1.0)
    out2 = RayTracer()(h, o2, d2, n2, f2, torch.zeros_like(m2))
    assert not bool(out2["convergent_mask"].any())
    # non-default tracer parameters (n_steps not a multiple of 32, fewer sphere-tracing iterations)
    K, W2C = scenes.fixture_camera_matrices(40, 40)
    cam = Camera(40, 40, K.cuda(), W2C.cuda())
    for kw in (dict(n_steps=100, sphere_tracing_iters=4), dict(n_steps=33, sphere_tracing_iters=0),
               dict(sdf_threshold=1e-3, n_steps=64)):
        tr = RayTracer(**kw)
        res = raytrace_camera(cam, sdf, tr, max_num_rays=50000)
        prm = R.TracerParams(sdf_threshold=kw.get("sdf_threshold", 5e-5), sphere_tracing_iters=kw.get("sphere_tracing_iters", 16),
                             n_steps=kw.get("n_steps", 128))
        ref = R.raytrace_camera(sc, R.CameraSpec(40, 40, K, W2C), max_num_rays=50000, prm=prm)
        conv = res["convergent_mask"].cpu().numpy()
        assert int((conv != ref["convergent_mask"].numpy()).sum()) <= 2, kw
        both = conv & ref["convergent_mask"].numpy()
        if both.any():
            tol = 2.5 * kw.get("sdf_threshold", 5e-5) + 1e-4
            assert np.abs(res["distance"].cpu().numpy() - ref["distance"].numpy())[both].max() <= tol, kw


def test_fullsize_silhouette_handling_invariants(frame800):
    """Row f-1 at 800x800 (the oracle needs minutes here): invariants of fill_holes / handle_edges."""
    nets, cam, fn, plain, _ = frame800
    from iron_amd.raytracer import morph_closing3x3, sobel_magnitude
    res = render_camera(cam, nets["sdf_network"], RayTracer(), nets, fn, fill_holes=True, handle_edges=True)
    torch.cuda.synchronize()
    H, W = cam.H, cam.W
    edge, conv = res["edge_mask"], res["convergent_mask"]
    idx = res["edge_pixel_idx"]
    n_edge = int(idx.numel())
    assert n_edge > 500  # a sphere of radius ~0.5 seen at 800^2 has a ~2 000 px silhouette
    assert not bool((edge & conv).any())                      # raytracer.py:583: convergent_mask &= ~edge_mask
    assert torch.equal(torch.sort(idx)[0], torch.nonzero(edge.reshape(-1)).reshape(-1))  # one entry per edge pixel
    assert res["edge_points"].shape == (n_edge, 3) and res["edge_uv"].shape == (n_edge, 2)
    # every edge pixel holds the projection of its edge point (raytracer.py:481-500)
    pix = torch.floor(res["edge_uv"]).long()
    assert torch.equal(pix[:, 1] * W + pix[:, 0], idx)
    # edge points lie on the surface and at the silhouette: |sdf| small, |n.v| <= 0.05 (+ rounding)
    sdf, grad = nets["sdf_network"].get_sdf_and_gradient(res["edge_points"])
    assert float(sdf.abs().max()) <= 5e-3
    view = cam.get_camera_origin().reshape(1, 3) - res["edge_points"]
    view = view / view.norm(dim=-1, keepdim=True)
    dot = ((grad / grad.norm(dim=-1, keepdim=True)) * view).sum(-1)
    assert float(dot.abs().max()) <= 5e-2 + 1e-4
    # they sit on the silhouette of the plain render: within 2 px of a depth discontinuity of the hole-filled depth
    depth = morph_closing3x3(plain["depth"])
    near_edge = torch.nn.functional.max_pool2d((sobel_magnitude(depth) > 1e-2).float()[None, None], 5, 1, 2)[0, 0] > 0
    assert float(near_edge.reshape(-1)[idx].float().mean()) >= 0.99
    # away from the silhouette nothing changed; the edge pixels were re-coloured (blend of an object and a background ray)
    untouched = ~near_edge
    assert torch.equal(res["color"][untouched], plain["color"][untouched])
    assert float((res["color"].reshape(-1, 3)[idx] - plain["color"].reshape(-1, 3)[idx]).abs().max()) > 0
    # hole filling is idempotent on its own output
    d2 = morph_closing3x3(depth)
    assert torch.equal(d2, depth)


def _tile_colour_check(res, g, label):
    """Every pixel, not only the sub-lattice: colour summed over 8x8 tiles (pixels that hit in the reference's fp32 AND fp64
    runs) against the reference's fp64 tile sums; the reference's own fp32 tile sums give the floor."""
    n, T = int(g["res"]), int(g["tile"])
    m = (np.unpackbits(g["mask_bits"])[: n * n].astype(bool) & np.unpackbits(g["mask64_bits"])[: n * n].astype(bool)).reshape(n, n)
    col = res["color"].cpu().numpy().astype(np.float64) * m[..., None]
    mine = col.reshape(n // T, T, n // T, T, 3).sum(axis=(1, 3))
    ref64, ref32 = g["color_tile_sum_fp64"], g["color_tile_sum"]
    r64 = float(np.linalg.norm(mine - ref64) / np.linalg.norm(ref64))
    floor = float(np.linalg.norm(ref32 - ref64) / np.linalg.norm(ref64))
    worst = np.abs(mine - ref64).max(axis=-1) / np.maximum(g["tile_hits"], 1)
    print("%s: all %d pixels in 8x8 tiles: colour tile-sum rel-L2 hip~ref64 %.3e (ref32~ref64 %.3e); worst tile mean |d| per hit %.2e" % (
        label, n * n, r64, floor, worst.max()))
    return r64, floor, float(worst.max())


def test_fullsize_vs_reference_golden(frame800):
    """BASELINE config C1 itself against the REAL reference (tests/golden/make_golden_800.py: fp32 and fp64 runs of the
    reference's render_camera at 800x800): the complete hit mask, and colour / normal / distance on the [::4, ::4]
    sub-lattice.  Target (BASELINE.json north_star): colour rel-L2 <= 1e-4 vs the reference; like at 128^2 the bound is read
    against the reference's own fp32-vs-fp64 figure on the same pixels."""
    from _util import golden, golden_meta, rel_l2
    _, _, _, res, _ = frame800
    g = golden("g12_S0_800.npz")
    n = int(g["res"]); st = int(g["stride"])
    mask_ref = np.unpackbits(g["mask_bits"])[: n * n].astype(bool).reshape(n, n)
    conv = res["convergent_mask"].cpu().numpy()
    flips = int((conv != mask_ref).sum())
    flips_ref = golden_meta()["mask_flips_ref32_ref64_S0_800"]
    sub = (slice(None, None, st), slice(None, None, st))
    both = conv[sub] & mask_ref[sub] & np.unpackbits(g["mask64_bits"])[: n * n].astype(bool).reshape(n, n)[sub]
    col = res["color"].cpu().numpy()[sub]
    r32 = rel_l2(col[both], g["color"][both])
    r64 = rel_l2(col[both], g["color_fp64"][both])
    floor = rel_l2(g["color"][both], g["color_fp64"][both])
    dist = np.abs(res["distance"].cpu().numpy()[sub][both] - g["distance"][both])
    print("800x800 S0: mask flips %d (reference fp32 vs fp64: %d)  colour rel-L2 hip~ref32 %.3e  hip~ref64 %.3e  ref32~ref64 %.3e"
          "  |d distance| p99 %.2e max %.2e" % (flips, flips_ref, r32, r64, floor, np.percentile(dist, 99), dist.max()))
    assert int(conv.sum()) == golden_meta()["n_conv_S0_800"]
    assert flips <= max(2, 2 * flips_ref)
    assert r32 <= 1e-4 and r64 <= 1e-4, (r32, r64)          # the north_star bound itself, against both runs (measured 1e-5)
    assert np.percentile(dist, 99) <= 2e-4
    t64, tfloor, tworst = _tile_colour_check(res, g, "800x800 S0")
    assert t64 <= 1e-4 and tworst <= 2e-3, (t64, tworst)


def test_fullsize_bumpy_scene_vs_reference_golden():
    """The same at 800x800 on scene S1 (the perturbed, bumpy SDF whose grazing pixels are chaotic) against the REAL reference
    (tests/golden/make_golden_800.py S1): complete hit mask, colour / distance on the [::4, ::4] sub-lattice, read against the
    reference's own fp32-vs-fp64 figures."""
    from _util import golden, golden_meta, rel_l2
    nets = {k: v.cuda() for k, v in scenes.build_networks("S1").items()}
    K, W2C = scenes.fixture_camera_matrices(800, 800)
    res = render_camera(Camera(800, 800, K.cuda(), W2C.cuda()), nets["sdf_network"], RayTracer(), nets, make_render_fn(GGXColocatedRenderer(use_cuda=True)),
                        fill_holes=False, handle_edges=False)
    g = golden("g12_S1_800.npz")
    n, st = int(g["res"]), int(g["stride"])
    mask_ref = np.unpackbits(g["mask_bits"])[: n * n].astype(bool).reshape(n, n)
    mask64 = np.unpackbits(g["mask64_bits"])[: n * n].astype(bool).reshape(n, n)
    conv = res["convergent_mask"].cpu().numpy()
    flips = int((conv != mask_ref).sum())
    flips_ref = golden_meta()["mask_flips_ref32_ref64_S1_800"]
    sub = (slice(None, None, st), slice(None, None, st))
    both = conv[sub] & mask_ref[sub] & mask64[sub]
    col = res["color"].cpu().numpy()[sub]
    r32, r64 = rel_l2(col[both], g["color"][both]), rel_l2(col[both], g["color_fp64"][both])
    floor = rel_l2(g["color"][both], g["color_fp64"][both])
    dist = np.abs(res["distance"].cpu().numpy()[sub][both] - g["distance"][both])
    print("800x800 S1: mask flips %d (reference fp32 vs fp64: %d)  colour rel-L2 hip~ref32 %.3e  hip~ref64 %.3e  ref32~ref64 %.3e"
          "  |d distance| p99 %.2e max %.2e" % (flips, flips_ref, r32, r64, floor, np.percentile(dist, 99), dist.max()))
    assert flips <= max(4, 2 * flips_ref)
    # the fixed north_star bound against the reference's fp64 run (measured 6e-5).  Against its fp32 run the figure (1.9e-4) is
    # information only: on this chaotic scene the reference's own fp32 and fp64 renders are 2.0e-4 apart on these pixels.
    assert r64 <= 1e-4, (r64, floor)
    assert r32 <= max(1e-4, 1.5 * floor), (r32, floor)
    assert np.percentile(dist, 99) <= 2e-4
    t64, tfloor, tworst = _tile_colour_check(res, g, "800x800 S1")
    assert t64 <= max(1e-4, 1.5 * tfloor), (t64, tfloor)
