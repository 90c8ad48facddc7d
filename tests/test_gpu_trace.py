"""GPU parity: the HIP sphere tracer (RayTracer.forward chain through the C ABI) vs the reference goldens
and the oracle.

Tolerances (protocol ii): the tracer stops at |sdf| <= 5e-5 and bisects to 1e-4-wide
intervals, so a hit position is only defined to ~1e-4 along the ray; MFMA summation order differs
from MKL's, so a ray may converge one step earlier or later.  We therefore require
  * convergent-mask flips: <= 0.1 % of rays (reported),
  * rays convergent in both: |d distance| <= 2e-4, |d point| <= 2e-4 and |sdf| <= 1e-4,
  * eval count of the HIP tracer <= the reference's E (early exit in the dense sampler only removes work).
"""
import numpy as np
import pytest
import torch

from oracle import iron_ref as R
from iron_amd import scenes
from iron_amd.raytracer import Camera, RayTracer, raytrace_camera

from _util import golden, oracle_scene, t

pytestmark = pytest.mark.gpu


def _trace(scene, tag):
    g = golden("g67_%s_%s.npz" % (scene, tag))
    nets = scenes.build_networks(scene)
    sdf = nets["sdf_network"].cuda()
    cam = Camera(int(g["W"]), int(g["H"]), t(g["K"]).cuda(), t(g["W2C"]).cuda())
    tracer = RayTracer()
    import iron_amd.raytracer as rt
    rt.VERBOSE_MODE = True  # collect stats
    try:
        res = raytrace_camera(cam, sdf, tracer, max_num_rays=50000)
    finally:
        rt.VERBOSE_MODE = False
    torch.cuda.synchronize()
    return g, res, tracer.last_stats


@pytest.mark.parametrize("scene,tag", [("S0", "c0"), ("S1", "c0"), ("S0", "v128"), ("S1", "v128")])
def test_trace_matches_reference(scene, tag):
    g, res, stats = _trace(scene, tag)
    conv = res["convergent_mask"].cpu().numpy()
    assert conv.dtype == np.bool_ and conv.shape == g["convergent_mask"].shape
    flips = int((conv != g["convergent_mask"]).sum())
    n = conv.size
    assert flips <= max(1, n // 1000), "mask flips %d / %d" % (flips, n)
    both = conv & g["convergent_mask"]
    dd = np.abs(res["distance"].cpu().numpy() - g["distance"])[both]
    assert dd.max() <= 2e-4, dd.max()
    assert np.abs(res["sdf"].cpu().numpy()[both]).max() <= 1e-4
    dp = np.abs(res["points"].cpu().numpy() - g["points"])[both]
    assert dp.max() <= 2e-4
    depth = res["depth"].cpu().numpy()
    assert np.all(depth[~conv] == 0.0)
    np.testing.assert_allclose(res["ray_d"].cpu().numpy(), g["ray_d"], rtol=2e-6, atol=2e-7)
    # the result dict keeps the reference's key set
    assert set(res.keys()) == {"convergent_mask", "points", "sdf", "distance", "depth", "uv", "ray_o", "ray_d", "ray_d_norm"}
    assert stats["n_evals"] <= int(g["trace_evals"])
    assert stats["n_conv"] == int(conv.sum())
    print(scene, tag, "flips", flips, "max|d dist|", dd.max(), "evals hip/ref", stats["n_evals"], int(g["trace_evals"]), stats)


def test_non_convergent_state_like_reference():
    """Rays that never touched the unit sphere keep their single initial evaluation (raytracer.py:110-111);
    sampled rays without a root are zeroed (raytracer.py:158-160)."""
    g, res, _ = _trace("S1", "v128")
    conv = res["convergent_mask"].cpu().numpy()
    same_nc = (~conv) & (~g["convergent_mask"])
    d = res["distance"].cpu().numpy()
    zero_ref = same_nc & (g["distance"] == 0.0) & (np.abs(g["points"]).sum(-1) == 0.0)
    zero_hip = same_nc & (d == 0.0) & (np.abs(res["points"].cpu().numpy()).sum(-1) == 0.0)
    # the zeroed (sampled, rootless) sets agree up to the few rays whose sphere-trace exit differs
    assert int((zero_ref != zero_hip).sum()) <= max(2, conv.size // 500)
    # rays that left the unit sphere without converging keep their last sphere-tracing state (not zeros): the
    # state agrees with the reference to the accumulated step error of <= 16 evaluations
    agree = same_nc & ~zero_ref & ~zero_hip
    assert agree.sum() > 100
    assert np.abs(res["distance"].cpu().numpy() - g["distance"])[agree].max() <= 5e-4
    # (rays that miss the unit sphere altogether: tests/test_gpu_fullsize.py::test_tracer_edge_cases)


def test_chunk_semantics_and_direct_forward():
    """RayTracer.forward called directly on a ray batch (one reference call = one chunk)."""
    from iron_amd.raytracer import SDFHandle, intersect_sphere
    nets = scenes.build_networks("S1")
    sc = oracle_scene(nets)
    sdf = nets["sdf_network"].cuda()
    K, W2C = scenes.fixture_camera_matrices(48, 48)
    cam = Camera(48, 48, K.cuda(), W2C.cuda())
    o, d, _ = cam.get_rays(cam.get_uv())
    o, d = o.reshape(-1, 3), d.reshape(-1, 3)
    m, near, far = intersect_sphere(o, d, 1.0)
    out = RayTracer()(SDFHandle(sdf), o, d, near, far, m)
    ref = R.raytracer_forward(sc.sdf_fn, o.cpu(), d.cpu(), near.cpu(), far.cpu(), m.cpu())
    conv = out["convergent_mask"].cpu().numpy()
    flips = int((conv != ref["convergent_mask"].numpy()).sum())
    assert flips <= 2
    both = conv & ref["convergent_mask"].numpy()
    assert np.abs(out["distance"].cpu().numpy() - ref["distance"].numpy())[both].max() <= 2e-4
    # the reference's own calling convention (raytracer.py:375: a lambda around the network) gives the same result;
    # a lambda that alters the distance, or one that wraps no network, is refused instead of being traced wrongly
    from iron_amd import _lib
    out_l = RayTracer()(lambda x: sdf(x)[..., 0], o, d, near, far, m)
    for k in ("convergent_mask", "distance", "points", "sdf"):
        assert torch.equal(out_l[k], out[k]), k
    out_b = RayTracer()(lambda x: sdf.sdf(x)[..., 0], o, d, near, far, m)
    assert torch.equal(out_b["distance"], out["distance"])
    with pytest.raises(_lib.IronError):
        RayTracer()(lambda x: sdf(x)[..., 0] - 0.01, o, d, near, far, m)
    with pytest.raises(_lib.IronError):
        RayTracer()(lambda x: x.norm(dim=-1) - 0.5, o, d, near, far, m)
    # empty batch
    e = torch.zeros(0, 3, device="cuda")
    out0 = RayTracer()(SDFHandle(sdf), e, e, e[:, 0], e[:, 0], torch.zeros(0, dtype=torch.bool, device="cuda"))
    assert out0["points"].shape == (0, 3)


@torch.no_grad()
def test_split_form_is_bit_equal():
    """iron_set_trace_split: the rays of one call run as 2 / 3 / 4 independent parts on separate streams (own queues and lists, the
    chunk-count table shared through atomicMax).  Rays are independent, so every output and every work counter must equal the
    one-part run bit for bit -- with several bisection chunks straddling the part boundaries."""
    from iron_amd import _lib
    from iron_amd.raytracer import Camera, RayTracer, raytrace_pixels
    import iron_amd.raytracer as RT
    dev = torch.device("cuda", 0)
    nets = {k: v.to(dev) for k, v in scenes.build_networks("S1").items()}
    K, W2C = scenes.fixture_camera_matrices(160, 160)
    cam = Camera(160, 160, K.to(dev), W2C.to(dev))
    lib = _lib.load()
    old = RT.VERBOSE_MODE
    prev = lib.iron_set_trace_split(1)
    try:
        RT.VERBOSE_MODE = True
        tr = RayTracer()
        want = raytrace_pixels(nets["sdf_network"], tr, cam.get_uv(), cam, max_num_rays=3000)
        want_stats = dict(tr.last_stats)
        assert int(want["convergent_mask"].sum()) > 1000 and want_stats["n_bisect"] > 50
        for parts in (2, 3, 4):
            lib.iron_set_trace_split(parts)
            tr2 = RayTracer()
            got = raytrace_pixels(nets["sdf_network"], tr2, cam.get_uv(), cam, max_num_rays=3000)
            torch.cuda.synchronize()
            for k in ("convergent_mask", "points", "sdf", "distance", "depth"):
                assert torch.equal(got[k], want[k]), (parts, k)
            assert tr2.last_stats == want_stats, (parts, tr2.last_stats, want_stats)
    finally:
        RT.VERBOSE_MODE = old
        lib.iron_set_trace_split(prev)


@torch.no_grad()
def test_stage_methods_match_the_oracle():
    """RayTracer.sphere_tracing / ray_sampler / rootfind as callable methods (raytracer.py:105-220; VERDICT r2 'missing' 3): each
    against the oracle's restatement of the same lines on the rays of a 96x96 view of S1 -- masks equal, values within the
    tracer's tolerance (abs(d distance) <= 2e-4, as test_trace_matches_reference), and chained by hand they reproduce forward()."""
    from iron_amd.raytracer import Camera, RayTracer, SDFHandle, intersect_sphere
    dev = torch.device("cuda", 0)
    nets_cpu = scenes.build_networks("S1")
    sc = oracle_scene(nets_cpu)
    net = nets_cpu["sdf_network"].to(dev)
    K, W2C = scenes.fixture_camera_matrices(96, 96)
    cam = Camera(96, 96, K.to(dev), W2C.to(dev))
    ro, rd, _ = cam.get_rays(cam.get_uv())
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    hit, near, far = intersect_sphere(ro, rd, 1.0)
    tr = RayTracer()
    h = SDFHandle(net)
    prm = R.TracerParams()
    sdf_fn = lambda x: R.sdf_forward(sc.sdf_sd, sc.sdf_spec, x)[:, 0]
    c_ro, c_rd, c_near, c_far, c_hit = ro.cpu(), rd.cpu(), near.cpu(), far.cpu(), hit.cpu()

    # sphere_tracing
    conv, unf, p, s, t = tr.sphere_tracing(h, ro, rd, near, far, hit)
    rconv, runf, rp, rs, rt = R.sphere_tracing(sdf_fn, c_ro, c_rd, c_near, c_far, c_hit, prm)
    assert int((conv.cpu() != rconv).sum()) <= 2 and int((unf.cpu() != runf).sum()) <= 2
    both = (conv.cpu() & rconv)
    assert float((t.cpu() - rt)[both].abs().max()) <= 2e-4
    same = ~(conv.cpu() ^ rconv) & ~(unf.cpu() ^ runf)
    assert float((s.cpu() - rs)[same & ~runf].abs().max()) <= 2e-4
    assert int(unf.sum()) > 500 and int(conv.sum()) > 500

    # ray_sampler on the oracle's unfinished rays (same inputs on both sides)
    m = runf
    pos = (rs[m] > 0.0).float()
    s_min = pos * rt[m] + (1.0 - pos) * c_near[m]
    s_max = pos * c_far[m] + (1.0 - pos) * rt[m]
    rroot, rsp, rss, rst, _ = R.ray_sampler(sdf_fn, c_ro[m], c_rd[m], s_min.clone(), s_max.clone(), prm)
    root, sp, ss, st = tr.ray_sampler(h, c_ro[m].to(dev), c_rd[m].to(dev), s_min.to(dev), s_max.to(dev))
    assert int((root.cpu() != rroot).sum()) <= 2
    bothr = root.cpu() & rroot
    assert int(bothr.sum()) > 50
    assert float((st.cpu() - rst)[bothr].abs().max()) <= 2e-4 and float((sp.cpu() - rsp)[bothr].abs().max()) <= 2e-4
    nor = ~root.cpu() & ~rroot
    assert float(st.cpu()[nor].abs().max()) == 0.0 and float(sp.cpu()[nor].abs().max()) == 0.0 and float(ss.cpu()[nor].abs().max()) == 0.0

    # rootfind on hand-made brackets around the oracle's roots (and two brackets that need no iteration)
    k = int(bothr.sum())
    d_lo = (rst[bothr] - 0.013).clone(); d_hi = (rst[bothr] + 0.011).clone()
    oo, dd = c_ro[m][bothr], c_rd[m][bothr]
    f_lo = sdf_fn(oo + dd * d_lo.unsqueeze(-1)); f_hi = sdf_fn(oo + dd * d_hi.unsqueeze(-1))
    f_lo[:2] = -1.0   # not a bracket: these two rays make no step of their own but are moved by the call's shared loop
    want_p, want_d, want_f, _ = R.rootfind(sdf_fn, f_lo.clone(), f_hi.clone(), d_lo.clone(), d_hi.clone(), oo, dd, prm)
    got_p, got_d, got_f = tr.rootfind(h, f_lo.to(dev), f_hi.to(dev), d_lo.to(dev), d_hi.to(dev), oo.to(dev), dd.to(dev))
    assert got_p.shape == (k, 3) and got_d.shape == (k,)
    assert float((got_d.cpu() - want_d).abs().max()) <= 2e-4 and float((got_f.cpu() - want_f).abs().max()) <= 2e-4

    # chained like raytracer.py:45-79 the stages give forward()'s result exactly (same kernels, same order)
    full = tr(h, ro, rd, near, far, hit)
    mm = unf
    posg = (s[mm] > 0.0).float()
    g_min = posg * t[mm] + (1.0 - posg) * near[mm]
    g_max = posg * far[mm] + (1.0 - posg) * t[mm]
    r2, p2, s2, t2 = tr.ray_sampler(h, ro[mm], rd[mm], g_min, g_max)
    conv[mm] = r2; p[mm] = p2; s[mm] = s2; t[mm] = t2
    assert torch.equal(conv, full["convergent_mask"]) and torch.equal(t, full["distance"]) and torch.equal(s, full["sdf"])
    assert torch.equal(p, full["points"])


@torch.no_grad()
@pytest.mark.parametrize("n_steps", [9, 31, 32, 33, 64, 100, 128, 200])
def test_sampler_work_items_at_any_step_count(n_steps):
    """k_sampler marches a ray as work items of 32 samples and hands the rest of an unfinished ray to the queue (trace.hip): step
    counts that end inside a block, inside an item, exactly on an item boundary and beyond 128 must all give the reference's root
    (first sign change of the n_steps samples, raytracer.py:142-197), and the whole trace with that n_steps the oracle's result."""
    from iron_amd.raytracer import Camera, RayTracer, SDFHandle, intersect_sphere
    dev = torch.device("cuda", 0)
    nets_cpu = scenes.build_networks("S1")
    sc = oracle_scene(nets_cpu)
    net = nets_cpu["sdf_network"].to(dev)
    K, W2C = scenes.fixture_camera_matrices(64, 64)
    cam = Camera(64, 64, K.to(dev), W2C.to(dev))
    ro, rd, _ = cam.get_rays(cam.get_uv())
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    hit, near, far = intersect_sphere(ro, rd, 1.0)
    m = hit.cpu()
    prm = R.TracerParams(n_steps=n_steps)
    sdf_fn = lambda x: R.sdf_forward(sc.sdf_sd, sc.sdf_spec, x)[:, 0]
    c_ro, c_rd, c_near, c_far = ro.cpu()[m], rd.cpu()[m], near.cpu()[m], far.cpu()[m]
    # every ray that meets the bounding sphere, sampled over [near, far]: roots early, late and never
    rroot, rsp, rss, rst, _ = R.ray_sampler(sdf_fn, c_ro, c_rd, c_near.clone(), c_far.clone(), prm)
    tr = RayTracer(n_steps=n_steps)
    root, sp, ss, st = tr.ray_sampler(SDFHandle(net), c_ro.to(dev), c_rd.to(dev), c_near.to(dev), c_far.to(dev))
    assert int((root.cpu() != rroot).sum()) <= 2, (n_steps, int((root.cpu() != rroot).sum()))
    both = root.cpu() & rroot
    assert int(both.sum()) > 100
    assert float((st.cpu() - rst)[both].abs().max()) <= 2e-4 and float((sp.cpu() - rsp)[both].abs().max()) <= 2e-4
    nor = ~root.cpu() & ~rroot
    assert float(st.cpu()[nor].abs().max()) == 0.0 and float(ss.cpu()[nor].abs().max()) == 0.0
    # an empty list and a one-ray list go through the same queue
    e = torch.empty(0, 3, device=dev)
    r0 = tr.ray_sampler(SDFHandle(net), e, e, torch.empty(0, device=dev), torch.empty(0, device=dev))
    assert r0[0].numel() == 0
    r1 = tr.ray_sampler(SDFHandle(net), c_ro[:1].to(dev), c_rd[:1].to(dev), c_near[:1].to(dev), c_far[:1].to(dev))
    assert bool(r1[0].cpu()[0]) == bool(rroot[0]) and abs(float(r1[3].cpu()[0]) - float(rst[0])) <= 2e-4


@torch.no_grad()
def test_h2_tracer_agrees_with_the_exact_core_on_a_fixed_ray_set():
    """Tripwire for the compiler option the default build depends on (-mllvm -amdgpu-mfma-vgpr-form=1 has miscompiled k_sampler
    once: 17 of 88 056 roots lost): the same rays through the h2 kernels and through the exact-fp32 kernels of the
    same library (iron_net_force_exact on a second handle) must list the same rays for dense sampling, bracket the same roots up
    to the handful that sit on the threshold, and agree on the traced distance."""
    from iron_amd.raytracer import Camera, RayTracer, raytrace_pixels
    import iron_amd.raytracer as RT
    dev = torch.device("cuda", 0)
    K, W2C = scenes.fixture_camera_matrices(320, 320)
    cam = Camera(320, 320, K.to(dev), W2C.to(dev))
    out = {}
    old = RT.VERBOSE_MODE
    try:
        RT.VERBOSE_MODE = True
        for core in ("h2", "f32"):
            net = scenes.build_networks("S1")["sdf_network"].to(dev)
            if core == "f32":
                net.force_exact(True)
            tr = RayTracer()
            res = raytrace_pixels(net, tr, cam.get_uv(), cam, max_num_rays=50000)
            assert net.numeric_status()["exact_core"] == (core == "f32")
            out[core] = (res, dict(tr.last_stats))
    finally:
        RT.VERBOSE_MODE = old
    (a, sa), (b, sb) = out["h2"], out["f32"]
    n = 320 * 320
    print("   h2 vs exact core: sampler rays %d / %d, roots %d / %d, hits %d / %d" % (sa["n_sampler"], sb["n_sampler"], sa["n_bisect"],
                                                                                     sb["n_bisect"], sa["n_conv"], sb["n_conv"]))
    assert sb["n_bisect"] > 5000
    assert abs(sa["n_sampler"] - sb["n_sampler"]) <= 3e-4 * n          # rays within rounding of the 5e-5 threshold may swap lists
    assert abs(sa["n_bisect"] - sb["n_bisect"]) <= 3, (sa["n_bisect"], sb["n_bisect"])   # a lost root is what the miscompile looked like
    flips = int((a["convergent_mask"] != b["convergent_mask"]).sum())
    assert flips <= 3, flips
    both = a["convergent_mask"] & b["convergent_mask"]
    assert float((a["distance"] - b["distance"])[both].abs().max()) <= 2e-4
