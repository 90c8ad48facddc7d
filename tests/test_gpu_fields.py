"""GPU parity: batched field queries and pointwise kernels through the C ABI vs the oracle."""
import numpy as np
import pytest
import torch

from oracle import iron_ref as R
from iron_amd import scenes

from _util import golden, oracle_scene, rel_l2, t

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def s1():
    nets = scenes.build_networks("S1")
    sc = oracle_scene(nets)
    gpu = {k: v.cuda() for k, v in nets.items()}
    return sc, gpu


@torch.no_grad()  # the inference kernels (under grad mode the same calls return attached tensors)
def test_sdf_forward_matches_oracle(s1):
    """SDFNetwork.sdf / .forward on random points in [-1,1]^3 (incl. a ragged tail) vs the oracle.
    Tolerance: rel-L2 <= 1e-5 on the sdf column (stage-wise MLP tolerance)."""
    sc, gpu = s1
    g = torch.Generator().manual_seed(5)
    for n in (1, 31, 32, 33, 1000, 4099):
        x = torch.rand(n, 3, generator=g) * 2 - 1
        ref = R.sdf_forward(sc.sdf_sd, sc.sdf_spec, x).numpy()
        out1 = gpu["sdf_network"].sdf(x.cuda()).cpu().numpy()
        assert out1.shape == (n, 1)
        assert np.abs(out1[:, 0] - ref[:, 0]).max() <= 2e-6, n
        full = gpu["sdf_network"](x.cuda()).cpu().numpy()
        assert full.shape == (n, 257)
        assert rel_l2(full, ref) <= 1e-5, n
        assert np.abs(full - ref).max() <= 5e-6, n


@torch.no_grad()  # the inference kernels (under grad mode the same calls return attached tensors)
def test_sdf_forward_golden(s1):
    _, gpu = s1
    g = golden("g2_sdf.npz")
    out = gpu["sdf_network"](t(g["x"]).cuda()).cpu().numpy()
    assert rel_l2(out[:, 0], g["sdf"]) <= 1e-5
    assert rel_l2(out[:256, 1:], g["feature256"]) <= 1e-5


@torch.no_grad()  # the inference kernels (under grad mode the same calls return attached tensors)
def test_sdf_empty_and_cpu_refused(s1):
    _, gpu = s1
    assert gpu["sdf_network"].sdf(torch.zeros(0, 3).cuda()).shape == (0, 1)
    with pytest.raises(RuntimeError):
        gpu["sdf_network"].sdf(torch.zeros(4, 3))


def test_camera_rays_and_sphere():
    from iron_amd.raytracer import Camera, intersect_sphere
    g = golden("g5_rays.npz")
    cam = Camera(512, 512, t(g["K"]).cuda(), t(g["W2C"]).cuda())
    o, d, dn = cam.get_rays(t(g["uv"]).cuda())
    np.testing.assert_allclose(d.cpu().numpy(), g["ray_d"], rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(dn.cpu().numpy(), g["ray_d_norm"], rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(o.cpu().numpy(), g["ray_o"], rtol=2e-6, atol=2e-7)
    m, near, far = intersect_sphere(t(g["ray_o"]).reshape(-1, 3).cuda(), t(g["ray_d"]).reshape(-1, 3).cuda(), 1.0)
    assert m.dtype == torch.bool
    assert np.array_equal(m.cpu().numpy(), g["mask"])
    np.testing.assert_allclose(near.cpu().numpy(), g["near"], rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(far.cpu().numpy(), g["far"], rtol=2e-6, atol=2e-7)


def test_ggx_golden():
    """GGXColocatedRenderer on the (dot, alpha) grid incl. clamp edges.  rel <= 1e-5 except where a
    1-ulp powf difference flips a table bin (counted, must stay rare)."""
    from iron_amd.renderer_ggx import GGXColocatedRenderer
    g = golden("g4_ggx.npz")
    r = GGXColocatedRenderer(use_cuda=True)
    prm = {k: t(g[k]).cuda() for k in ("diffuse_albedo", "specular_albedo", "specular_roughness")}
    res = r(float(g["light"]), t(g["distance"]).cuda(), t(g["normal"]).cuda(), t(g["viewdir"]).cuda(), params=prm)
    spec = res["specular_rgb"].cpu().numpy()
    np.testing.assert_allclose(spec, g["specular_rgb"], rtol=2e-5, atol=1e-30)
    diff = res["diffuse_rgb"].cpu().numpy()
    bad = np.abs(diff - g["diffuse_rgb"]) > 2e-5 * np.abs(g["diffuse_rgb"]) + 1e-30
    flips = int(bad.any(axis=-1).sum())
    assert flips <= max(2, diff.shape[0] // 500), "table-bin flips: %d of %d" % (flips, diff.shape[0])


@torch.no_grad()  # the inference kernels (under grad mode the same calls return attached tensors)
def test_fp16_overflowing_weight_selects_the_fp32_core():
    """A folded weight beyond fp16's range cannot be split for the h2 core: the library keeps only the fp32 pack for that
    network and every kernel runs it on the exact-fp32 MFMA core (never a CPU path) -- results stay at parity."""
    from oracle import iron_ref as R
    from _util import cpu_sd
    nets = scenes.build_networks("S1")
    sdf = nets["sdf_network"]
    with torch.no_grad():
        sdf.lin3.weight_g[7] *= 3.0e5   # row norm g scales the whole folded row: |w| ~ 1e4..1e5 > 65504 for some entries
    assert float((sdf.lin3.weight_g[7] * sdf.lin3.weight_v[7] / sdf.lin3.weight_v[7].norm()).detach().abs().max()) > 65504.0
    g = torch.Generator().manual_seed(5)
    x = torch.rand(300, 3, generator=g) * 2 - 1
    ref = R.sdf_forward(cpu_sd(sdf), R.SDFSpec(), x)[:, 0].numpy()
    y = sdf.cuda().sdf(x.cuda())[:, 0].cpu().numpy()
    assert np.all(np.isfinite(y))
    assert rel_l2(y, ref) <= 1e-5


def test_stage1_colour_network_with_skip():
    """RenderingNetwork(n_layers=8, skip_in=[4], multires=10, multires_view=4) -- the stage-1 colour net of
    confs/womask_iron.conf: 48 head slots (PE-10 points, PE-4 views, normals: two head ring slots) and a skip connection at hidden
    layer 4, on the h2 core (k_material_h2_skip; tests/run_f32core_check.py covers the exact-fp32 kernel).  Golden: G13 (the real
    reference).  Ragged batches around the 32-point wave tile and the 128-point workgroup must reproduce the full batch."""
    from iron_amd.fields import RenderingNetwork, SDFNetwork
    torch.manual_seed(0)
    SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5, scale=1.0, geometric_init=True,
               weight_norm=True)  # consumes the RNG like make_golden_neus.build_stage1
    net = RenderingNetwork(d_feature=256, mode="idr", d_in=9, d_out=3, d_hidden=256, n_layers=8, skip_in=[4], weight_norm=True,
                           multires=10, multires_view=4, squeeze_out=True).cuda()
    g = golden("g13_neus.npz")
    out = net(t(g["color_pts"]).cuda(), t(g["color_nrm"]).cuda(), t(g["color_view"]).cuda(), t(g["color_feat"]).cuda())
    assert out.requires_grad  # like the reference module: attached to the parameters under grad mode (iron_amd.autograd)
    out = out.detach()
    assert tuple(out.shape) == g["color_out"].shape
    err = np.abs(out.cpu().numpy() - g["color_out"]).max()
    print("stage-1 colour net: rel-L2 %.2e  max|d| %.2e" % (rel_l2(out.cpu().numpy(), g["color_out"]), err))
    assert rel_l2(out.cpu().numpy(), g["color_out"]) <= 1e-5
    n_all = out.shape[0]
    for n in (1, 33, 127, 129, min(513, n_all)):
        with torch.no_grad():
            o2 = net(t(g["color_pts"]).cuda()[:n], t(g["color_nrm"]).cuda()[:n], t(g["color_view"]).cuda()[:n], t(g["color_feat"]).cuda()[:n])
        assert torch.equal(o2, out[:n]), n   # every point is computed independently of its neighbours


def test_nerf_background_field():
    """NeRF.forward (fields.py:299-325; the stage-1 background field: 4-D points PE-10, views PE-4, skip after layer 4,
    alpha / feature / view / rgb heads) vs the reference (G13); constructor parity is part of the stage-1 state hash."""
    from iron_amd.fields import NeRF, RenderingNetwork, SDFNetwork, SingleVarianceNetwork
    from _util import golden_meta, state_hash
    torch.manual_seed(0)
    nets = {
        "sdf_network": SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5, scale=1.0,
                                  geometric_init=True, weight_norm=True),
        "color_network": RenderingNetwork(d_feature=256, mode="idr", d_in=9, d_out=3, d_hidden=256, n_layers=8, skip_in=[4],
                                          weight_norm=True, multires=10, multires_view=4, squeeze_out=True),
        "nerf": NeRF(D=8, d_in=4, d_in_view=3, W=256, multires=10, multires_view=4, output_ch=4, skips=[4], use_viewdirs=True),
        "deviation_network": SingleVarianceNetwork(0.3),
    }
    assert state_hash(nets) == golden_meta()["state_sha256_stage1"]
    g = golden("g13_neus.npz")
    nerf = nets["nerf"].cuda()
    with torch.no_grad():  # forward only: under grad mode NeRF.forward refuses rather than return detached values
        alpha, rgb = nerf(t(g["nerf_pts"]).cuda(), t(g["nerf_views"]).cuda())
    assert tuple(alpha.shape) == g["nerf_alpha"].shape and tuple(rgb.shape) == g["nerf_rgb"].shape
    ra, rc = rel_l2(alpha.cpu().numpy(), g["nerf_alpha"]), rel_l2(rgb.cpu().numpy(), g["nerf_rgb"])
    print("NeRF: alpha rel-L2 %.2e  rgb rel-L2 %.2e" % (ra, rc))
    assert ra <= 1e-5 and rc <= 1e-5
    pts, views = t(g["nerf_pts"]).cuda(), t(g["nerf_views"]).cuda()
    for n in (1, 33, 127, 129, min(513, alpha.shape[0])):   # ragged batches (k_nerf_h2: 4 waves x 32 points per workgroup)
        with torch.no_grad():
            a2, c2 = nerf(pts[:n].contiguous(), views[:n].contiguous())
        assert torch.equal(a2, alpha[:n]) and torch.equal(c2, rgb[:n]), n


def test_material_predictor_and_bulk_query():
    """render_surface.py:434-450 + the query loop of export_materials.py:181-191: values vs the oracle on surface points, and
    the bulk form is invariant to how the points are split."""
    from iron_amd.rendering_func import MaterialPredictor, query_materials
    from oracle import iron_ref as R
    from _util import oracle_scene
    cpu = scenes.build_networks("S1")
    sc = oracle_scene(cpu)
    nets = {k: v.cuda() for k, v in cpu.items()}
    gen = torch.Generator().manual_seed(5)
    pts = torch.nn.functional.normalize(torch.randn(3001, 3, generator=gen), dim=-1) * 0.5
    mp = MaterialPredictor(nets["sdf_network"], nets)
    kd, ks, rough = mp(pts.cuda())
    _, feat, grad = R.sdf_get_all(sc.sdf_sd, sc.sdf_spec, pts)
    nrm = grad / (grad.norm(dim=-1, keepdim=True) + 1e-10)
    ref = R.get_materials(sc.nets, pts, nrm, feat)
    for got, key in ((kd, "diffuse_albedo"), (ks, "specular_albedo"), (rough, "specular_roughness")):
        want = ref[key].numpy()
        err = np.abs(got.cpu().numpy().reshape(want.shape) - want).max()
        assert err <= 2e-5, (key, err)
    big = pts.cuda().repeat(120, 1)[:350001]
    a = query_materials(mp, big, max_num_pts=320000)
    b = query_materials(mp, big, max_num_pts=100003)
    assert a.shape == (350001, 7) and torch.equal(a, b)
    assert torch.equal(a[:3001, 0:3], kd) and torch.equal(a[:3001, 6:7], rough.reshape(-1, 1))
    assert float(a[:, 6].min()) >= 0.01 and torch.isfinite(a).all()


def test_smith_g1_and_sample_pdf_operators():
    """The two small operators of the surface that the fused kernels otherwise absorb: smithG1 (renderer_ggx.py:12-16) and
    sample_pdf (renderer.py:45-75), against their torch expressions / the oracle."""
    from iron_amd.renderer import sample_pdf
    from iron_amd.renderer_ggx import smithG1
    from oracle import neus_ref as N
    gen = torch.Generator().manual_seed(9)
    c = torch.rand(500, 1, generator=gen).clamp(1e-5, 0.99999)
    al = torch.rand(500, 1, generator=gen) * 2 + 1e-4
    got = smithG1(c.cuda(), al.cuda()).cpu()
    root = al * (torch.sqrt(1.0 - c * c) / (c + 1e-10))
    want = 2.0 / (1.0 + torch.hypot(root, torch.ones_like(root)))
    assert got.shape == want.shape and float((got - want).abs().max()) <= 2e-6
    assert smithG1(c.cuda(), torch.tensor([[0.3]]).cuda()).shape == (500, 1)      # broadcasting like the tensor expression
    bins = torch.sort(torch.rand(37, 24, generator=gen) * 3, dim=-1)[0]
    w = torch.rand(37, 23, generator=gen) ** 3
    w[5] = 0.0                                                                      # a flat row: the 1e-5 floor decides
    got = sample_pdf(bins.cuda(), w.cuda(), 16, det=True).cpu()
    want = N.sample_pdf(bins, w, 16, det=True)
    err = (got - want).abs()
    # a position that falls into a section of tiny probability is (u - cdf_below) / denom with denom ~1e-4: the 6e-8 by which
    # two summation orders of the CDF differ becomes ~1e-4 of that section's width; everywhere else the agreement is at rounding
    assert float(err.max()) <= 2e-4 and float(err.flatten().kthvalue(int(0.97 * err.numel()))[0]) <= 2e-6, (float(err.max()), int(err.argmax()))
    torch.manual_seed(3)
    r = sample_pdf(bins.cuda(), w.cuda(), 64, det=False).cpu()                     # random draws: inside the bins, right shape
    assert r.shape == (37, 64) and bool((r >= bins[:, :1] - 1e-6).all()) and bool((r <= bins[:, -1:] + 1e-6).all())


def test_parameter_writes_and_the_packed_copy():
    """The kernels run on a packed device copy of the parameters.  In-place ops on a parameter re-pack it automatically (version
    counter); a write through `.data` does not bump the counter and needs invalidate() -- both behaviours are part of the contract
    (INTEGRATION.md)."""
    net = scenes.build_networks("S0")["sdf_network"].cuda()
    x = torch.rand(256, 3, device="cuda") - 0.5
    y0 = net.sdf(x).clone()
    with torch.no_grad():
        net.lin8.bias.add_(0.25)                       # an ordinary in-place update: seen
    y1 = net.sdf(x)
    assert float((y1 - y0 - 0.25).abs().max()) <= 1e-6
    net.lin8.bias.data[0] += 0.5                       # through .data: no version bump, the packed copy is stale ...
    assert torch.equal(net.sdf(x), y1)
    net.invalidate()                                   # ... until it is dropped
    assert float((net.sdf(x) - y1 - 0.5).abs().max()) <= 1e-6


def test_get_all_refuses_a_non_leaf_input_and_moved_parameters():
    net = scenes.build_networks("S0")["sdf_network"].cuda()
    leaf = (torch.rand(64, 3, device="cuda") - 0.5).requires_grad_(True)
    net.get_all(leaf, is_training=True)                # a leaf with requires_grad (the reference sets the flag in place): fine
    with pytest.raises(NotImplementedError):
        net.get_all(leaf * 1.0, is_training=True)      # a computed x would silently lose its gradient
    sdf, _, _ = net.get_all(leaf.detach(), is_training=True)
    with torch.no_grad():
        net.lin0.bias.add_(1e-3)                       # parameters moved between forward and backward
    with pytest.raises(RuntimeError):
        sdf.sum().backward()


@torch.no_grad()
@pytest.mark.parametrize("scene", ["S1", "S3"])
def test_get_all_reverse_mode_vs_forward_mode_and_fp64(scene):
    """iron_sdf_get_all with the tape workspace runs the reverse-mode kernel (forward + one reverse sweep over transposed weights,
    csrc/getall_rev.hip); without it, the forward-mode kernel (value + three tangents).  Same function: sdf and features to 2e-6,
    the gradient to 1e-5 of its scale, on ragged batch sizes (1 ... 5000: partial waves, partial workgroups, several passes per
    workgroup) -- and both against an fp64 torch evaluation of models/fields.py:120-137 (autograd) on the same weights."""
    import ctypes as C
    from iron_amd import _lib
    from oracle import iron_ref as R
    nets = scenes.build_networks(scene)
    net = nets["sdf_network"].cuda()
    lib = _lib.load()
    h = net.hip_net()
    g = torch.Generator().manual_seed(11)
    assert lib.iron_sdf_get_all_workspace_bytes(h.handle, 1000) > 0, "no reverse stream for the reference-shaped SDF network"
    for n in (1, 33, 127, 129, 1000, 5000):
        x = (torch.rand(n, 3, generator=g) * 1.6 - 0.8).cuda()
        outs = []
        for rev in (True, False):
            sdf = torch.full((n,), 7.0, device="cuda"); feat = torch.full((n, 256), 7.0, device="cuda"); grad = torch.full((n, 3), 7.0, device="cuda")
            nbytes = lib.iron_sdf_get_all_workspace_bytes(h.handle, n) if rev else 0
            ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device="cuda")
            _lib.check(lib.iron_sdf_get_all(h.handle, x.data_ptr(), n, sdf.data_ptr(), feat.data_ptr(), grad.data_ptr(),
                                            ws.data_ptr() if rev else None, nbytes, _lib.stream_ptr(x.device)))
            torch.cuda.synchronize()
            outs.append((sdf.cpu(), feat.cpu(), grad.cpu()))
        (s_r, f_r, g_r), (s_f, f_f, g_f) = outs
        gscale = float(g_f.abs().max())
        assert float((s_r - s_f).abs().max()) <= 2e-6, n
        assert float((f_r - f_f).abs().max()) <= 2e-6 * max(1.0, float(f_f.abs().max())), n
        assert float((g_r - g_f).abs().max()) <= 1e-5 * max(1.0, gscale), (n, float((g_r - g_f).abs().max()), gscale)
        if n == 5000:   # fp64 autograd of the same network (oracle restatement of fields.py:82-137)
            from _util import cpu_sd
            sd64 = {k: v.double() for k, v in cpu_sd(nets["sdf_network"]).items()}
            x64 = x.cpu().double().requires_grad_(True)
            with torch.enable_grad():
                y = R.sdf_forward(sd64, R.SDFSpec(), x64)
                (g64,) = torch.autograd.grad(y[:, 0].sum(), x64)
            e_rev = float((g_r.double() - g64).norm() / g64.norm())
            e_fwd = float((g_f.double() - g64).norm() / g64.norm())
            print("   %s get_all gradient rel-L2 vs fp64 autograd: reverse-mode %.2e, forward-mode %.2e" % (scene, e_rev, e_fwd))
            assert e_rev <= max(2e-6, 2.0 * e_fwd), (e_rev, e_fwd)
            assert float((s_r.double() - y[:, 0].detach()).abs().max()) <= 2e-6
