"""SURVEY 8 row f-2 on the GPU: the backward passes of libiron_train.so behind iron_amd.autograd, each operator against
torch.autograd over the oracle restatement (differentiable torch code) on the same inputs, and the whole training-mode
render against the real reference's parameter gradients (golden G14)."""
import numpy as np
import pytest
import torch

from _util import cpu_sd, golden, golden_meta, t, tables

pytestmark = pytest.mark.gpu

NETS = ("sdf_network", "diffuse_albedo_network", "specular_albedo_network", "specular_roughness_network")


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64).reshape(-1), np.asarray(b, dtype=np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _compare_param_grads(module, leaf_sd, tol, tag, tol_for=None):
    """tol_for: {parameter-name prefix: tolerance} overrides for individual tensors (every other tensor is held to `tol`)."""
    worst = 0.0
    base_tol = tol
    for name, p in module.named_parameters():
        tol = base_tol
        for prefix, t_ in (tol_for or {}).items():
            if name.startswith(prefix):
                tol = t_
        assert p.grad is not None, name
        ref = leaf_sd[name].grad
        if ref is None:  # torch found no path to this parameter (e.g. the last bias from a gradient-only loss): ours must be 0
            assert float(p.grad.abs().max()) == 0.0, (tag, name)
            p.grad = None
            continue
        r = _rel(p.grad.cpu().numpy(), ref.numpy())
        if name.endswith("weight_g") and r > tol:
            # d/dg_i = <dW_i, v_i/|v_i|> is a projection of the effective-weight gradient dW (what the GEMMs produce) that can cancel
            # by orders of magnitude (lin0 of the PE-10 colour net: |d/dg| ~ 1e-3 |dW_i|), so its own norm is the wrong yardstick:
            # the split-fp16 GEMM carries 2^-22 per operand, relative to dW.  |dW_i| = |d/dv_i| |v_i| / g_i up to that projection.
            v = dict(module.named_parameters())[name.replace("weight_g", "weight_v")].detach().cpu()
            gv = leaf_sd[name.replace("weight_g", "weight_v")].grad
            dw_rows = gv.norm(dim=1, keepdim=True) * v.norm(dim=1, keepdim=True) / dict(module.named_parameters())[name].detach().cpu().abs()
            r = float(((p.grad.cpu() - ref).abs() / dw_rows.clamp_min(1e-30)).max())
        # a parameter whose gradient is pure rounding noise (e.g. zero-initialised PE columns' norm direction) is compared in
        # absolute terms against the largest gradient of the network
        if r > worst:
            worst = r
            _compare_param_grads.last = "%s |ref| %.2e" % (name, float(ref.norm()))
        assert r <= tol or float(ref.abs().max()) <= 1e-9, (tag, name, r)
        p.grad = None
    return worst


def test_sdf_get_all_backward_vs_autograd():
    """d(loss)/d(theta) through sdf, feature AND the normal (second order): iron_sdf_backward vs torch double backward."""
    from iron_amd import scenes
    from oracle import iron_ref as R
    from oracle import train_ref as T
    nets = scenes.build_networks("S1")
    net = nets["sdf_network"].cuda()
    gen = torch.Generator().manual_seed(3)
    for n, parts in ((301, "sfg"), (64, "g"), (64, "s"), (1, "sfg"), (5003, "sfg")):  # 5003: the split-K weight-gradient path
        x = torch.rand(n, 3, generator=gen) * 1.2 - 0.6
        a, B, Cc = torch.randn(n, 1, generator=gen), torch.randn(n, 256, generator=gen) * 0.1, torch.randn(n, 3, generator=gen)
        sd = T.leaf_state(cpu_sd(nets["sdf_network"]))
        y, f, g = T.sdf_get_all_train(sd, R.SDFSpec(), x)
        loss = 0
        if "s" in parts: loss = loss + (y * a).sum()
        if "f" in parts: loss = loss + (f * B).sum()
        if "g" in parts: loss = loss + (g * Cc).sum() + (g.norm(dim=-1) - 1).pow(2).sum()
        loss.backward()
        y2, f2, g2 = net.get_all(x.cuda(), is_training=True)
        assert y2.requires_grad and g2.requires_grad
        l2 = 0
        if "s" in parts: l2 = l2 + (y2 * a.cuda()).sum()
        if "f" in parts: l2 = l2 + (f2 * B.cuda()).sum()
        if "g" in parts: l2 = l2 + (g2 * Cc.cuda()).sum() + (g2.norm(dim=-1) - 1).pow(2).sum()
        l2.backward()
        w = _compare_param_grads(net, sd, 2e-4, "sdf n=%d %s" % (n, parts))
        print("sdf backward n=%d parts=%s worst rel-L2 %.2e" % (n, parts, w))


def test_backward_is_additive_over_points_across_chunks():
    """Size-independent property at sizes the oracle does not reach: the parameter gradient of a sum over points equals the sum
    of the gradients of its parts.  70 001 SDF points span two 65 536-point chunks (weight gradients accumulated across
    chunks, split-K inside each); 140 001 material points span two 131 072-point chunks."""
    from iron_amd import scenes
    nets = scenes.build_networks("S1")
    sdf = nets["sdf_network"].cuda()
    gen = torch.Generator().manual_seed(17)

    def grads_of(mod, loss_fn, sl):
        for p in mod.parameters():
            p.grad = None
        loss_fn(sl).backward()
        return [p.grad.detach().clone() for p in mod.parameters()]

    n = 70001
    x = (torch.rand(n, 3, generator=gen) * 1.6 - 0.8).cuda()
    a, B, Cc = torch.randn(n, 1, generator=gen).cuda(), (torch.randn(n, 256, generator=gen) * 0.05).cuda(), torch.randn(n, 3, generator=gen).cuda()

    def sdf_loss(sl):
        y, f, g = sdf.get_all(x[sl], is_training=True)
        return (y * a[sl]).sum() + (f * B[sl]).sum() + (g * Cc[sl]).sum()

    whole = grads_of(sdf, sdf_loss, slice(0, n))
    h1 = grads_of(sdf, sdf_loss, slice(0, 30000))
    h2 = grads_of(sdf, sdf_loss, slice(30000, n))
    w = max(_rel(a_.cpu().numpy(), (b_ + c_).cpu().numpy()) for a_, b_, c_ in zip(whole, h1, h2))
    print("sdf additivity over 2 chunks: worst rel-L2 %.2e" % w)
    assert w <= 2e-5

    net = nets["specular_albedo_network"].cuda()
    n = 140001
    pts = (torch.rand(n, 3, generator=gen) * 1.2 - 0.6).cuda()
    nrm = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1).cuda()
    ft = (torch.randn(n, 256, generator=gen) * 0.3).cuda()
    up = torch.randn(n, 3, generator=gen).cuda()

    def mat_loss(sl):
        return (net(pts[sl], nrm[sl], None, ft[sl]) * up[sl]).sum()

    whole = grads_of(net, mat_loss, slice(0, n))
    h1 = grads_of(net, mat_loss, slice(0, 50000))
    h2 = grads_of(net, mat_loss, slice(50000, n))
    w = max(_rel(a_.cpu().numpy(), (b_ + c_).cpu().numpy()) for a_, b_, c_ in zip(whole, h1, h2))
    print("material additivity over 2 chunks: worst rel-L2 %.2e" % w)
    assert w <= 2e-5


def test_sdf_gradient_eikonal_term():
    """sdf_network.gradient(x) under grad mode (fields.py:106-118, the eikonal regulariser of render_surface.py) is attached."""
    from iron_amd import scenes
    from oracle import iron_ref as R
    from oracle import train_ref as T
    nets = scenes.build_networks("S1")
    net = nets["sdf_network"].cuda()
    x = torch.rand(200, 3, generator=torch.Generator().manual_seed(4)) * 2 - 1
    sd = T.leaf_state(cpu_sd(nets["sdf_network"]))
    _, _, g = T.sdf_get_all_train(sd, R.SDFSpec(), x)
    ((g.norm(dim=-1) - 1) ** 2).mean().backward()
    g2 = net.gradient(x.cuda())
    ((g2.norm(dim=-1) - 1) ** 2).mean().backward()
    w = _compare_param_grads(net, sd, 2e-4, "eikonal")
    print("eikonal worst rel-L2 %.2e" % w)
    with torch.no_grad():
        assert not net.gradient(x.cuda()).requires_grad
        plain = net(x.cuda())
    # forward() / sdf() follow the same rule: attached under grad mode (a loss on the raw SDF values trains the network)
    out = net(x.cuda())
    assert out.requires_grad and net.sdf(x.cuda()).requires_grad and not plain.requires_grad
    assert float((out.detach() - plain).abs().max()) <= 2e-6
    sd2 = T.leaf_state(cpu_sd(nets["sdf_network"]))
    R.sdf_forward(sd2, R.SDFSpec(), x)[:, :8].square().sum().backward()
    out[:, :8].square().sum().backward()
    print("forward() attached: worst rel-L2 %.2e" % _compare_param_grads(net, sd2, 2e-4, "sdf forward"))


@pytest.mark.parametrize("name", ["diffuse_albedo_network", "specular_albedo_network", "specular_roughness_network", "stage1_color"])
def test_render_network_backward_vs_autograd(name):
    from iron_amd import scenes
    from iron_amd.fields import RenderingNetwork
    from oracle import iron_ref as R
    from oracle import neus_ref as N
    from oracle import train_ref as T
    if name == "stage1_color":
        torch.manual_seed(7)
        mod = RenderingNetwork(d_feature=256, mode="idr", d_in=9, d_out=3, d_hidden=256, n_layers=8, skip_in=[4], weight_norm=True,
                               multires=10, multires_view=4, squeeze_out=True)
        spec = N.COLOR_SPEC
    else:
        mod = scenes.build_networks("S1")[name]
        spec = R.GGX_SPECS[name]
    sd = T.leaf_state(cpu_sd(mod))
    net = mod.cuda()
    gen = torch.Generator().manual_seed(11)
    n = 4517  # >= 4096 rows: the split-K weight-gradient path (with a ragged tail)
    ins = [torch.rand(n, 3, generator=gen) * 1.2 - 0.6, torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1),
           torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1), torch.randn(n, 256, generator=gen) * 0.3]
    use_view = mod.mode in ("idr", "no_normal")
    up = torch.randn(n, mod.d_out, generator=gen)
    cpu_in = [v.clone().requires_grad_(True) for v in ins]
    out = R.rendering_forward(sd, spec, cpu_in[0], cpu_in[1], cpu_in[2] if use_view else None, cpu_in[3])
    (out * up).sum().backward()
    gpu_in = [v.cuda().requires_grad_(True) for v in ins]
    out2 = net(gpu_in[0], gpu_in[1], gpu_in[2] if use_view else None, gpu_in[3])
    assert out2.requires_grad
    assert _rel(out2.detach().cpu().numpy(), out.detach().numpy()) <= 1e-5
    (out2 * up.cuda()).sum().backward()
    # the 8-layer colour net: layers 2..8 agree to 2e-7 ... 9e-7 (tests/diag_stage1_grads.py); ONE of the 1.16 M layer-1 pre-activations of
    # this batch lies within rounding of 0 and the backward's recomputed forward (split-fp16 GEMM, tests/test_gpu_gemm.py: 1-3e-7 per
    # product) lands on the other side of ReLU's kink than MKL's: that single element is 6.7e-4 of |dZ_1|, hence of lin1 / lin0's gradients
    # -> only lin0 / lin1 of that net are allowed the kink element; everything else stays at 2e-4
    w = _compare_param_grads(net, sd, 2e-4, name, tol_for={"lin0.": 1.5e-3, "lin1.": 1.5e-3} if name == "stage1_color" else None)
    for i, what in enumerate(("points", "normals", "view_dirs", "features")):
        if cpu_in[i].grad is None:
            assert gpu_in[i].grad is None or float(gpu_in[i].grad.abs().max()) == 0.0, what
            continue
        r = _rel(gpu_in[i].grad.cpu().numpy(), cpu_in[i].grad.numpy())
        print("   d/d%s rel-L2 %.2e" % (what, r))
        w = max(w, r)
        assert r <= 2e-4, (name, what, r)
    print("%s backward worst rel-L2 %.2e (parameter side: %s)" % (name, w, getattr(_compare_param_grads, "last", "")))


def test_ggx_backward_vs_autograd():
    from iron_amd.renderer_ggx import GGXColocatedRenderer
    from oracle import iron_ref as R
    mt, md = tables()
    gen = torch.Generator().manual_seed(13)
    n = 4099
    nrm = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1)
    vd = torch.nn.functional.normalize(nrm + 0.8 * torch.randn(n, 3, generator=gen), dim=-1)
    vd[:50] = -vd[:50]            # back-facing: cos clamped at 1e-5, no gradient through the clamp
    ins = {"light": torch.tensor(31.0), "distance": torch.rand(n, 1, generator=gen) * 2 + 0.5, "normal": nrm, "viewdir": vd,
           "kd": torch.rand(n, 3, generator=gen), "ks": torch.rand(n, 3, generator=gen) * 0.5,
           "rough": torch.rand(n, 1, generator=gen) * 0.6 + 0.01}
    ins["rough"][:20] = 5e-5      # below the alpha clamp
    ups = [torch.randn(n, 3, generator=gen) for _ in range(3)]

    def run(dev, fn):
        v = {k: x.clone().to(dev).requires_grad_(True) for k, x in ins.items()}
        out = fn(v)
        loss = sum((out[k] * u.to(dev)).sum() for k, u in zip(("diffuse_rgb", "specular_rgb", "rgb"), ups))
        loss.backward()
        return {k: x.grad.detach().cpu().numpy() for k, x in v.items()}

    ref = run("cpu", lambda v: R.ggx_colocated(v["light"], v["distance"], v["normal"], v["viewdir"],
                                               {"diffuse_albedo": v["kd"], "specular_albedo": v["ks"], "specular_roughness": v["rough"]}, mt, md))
    rend = GGXColocatedRenderer(use_cuda=True)
    got = run("cuda", lambda v: rend(v["light"], v["distance"], v["normal"], v["viewdir"],
                                     {"diffuse_albedo": v["kd"], "specular_albedo": v["ks"], "specular_roughness": v["rough"]}))
    for k in ins:
        r = _rel(got[k], ref[k])
        print("ggx d/d%s rel-L2 %.2e" % (k, r))
        assert r <= 1e-4, (k, r)
    # clamped inputs (cos outside [1e-5, 0.99999], roughness below 1e-4) carry exactly no gradient, as in torch
    assert int((ref["rough"] == 0).sum()) >= 20 and float(np.abs(got["rough"][ref["rough"] == 0]).max()) == 0.0
    dead = np.all(ref["normal"] == 0, axis=-1)
    assert int(dead.sum()) >= 10 and float(np.abs(got["normal"][dead]).max()) == 0.0


def _check_against_golden(nets, g, tol_n=5e-4, tol_s=2e-3):
    n, worst_n, worst_s, bad = 0, 0.0, 0.0, []
    for name in NETS:
        for pname, p in nets[name].named_parameters():
            key = "%s/%s" % (name, pname)
            assert p.grad is not None, key
            gr = p.grad.reshape(-1).double().cpu().numpy()
            ref_n = float(g["gnorm:" + key])
            en = abs(np.linalg.norm(gr) - ref_n) / max(ref_n, 1e-12)
            ref_s = g["gsample:" + key]
            idx = np.concatenate([np.arange(min(16, gr.size)), np.linspace(0, gr.size - 1, 32).astype(np.int64)])
            es = float(np.abs(gr[idx] - ref_s).max() / max(np.abs(ref_s).max(), 1e-12))
            worst_n, worst_s = max(worst_n, en), max(worst_s, es)
            if en > tol_n or es > tol_s:
                bad.append((key, en, es))
            n += 1
    assert not bad, bad
    return n, worst_n, worst_s


def _grads_vs_fp64_reference(nets, g, g64_prefix_n, g64_prefix_s, floor_of, base_n=2e-3, base_s=4e-3, only=None):
    """Every parameter tensor's gradient against the reference's fp64 run, tolerance = max(base, 1.5 x the reference's own
    fp32-vs-fp64 discrepancy for THAT tensor) -- the conditioning floor recorded next to the golden (make_golden_train.py)."""
    worst = {"n": 0.0, "s": 0.0, "fn": 0.0, "fs": 0.0}
    bad, n = [], 0
    for name in NETS:
        if only is not None and name not in only:
            continue
        for pname, p in nets[name].named_parameters():
            key = "%s/%s" % (name, pname)
            gr = p.grad.reshape(-1).double().cpu().numpy()
            idx = np.concatenate([np.arange(min(16, gr.size)), np.linspace(0, gr.size - 1, 32).astype(np.int64)])
            ref_n, ref_s = float(g[g64_prefix_n + key]), g[g64_prefix_s + key]
            fn, fs = floor_of(key)
            en = abs(np.linalg.norm(gr) - ref_n) / max(ref_n, 1e-12)
            es = float(np.abs(gr[idx] - ref_s).max() / max(np.abs(ref_s).max(), 1e-12))
            for k, v in (("n", en), ("s", es), ("fn", fn), ("fs", fs)):
                worst[k] = max(worst[k], v)
            if en > max(base_n, 1.5 * fn) or es > max(base_s, 1.5 * fs):
                bad.append((key, "norm %.2e (floor %.2e)" % (en, fn), "entries %.2e (floor %.2e)" % (es, fs)))
            n += 1
    return n, worst, bad


def _render_edges_training(g, dem, scene="S1"):
    from iron_amd import scenes
    from iron_amd.raytracer import Camera, RayTracer, render_camera
    from iron_amd.renderer_ggx import GGXColocatedRenderer
    from iron_amd.rendering_func import make_render_fn
    nets = {k: v.cuda() for k, v in scenes.build_networks(scene).items()}
    cam = Camera(int(g["W"]), int(g["H"]), t(g["K"]).cuda(), t(g["W2C"]).cuda())
    res = render_camera(cam, nets["sdf_network"], RayTracer(), nets, make_render_fn(GGXColocatedRenderer(use_cuda=True)),
                        fill_holes=False, handle_edges=True, is_training=True, depth_edge_mask=dem.cuda())
    return nets, res


def test_g15_training_render_with_edge_sampling():
    """render_camera(handle_edges=True, is_training=True): the setting render_surface.py trains with.  96x96 view of S1, 122
    edge pixels (blend weights + both side colours in the graph), behind the fixture's depth-edge mask; vs the REAL
    reference (golden G15 = its fp32 run, g15_floor_fp64 = its fp64 run and the fp32-vs-fp64 discrepancy per tensor)."""
    g, f = golden("g15_train_edges_S1.npz"), golden("g15_floor_fp64.npz")
    nets, res = _render_edges_training(g, t(g["depth_edge_mask_input"]))
    assert np.array_equal(res["edge_mask"].cpu().numpy(), g["edge_mask"])
    assert np.array_equal(res["convergent_mask"].cpu().numpy(), g["convergent_mask"])
    col = res["color"].detach().cpu().numpy()
    print("G15 forward: colour rel-L2 %.2e max|d| %.2e" % (_rel(col, g["color"]), np.abs(col - g["color"]).max()))
    assert _rel(col, g["color"]) <= 2e-4
    wt = t(g["loss_weights"]).cuda()
    loss = (res["color"] * wt).sum() + 0.1 * (res["normal"] * wt).sum()
    assert abs(loss.item() - float(g["loss"])) <= 2e-4 * abs(float(g["loss"]))
    loss.backward()
    floor = lambda key: (float(f["floor_n:" + key]), float(f["floor_s:" + key]))
    # The three material networks: tolerance max(2e-3, 1.5 x floor), measured 1e-6 ... 5e-5.
    n_m, w_m, bad = _grads_vs_fp64_reference(nets, f, "gnorm:", "gsample:", floor, only=NETS[1:])
    assert not bad, bad
    # The SDF network of THIS fixture is not defined by the algorithm to better than a few per cent: 5 of the 122 edge points are
    # candidates that were already "found" before walking, so they project onto their pixel CENTRE and x = h / 0.707 = 0 up to
    # the rounding of the projection; torch.clamp(x, 0, 1) passes gradient for x >= +0 and none below, so whether such a pixel's
    # blend weight has a gradient at all -- each ~100 of a total |dB/d sdf| ~ 500 -- is decided by 1e-6 px of rounding (found
    # with tests/diag_g15_dump.py: one of them, pixel 4821, has the gradient in this path and on the GPU box's EPYC host CPU,
    # not in the build container's Xeon run that made the fixture).  The bound here is therefore loose; the same render with
    # those pixels and the other rounding-decided ones out of the loss is held to the floor in the next test, and the C3 size
    # itself (G17, 2359 edge pixels) in the one after.
    n, wn, ws = _check_against_golden(nets, g, tol_n=3e-2, tol_s=6e-2)
    print("G15: material nets vs ref64 worst |norm| %.2e entries %.2e (floors %.2e / %.2e); all %d tensors vs ref32 worst |norm| %.2e "
          "entries %.2e" % (w_m["n"], w_m["s"], w_m["fn"], w_m["fs"], n, wn, ws))
    assert n == golden_meta()["n_param_tensors_train_golden"]


def test_g15s_edge_sampling_gradients_on_the_well_defined_pixels():
    """G15 with the pixels whose contribution is decided by rounding taken out of the loss -- found on the REFERENCE side by
    make_golden_train.py --stable: edge points projecting onto the clamp corner x = 0 (or near the rim x -> 1), side rays
    grazing at |n.d| < 0.1, walks the reference's own fp32 / fp64 runs place > 1e-5 apart, and every pixel whose colour moves
    by > 2e-5 when the SDF weights are nudged by one rounding error (6 realisations).  The mask is a fixture INPUT; golden
    values are the reference's fp32 and fp64 gradients of the masked loss.  Tolerance: max(2e-3 | 4e-3, 1.5 x fp32-vs-fp64 floor)."""
    g0, g = golden("g15_train_edges_S1.npz"), golden("g15s_train_edges_stable_S1.npz")
    nets, res = _render_edges_training(g0, t(g0["depth_edge_mask_input"]))
    wt = (t(g0["loss_weights"]) * t(g["stable_pixel_mask"]).float()[..., None]).cuda()
    loss = (res["color"] * wt).sum() + 0.1 * (res["normal"] * wt).sum()
    assert abs(loss.item() - float(g["loss_fp64"])) <= 2e-4 * abs(float(g["loss_fp64"]))
    loss.backward()
    floor = lambda key: (float(g["floor_n:" + key]), float(g["floor_s:" + key]))
    n, w, bad = _grads_vs_fp64_reference(nets, g, "gnorm64:", "gsample64:", floor)
    print("G15s (%d edge pixels masked): %d tensors vs ref64 worst |norm| %.2e entries %.2e; the reference's fp32 vs fp64: %.2e / %.2e" % (
        int(g["n_masked_edge_pixels"]), n, w["n"], w["s"], w["fn"], w["fs"]))
    assert not bad, bad
    assert n == golden_meta()["n_param_tensors_train_golden"]


def test_g17_c3_training_step_at_512_vs_reference():
    """BASELINE config C3 at its real size: S1, 512x512, silhouette edge sampling in the graph (87 123 hits, 2 359 edge pixels;
    the depth-edge mask is the fixture's input as in G8 / G15), loss.backward() through the HIP operators -- against the REAL
    reference's fp64 run (golden G17, make_golden_train.py --c3: 60 s fp32 + 180 s fp64 on the build container), per tensor
    within max(2e-3 | 4e-3, 1.5 x the reference's fp32-vs-fp64 discrepancy)."""
    g = golden("g17_train_c3_S1_512.npz")
    size = int(g["W"])
    unpack = lambda key: torch.from_numpy(np.unpackbits(g[key])[: size * size].reshape(size, size).astype(bool))
    nets, res = _render_edges_training(g, unpack("depth_edge_mask_input_bits"))
    conv, edge = res["convergent_mask"].cpu(), res["edge_mask"].cpu()
    flips = int((conv != unpack("convergent_mask_bits")).sum())
    eflips = int((edge != unpack("edge_mask_bits")).sum())
    assert flips <= 4 and eflips <= 4, (flips, eflips)
    col = res["color"].detach().cpu().numpy()[::4, ::4]
    both = (conv & unpack("convergent_mask_bits")).numpy()[::4, ::4]
    rel = _rel(col[both], g["color_sub4"][both])
    yy, xx = np.meshgrid(np.arange(size, dtype=np.float64), np.arange(size, dtype=np.float64), indexing="ij")
    wt = np.stack([0.2 + 0.5 * np.sin(0.37 * xx + 0.11 * yy + 1.3 * c) + 0.3 * np.cos(0.05 * xx - 0.23 * yy) for c in range(3)], axis=-1)
    wt = torch.from_numpy(wt.astype(np.float32)).cuda()
    loss = (res["color"] * wt).sum() + 0.1 * (res["normal"] * wt).sum()
    loss.backward()
    floor = lambda key: (float(g["floor_n:" + key]), float(g["floor_s:" + key]))
    n, w, bad = _grads_vs_fp64_reference(nets, g, "gnorm64:", "gsample64:", floor)
    print("G17 (C3, 512x512): hits %d (reference %d), edge pixels %d (%d), mask flips %d / %d, colour rel-L2 %.2e, loss %.4f (ref64 %.4f); "
          "%d tensors vs ref64 worst |norm| %.2e entries %.2e; the reference's fp32 vs fp64: %.2e / %.2e" % (
              int(conv.sum()), int(g["n_hits"]), int(edge.sum()), int(g["n_edge"]), flips, eflips, rel, loss.item(), float(g["loss_fp64"]),
              n, w["n"], w["s"], w["fn"], w["fs"]))
    assert rel <= 2e-4
    assert abs(loss.item() - float(g["loss_fp64"])) <= 3e-4 * abs(float(g["loss_fp64"]))
    assert not bad, bad
    assert n == golden_meta()["n_param_tensors_train_golden"]


def test_edge_pixel_backward_with_the_oracles_edge_points():
    """render_edge_pixels(is_training=True) (raytracer.py:665-729) isolated from the walk: the oracle's edge points are
    injected into the product's results, then the gradient of the edge-pixel colour loss is compared with torch.autograd
    over the oracle (itself pinned to the reference by G15, tests/test_oracle_train.py)."""
    from iron_amd import scenes
    from iron_amd.raytracer import Camera, RayTracer, raytrace_camera, render_edge_pixels, render_normal_and_color
    from iron_amd.renderer_ggx import GGXColocatedRenderer
    from iron_amd.rendering_func import make_render_fn
    from oracle import iron_ref as R
    from oracle import train_ref as T
    g = golden("g15_train_edges_S1.npz")
    mt, md = tables()
    # the edge pixels whose blend-weight gradient is decided by rounding (un-walked candidates sitting on torch.clamp's corner, grazing
    # side rays, ...: make_golden_train.py --stable, see test_g15s_...) are left out of the loss: with them in, the oracle run on another
    # host CPU disagrees with itself by per cent
    wt, em = t(g["loss_weights"]), t(g["edge_mask"]).bool() & t(golden("g15s_train_edges_stable_S1.npz")["stable_pixel_mask"]).bool()
    cpu_nets = scenes.build_networks("S1")
    sd = {k: T.leaf_state(cpu_sd(cpu_nets[k])) for k in NETS}
    sc = R.Scene(sd["sdf_network"], R.SDFSpec(), {k: (sd[k], R.GGX_SPECS[k]) for k in R.GGX_SPECS}, golden_meta()["light"], mt, md)
    torch.set_num_threads(8)
    ref = T.render_camera_edges_train(sc, R.CameraSpec(int(g["W"]), int(g["H"]), t(g["K"]), t(g["W2C"])), t(g["depth_edge_mask_input"]))
    (ref["color"] * wt)[em].sum().backward()
    nets = {k: v.cuda() for k, v in scenes.build_networks("S1").items()}
    cam = Camera(int(g["W"]), int(g["H"]), t(g["K"]).cuda(), t(g["W2C"]).cuda())
    fn = make_render_fn(GGXColocatedRenderer(use_cuda=True))
    res = raytrace_camera(cam, nets["sdf_network"], RayTracer(), max_num_rays=50000, fill_holes=False, detect_edges=True,
                          depth_edge_mask=t(g["depth_edge_mask_input"]).cuda())
    assert np.array_equal(np.sort(res["edge_pixel_idx"].cpu().numpy()), np.sort(ref["edge_pixel_idx"].numpy()))
    with torch.no_grad():
        res["edge_points"] = ref["edge_points"].detach().cuda()
        res["edge_uv"] = cam.project(res["edge_points"])
        res["edge_pixel_idx"] = ref["edge_pixel_idx"].cuda()
    render_normal_and_color(res, nets["sdf_network"], nets, fn, is_training=True)
    render_edge_pixels(res, cam, nets["sdf_network"], RayTracer(), nets, fn, is_training=True)
    (res["color"] * wt.cuda())[em.cuda()].sum().backward()
    worst = {}
    for name in NETS:
        for pname, p in nets[name].named_parameters():
            r = sd[name][pname].grad
            if r is None or float(r.abs().max()) < 1e-9:
                continue
            worst[name] = max(worst.get(name, 0.0), _rel(p.grad.cpu().numpy(), r.numpy()))
    print("edge-pixel backward, same edge points:", {k: "%.1e" % v for k, v in worst.items()})
    # the side rays graze the surface (|n.v| ~ 0.05-0.2): the specular lobe's derivative amplifies the 1e-6 of the traced roots
    assert worst["sdf_network"] <= 5e-3
    assert max(worst.values()) <= 4e-2


def test_g14_training_render_matches_reference_gradients():
    """render_camera(is_training=True) + loss.backward() on the GPU vs the REAL reference (golden G14: S1, 32x32 crop):
    colour, normal, loss, and the gradient of all 72 parameter tensors (norm + sampled entries)."""
    from iron_amd import scenes
    from iron_amd.raytracer import Camera, RayTracer, render_camera
    from iron_amd.renderer_ggx import GGXColocatedRenderer
    from iron_amd.rendering_func import make_render_fn
    g = golden("g14_train_S1_c32.npz")
    nets = {k: v.cuda() for k, v in scenes.build_networks("S1").items()}
    cam = Camera(int(g["W"]), int(g["H"]), t(g["K"]).cuda(), t(g["W2C"]).cuda())
    res = render_camera(cam, nets["sdf_network"], RayTracer(), nets, make_render_fn(GGXColocatedRenderer(use_cuda=True)),
                        fill_holes=False, handle_edges=False, is_training=True)
    assert np.array_equal(res["convergent_mask"].cpu().numpy(), g["convergent_mask"])
    assert res["color"].requires_grad
    dc = np.abs(res["color"].detach().cpu().numpy() - g["color"])
    dn = np.abs(res["normal"].detach().cpu().numpy() - g["normal"])
    print("G14 forward: colour max|d| %.2e p99 %.2e rel-L2 %.2e; normal max|d| %.2e" % (dc.max(), np.percentile(dc, 99), _rel(
        res["color"].detach().cpu().numpy(), g["color"]), dn.max()))
    # same bar as the inference path (grazing pixels amplify the split-fp16 core's 1e-6 in the normal)
    assert _rel(res["color"].detach().cpu().numpy(), g["color"]) <= 1e-4 and np.percentile(dc, 99) <= 5e-5 and dn.max() <= 5e-4
    wt = t(g["loss_weights"]).cuda()
    loss = (res["color"] * wt).sum() + 0.1 * (res["normal"] * wt).sum()
    assert abs(loss.item() - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    loss.backward()
    # Against the reference's fp64 run of the same setting (g14_floor_fp64.npz, make_golden_train.py --floor14), per tensor within
    # max(5e-4 | 2e-3, 1.5 x the reference's own fp32-vs-fp64 discrepancy).  The floor matters for two networks: d loss / d roughness
    # at a highlight pixel moves by ~1e-2 per 5e-7 of normal direction (GGX: (1 - c^2) / alpha^2 with alpha ~ 1e-2), so the
    # reference's fp32 run is itself 1.5e-3 ... 4.3e-3 off its fp64 run on the roughness net's tensors and 1.2e-4 on the specular
    # albedo net's; everywhere else the floor is ~1e-7 and the flat bound applies.
    f = golden("g14_floor_fp64.npz")
    floor = lambda key: (float(f["floor_n:" + key]), float(f["floor_s:" + key]))
    n, w, bad = _grads_vs_fp64_reference(nets, f, "gnorm:", "gsample:", floor, base_n=5e-4, base_s=2e-3)
    # and for the record against the fp32 run (not asserted beyond a loose 2 x floor + 5e-4: two fp32-accurate normals differ)
    worst32 = 0.0
    for name in NETS:
        for pname, p in nets[name].named_parameters():
            key = "%s/%s" % (name, pname)
            gr = p.grad.reshape(-1).double().cpu().numpy()
            en32 = abs(np.linalg.norm(gr) - float(g["gnorm:" + key])) / max(float(g["gnorm:" + key]), 1e-12)
            worst32 = max(worst32, en32)
            assert en32 <= 5e-4 + 2.5 * floor(key)[0], (key, en32, floor(key)[0])
    print("G14: %d parameter tensors vs ref64 worst |norm| %.2e entries %.2e (the reference's fp32 vs fp64: %.2e / %.2e); vs ref32 worst |norm| %.2e"
          % (n, w["n"], w["s"], w["fn"], w["fs"], worst32))
    assert not bad, bad
    assert n == golden_meta()["n_param_tensors_train_golden"]
    assert nets["point_light_network"].light.grad is not None


@pytest.mark.parametrize("seed,sigma", [(2, 0.008), (3, 0.014)])
def test_training_render_random_scenes_vs_oracle_autograd(seed, sigma):
    """Beyond the golden scenes: freshly seeded networks, 24x24 crop, every output of render_fn in the loss (colour, the two
    lobes, albedos, roughness, normal); product gradients vs torch.autograd over the pinned oracle, light included."""
    from iron_amd import scenes
    from iron_amd.raytracer import Camera, RayTracer, render_camera
    from iron_amd.renderer_ggx import GGXColocatedRenderer
    from iron_amd.rendering_func import make_render_fn
    from oracle import iron_ref as R
    from oracle import train_ref as T
    torch.manual_seed(seed)
    cpu_nets = scenes.build_networks("S0", seed=seed)
    with torch.no_grad():
        v = cpu_nets["sdf_network"].lin0.weight_v
        v[:, 3:] += sigma * torch.randn(v[:, 3:].shape, generator=torch.Generator().manual_seed(100 + seed))
    mt, md = tables()
    sd = {k: T.leaf_state(cpu_sd(cpu_nets[k])) for k in NETS}
    light = torch.tensor(float(cpu_nets["point_light_network"].light), requires_grad=True)
    sc = R.Scene(sd["sdf_network"], R.SDFSpec(), {k: (sd[k], R.GGX_SPECS[k]) for k in R.GGX_SPECS}, light, mt, md)
    K, W2C = scenes.fixture_camera_matrices(512, 512)
    cam_o = R.CameraSpec(512, 512, K, W2C).crop(24, 24, (250, 236))
    keys = ("color", "diffuse_color", "specular_color", "diffuse_albedo", "specular_albedo", "specular_roughness", "normal")
    gen = torch.Generator().manual_seed(seed)
    wts = {k: torch.randn(24, 24, generator=gen) if k == "specular_roughness" else torch.randn(24, 24, 3, generator=gen) for k in keys}
    ref = T.render_camera_train(sc, cam_o)
    sum((ref[k] * wts[k]).sum() for k in keys).backward()
    nets = {k: m.cuda() for k, m in cpu_nets.items()}
    cam = Camera(512, 512, K.cuda(), W2C.cuda()).crop_region(24, 24, ul_corner=(250, 236))[0]
    res = render_camera(cam, nets["sdf_network"], RayTracer(), nets, make_render_fn(GGXColocatedRenderer(use_cuda=True)),
                        fill_holes=False, handle_edges=False, is_training=True)
    assert np.array_equal(res["convergent_mask"].cpu().numpy(), ref["convergent_mask"].numpy())
    sum((res[k] * wts[k].cuda()).sum() for k in keys).backward()
    worst = 0.0
    for name in NETS:
        for pname, p in nets[name].named_parameters():
            r = sd[name][pname].grad
            if r is None or float(r.abs().max()) < 1e-9:
                continue
            e = _rel(p.grad.cpu().numpy(), r.numpy())
            worst = max(worst, e)
            # every lobe and map is in the loss; the specular lobe's derivative in the roughness (~1/alpha^3) amplifies the forward's
            # 1e-6 at the few grazing pixels of a 24x24 crop (operator-level agreement is 1e-6, see the tests above)
            assert e <= 2e-2, (name, pname, e)
    lg = float(nets["point_light_network"].light.grad)
    assert abs(lg - float(light.grad)) <= 2e-4 * abs(float(light.grad)), (lg, float(light.grad))
    print("seed %d: hits %d, worst parameter-gradient rel-L2 %.2e, d/dlight %.6g vs %.6g" % (seed, int(ref["convergent_mask"].sum()), worst, lg,
                                                                                       float(light.grad)))


def test_a_few_adam_steps_reduce_the_image_loss():
    """The operators are used the way render_surface.py uses them: parameters change every step (the packed device copies
    are rebuilt from the new values), and the image loss against a target rendered from other parameters goes down."""
    from iron_amd import scenes
    from iron_amd.raytracer import Camera, RayTracer, render_camera
    from iron_amd.renderer_ggx import GGXColocatedRenderer
    from iron_amd.rendering_func import make_render_fn
    fn = make_render_fn(GGXColocatedRenderer(use_cuda=True))
    K, W2C = scenes.fixture_camera_matrices(512, 512)
    cam = Camera(512, 512, K.cuda(), W2C.cuda()).crop_region(48, 48, ul_corner=(232, 232))[0]
    target_nets = {k: m.cuda() for k, m in scenes.build_networks("S0", seed=5).items()}
    with torch.no_grad():
        target = render_camera(cam, target_nets["sdf_network"], RayTracer(), target_nets, fn, handle_edges=False)["color"]
    nets = {k: m.cuda() for k, m in scenes.build_networks("S0", seed=0).items()}
    mats = [p for k in NETS[1:] for p in nets[k].parameters()] + list(nets["point_light_network"].parameters())
    opt = torch.optim.Adam([{"params": mats, "lr": 2e-3}, {"params": list(nets["sdf_network"].parameters()), "lr": 1e-5}])
    losses = []
    for _ in range(8):
        res = render_camera(cam, nets["sdf_network"], RayTracer(), nets, fn, handle_edges=False, is_training=True)
        m = res["convergent_mask"]
        loss = (res["color"][m] - target[m]).abs().mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    print("image loss over 8 Adam steps:", " ".join("%.4f" % v for v in losses))
    assert losses[-1] < 0.85 * losses[0] and all(np.isfinite(losses))


def test_composite_backward_vs_autograd():
    """CompositeRenderer (renderer_ggx.py:781-858, point light): all ten inputs, upstream on every returned key (the reference
    returns ONE tensor under "rgb" and "diffuse_rgb"); clamp edges of every map included."""
    from iron_amd.renderer_ggx import CompositeRenderer
    from oracle import iron_ref as R
    mt, md = tables()
    gen = torch.Generator().manual_seed(23)
    n = 3001
    nrm = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1)
    vd = torch.nn.functional.normalize(nrm + 0.9 * torch.randn(n, 3, generator=gen), dim=-1)
    ins = {"light": torch.tensor(27.0), "distance": torch.rand(n, 1, generator=gen) * 2 + 0.5, "normal": nrm, "viewdir": vd,
           "diffuse_albedo": torch.rand(n, 3, generator=gen), "specular_albedo": torch.rand(n, 3, generator=gen) * 0.5,
           "specular_roughness": torch.rand(n, 1, generator=gen) * 0.6 + 0.01, "metallic_eta": torch.rand(n, 1, generator=gen) * 5.5,
           "metallic_k": torch.rand(n, 1, generator=gen) * 11.0, "dielectric_eta": torch.rand(n, 1, generator=gen) * 1.2 + 0.9}
    ins["specular_roughness"][:20] = 5e-6
    ins["diffuse_albedo"][:30, 0] = 1e-6
    ups = {k: torch.randn(n, 3, generator=gen) for k in ("rgb", "diffuse_rgb", "specular_rgb", "metallic_rgb", "dielectric_rgb")}

    def run(dev, fn):
        v = {k: x.clone().to(dev).requires_grad_(True) for k, x in ins.items()}
        prm = {k: v[k] for k in ("diffuse_albedo", "specular_albedo", "specular_roughness", "metallic_eta", "metallic_k", "dielectric_eta")}
        prm["metallic"] = prm["dielectric"] = torch.ones(n, 1, device=dev)
        out = fn(v, prm)
        sum((out[k] * u.to(dev)).sum() for k, u in ups.items()).backward()
        return {k: x.grad.detach().cpu().numpy() for k, x in v.items()}, {k: out[k].detach().cpu().numpy() for k in ups}

    ref, ref_out = run("cpu", lambda v, prm: R.composite_forward(v["light"], v["distance"], v["normal"], v["viewdir"], prm, mt, md))
    rend = CompositeRenderer(use_cuda=True)
    got, got_out = run("cuda", lambda v, prm: rend(v["light"], v["distance"], v["normal"], v["viewdir"], params=prm))
    for k in ups:
        assert _rel(got_out[k], ref_out[k]) <= 2e-5, k
    for k in ins:
        r = _rel(got[k], ref[k])
        print("composite d/d%s rel-L2 %.2e" % (k, r))
        # the dielectric Fresnel term is a difference of nearly equal quantities as eta -> 1 (the map is sampled down to the clamp)
        assert r <= (1e-3 if k == "dielectric_eta" else 2e-5), (k, r)
        dead = ref[k] == 0
        assert float(np.abs(got[k][dead]).max() if dead.any() else 0.0) == 0.0, k  # clamped entries: exactly no gradient
    # the env-light branch (use_env_light=True): intensity = clamp(env_light, 1e-6, 20), returned under "env_light" as well
    ins2 = {k: v for k, v in ins.items() if k not in ("light", "distance")}
    ins2["env_light"] = torch.rand(n, 1, generator=gen) * 24.0 - 1.0
    ups2 = dict(ups, env_light=torch.randn(n, 1, generator=gen))

    def run_env(dev, fn):
        v = {k: x.clone().to(dev).requires_grad_(True) for k, x in ins2.items()}
        prm = {k: v[k] for k in ("diffuse_albedo", "specular_albedo", "specular_roughness", "metallic_eta", "metallic_k", "dielectric_eta",
                                 "env_light")}
        prm["metallic"] = prm["dielectric"] = torch.ones(n, 1, device=dev)
        out = fn(v, prm)
        sum((out[k] * u.to(dev)).sum() for k, u in ups2.items()).backward()
        return {k: x.grad.detach().cpu().numpy() for k, x in v.items()}

    dist0 = ins["distance"]
    ref2 = run_env("cpu", lambda v, prm: R.composite_forward(5.0, dist0, v["normal"], v["viewdir"], prm, mt, md, use_env_light=True))
    got2 = run_env("cuda", lambda v, prm: rend(5.0, dist0.cuda(), v["normal"], v["viewdir"], params=prm, use_env_light=True))
    for k in ins2:
        r = _rel(got2[k], ref2[k])
        print("composite(env) d/d%s rel-L2 %.2e" % (k, r))
        assert r <= (1e-3 if k == "dielectric_eta" else 2e-5), (k, r)


def test_composite_training_render_vs_oracle_autograd():
    """render_camera(is_training=True) with render_fn_comp (render_surface.py:159-234; the network set model_bed.py trains):
    SDF net + eight material nets + light, gradients vs torch.autograd over the pinned oracle on a 24x24 crop of scene S2."""
    from iron_amd import scenes
    from iron_amd.raytracer import Camera, RayTracer, render_camera
    from iron_amd.renderer_ggx import CompositeRenderer
    from iron_amd.rendering_func import make_render_fn_comp
    from oracle import iron_ref as R
    from oracle import train_ref as T
    cpu_nets = scenes.build_comp_networks()
    mt, md = tables()
    names = ["sdf_network"] + list(R.COMP_SPECS)
    sd = {k: T.leaf_state(cpu_sd(cpu_nets[k])) for k in names}
    light = torch.tensor(float(cpu_nets["point_light_network"].light), requires_grad=True)
    sc = R.Scene(sd["sdf_network"], R.SDFSpec(), {k: (sd[k], R.COMP_SPECS[k]) for k in R.COMP_SPECS}, light, mt, md, renderer="comp")
    K, W2C = scenes.fixture_camera_matrices(512, 512)
    gen = torch.Generator().manual_seed(29)
    wt = torch.rand(24, 24, 3, generator=gen) - 0.3
    ref = T.render_camera_train(sc, R.CameraSpec(512, 512, K, W2C).crop(24, 24, (244, 244)))
    ((ref["color"] * wt).sum() + 0.1 * (ref["normal"] * wt).sum() + 0.05 * (ref["specular_color"] * wt).sum()).backward()
    nets = {k: m.cuda() for k, m in cpu_nets.items()}
    cam = Camera(512, 512, K.cuda(), W2C.cuda()).crop_region(24, 24, ul_corner=(244, 244))[0]
    res = render_camera(cam, nets["sdf_network"], RayTracer(), nets, make_render_fn_comp(CompositeRenderer(use_cuda=True)),
                        fill_holes=False, handle_edges=False, is_training=True)
    assert np.array_equal(res["convergent_mask"].cpu().numpy(), ref["convergent_mask"].numpy())
    assert _rel(res["color"].detach().cpu().numpy(), ref["color"].detach().numpy()) <= 1e-4
    w = wt.cuda()
    ((res["color"] * w).sum() + 0.1 * (res["normal"] * w).sum() + 0.05 * (res["specular_color"] * w).sum()).backward()
    worst = {}
    for name in names:
        for pname, p in nets[name].named_parameters():
            r = sd[name][pname].grad
            if r is None or float(r.abs().max()) < 1e-9:
                assert p.grad is None or float(p.grad.abs().max()) <= 1e-6, (name, pname)
                continue
            worst[name] = max(worst.get(name, 0.0), _rel(p.grad.cpu().numpy(), r.numpy()))
    lg = float(nets["point_light_network"].light.grad)
    print("composite training render: worst rel-L2 per net", {k: "%.1e" % v for k, v in worst.items()}, "d/dlight %.6g vs %.6g" % (lg, float(light.grad)))
    assert max(worst.values()) <= 2e-3
    assert abs(lg - float(light.grad)) <= 2e-4 * abs(float(light.grad))


@pytest.mark.parametrize("with_bg,with_rgb", [(True, False), (False, True), (True, True)])
def test_neus_composite_backward_vs_autograd(with_bg, with_rgb):
    """The compositing of render_core (renderer.py:279-344) and its reverse scan vs torch.autograd over oracle.neus_ref.composite:
    gradients w.r.t. sdf, normals, sample colours, 1/s and the background density / colour, from colour, weight_sum (the mask
    loss) and the eikonal statistic."""
    from iron_amd.autograd import NeusCompositeFn
    from oracle import neus_ref as N
    gen = torch.Generator().manual_seed(41)
    n, m, mo = 193, 128, 160
    z = torch.sort(torch.rand(n, mo, generator=gen) * 2.2 + 1.2, dim=-1)[0]
    o = torch.tensor([[0.0, 0.0, -2.3]]).expand(n, 3)
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen) * torch.tensor([0.25, 0.25, 0.0]) + torch.tensor([0.0, 0.0, 1.0]), dim=-1)
    zi = z[:, :m]
    dists = torch.cat([zi[:, 1:] - zi[:, :-1], torch.full((n, 1), 2.0 / 64)], dim=-1)
    pts = (o[:, None, :] + d[:, None, :] * (zi + dists * 0.5)[..., None]).reshape(-1, 3)
    dirs = d[:, None, :].expand(n, m, 3).reshape(-1, 3)
    bgd_ = torch.cat([z[:, 1:] - z[:, :-1], torch.full((n, 1), 2.0 / 64)], dim=-1)
    ins = {"sdf": (pts.norm(dim=-1, keepdim=True) - 0.6 + 0.01 * torch.randn(n * m, 1, generator=gen)),
           "grad": torch.nn.functional.normalize(pts, dim=-1) * (1 + 0.1 * torch.randn(n * m, 1, generator=gen)) + 0.05 * torch.randn(n * m, 3, generator=gen),
           "color": torch.rand(n * m, 3, generator=gen), "inv_s": torch.tensor(37.0)}
    if with_bg:
        ins["bg_density"] = torch.randn(n * mo, 1, generator=gen)
        ins["bg_color"] = torch.randn(n * mo, 3, generator=gen) * 0.5
    bgrgb = torch.tensor([[0.2, 0.5, 0.9]]) if with_rgb else None
    up_c, up_w, up_g = torch.randn(n, 3, generator=gen), torch.randn(n, 1, generator=gen), 0.7

    v = {k: x.clone().requires_grad_(True) for k, x in ins.items()}
    bga = bgc = None
    if with_bg:
        bga = (1.0 - torch.exp(-torch.nn.functional.softplus(v["bg_density"].reshape(n, mo)) * bgd_)).reshape(n, mo)
        bgc = v["bg_color"].reshape(n, mo, 3)
    ref = N.composite(v["sdf"], v["grad"], v["color"].reshape(n, m, 3), dists, pts, dirs, v["inv_s"], bga, bgc, bgrgb, 0.3)
    ((ref["color"] * up_c).sum() + (ref["weights"].sum(dim=-1, keepdim=True) * up_w).sum() + up_g * ref["gradient_error"]).backward()

    g = {k: x.clone().cuda().requires_grad_(True) for k, x in ins.items()}
    col, w, wsum, gerr, cdf, inside, wmax = NeusCompositeFn.apply(g["sdf"], g["grad"], g["color"], g["inv_s"], g.get("bg_density"), g.get("bg_color"),
                                                                  dists.cuda(), pts.cuda(), dirs.cuda(), bgd_.cuda() if with_bg else None,
                                                                  bgrgb.cuda() if with_rgb else None, 0.3)
    assert _rel(col.detach().cpu().numpy(), ref["color"].detach().numpy()) <= 1e-5
    assert abs(float(gerr) - float(ref["gradient_error"])) <= 1e-5 * abs(float(ref["gradient_error"]))
    assert not cdf.requires_grad and not wmax.requires_grad
    ((col * up_c.cuda()).sum() + (wsum * up_w.cuda()).sum() + up_g * gerr).backward()
    for k in ins:
        r = _rel(g[k].grad.cpu().numpy(), v[k].grad.numpy())
        print("neus composite (bg=%s rgb=%s) d/d%s rel-L2 %.2e" % (with_bg, with_rgb, k, r))
        assert r <= 2e-4, (k, r)


def test_nerf_backward_vs_autograd():
    """NeRF background field (fields.py:243-327): parameter gradients from alpha and rgb vs torch.autograd over the oracle, at a
    size that takes the split-K path, and from one head only."""
    from iron_amd.fields import NeRF
    from oracle import neus_ref as N
    from oracle import train_ref as T
    torch.manual_seed(5)
    mod = NeRF(D=8, d_in=4, d_in_view=3, W=256, multires=10, multires_view=4, output_ch=4, skips=[4], use_viewdirs=True)
    net_sd = cpu_sd(mod)
    net = mod.cuda()
    gen = torch.Generator().manual_seed(6)
    for n, heads in ((4603, "ar"), (257, "a"), (257, "r")):
        pts = torch.rand(n, 4, generator=gen) * 2 - 1
        views = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1)
        ua, ur = torch.randn(n, 1, generator=gen), torch.randn(n, 3, generator=gen)
        sd = T.leaf_state(net_sd)
        alpha, rgb = N.nerf_forward(sd, N.NerfSpec(), pts, views)
        ((alpha * ua).sum() * ("a" in heads) + (rgb * ur).sum() * ("r" in heads)).backward()
        a2, r2 = net(pts.cuda(), views.cuda())
        assert a2.requires_grad and _rel(r2.detach().cpu().numpy(), rgb.detach().numpy()) <= 1e-5
        ((a2 * ua.cuda()).sum() * ("a" in heads) + (r2 * ur.cuda()).sum() * ("r" in heads)).backward()
        w = _compare_param_grads(net, sd, 2e-4, "nerf n=%d %s" % (n, heads))
        print("nerf backward n=%d heads=%s worst rel-L2 %.2e (%s)" % (n, heads, w, getattr(_compare_param_grads, "last", "")))


def _stage1_nets():
    from iron_amd.fields import NeRF, RenderingNetwork, SDFNetwork, SingleVarianceNetwork
    torch.manual_seed(0)
    return {
        "sdf_network": SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5, scale=1.0,
                                  geometric_init=True, weight_norm=True),
        "color_network": RenderingNetwork(d_feature=256, mode="idr", d_in=9, d_out=3, d_hidden=256, n_layers=8, skip_in=[4],
                                          weight_norm=True, multires=10, multires_view=4, squeeze_out=True),
        "nerf": NeRF(D=8, d_in=4, d_in_view=3, W=256, multires=10, multires_view=4, output_ch=4, skips=[4], use_viewdirs=True),
        "deviation_network": SingleVarianceNetwork(0.3),
    }


def test_g16_neus_training_render_matches_reference_gradients():
    """NeuSRenderer.render under autograd + loss.backward() (render_volume.py:160-200: colour, eikonal statistic, weight_sum) vs
    the REAL reference: gradient norm and sampled entries of all 79 parameter tensors of the four stage-1 networks (G16)."""
    from iron_amd.renderer import NeuSRenderer
    g, g13 = golden("g16_neus_train.npz"), golden("g13_neus.npz")
    nets = {k: v.cuda() for k, v in _stage1_nets().items()}
    r = NeuSRenderer(nets["nerf"], nets["sdf_network"], nets["deviation_network"], nets["color_network"], n_samples=64, n_importance=64,
                     n_outside=32, up_sample_steps=4, perturb=0.0)
    out = r.render(t(g13["rays_o"]).cuda(), t(g13["rays_d"]).cuda(), t(g13["near"]).cuda(), t(g13["far"]).cuda(), perturb_overwrite=0,
                   background_rgb=None, cos_anneal_ratio=0.3)
    assert out["color_fine"].requires_grad and out["gradient_error"].requires_grad and out["s_val"].requires_grad
    assert np.abs(out["color_fine"].detach().cpu().numpy() - g["color_fine"]).max() <= 1e-4
    loss = (out["color_fine"] * t(g["loss_wc"]).cuda()).sum() + 0.1 * out["gradient_error"] + (out["weight_sum"] * t(g["loss_ww"]).cuda()).sum()
    assert abs(loss.item() - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    loss.backward()
    n, worst_n, worst_s, bad = 0, 0.0, 0.0, []
    for name in sorted(nets):
        for pname, p in nets[name].named_parameters():
            key = "%s/%s" % (name, pname)
            assert p.grad is not None, key
            gr = p.grad.reshape(-1).double().cpu().numpy()
            ref_n = float(g["gnorm:" + key])
            en = abs(np.linalg.norm(gr) - ref_n) / max(ref_n, 1e-12)
            ref_s = g["gsample:" + key]
            idx = np.concatenate([np.arange(min(16, gr.size)), np.linspace(0, gr.size - 1, 32).astype(np.int64)])
            es = float(np.abs(gr[idx] - ref_s).max() / max(np.abs(ref_s).max(), 1e-12))
            worst_n, worst_s = max(worst_n, en), max(worst_s, es)
            if en > 2e-3 or es > 1e-2:  # measured 1.8e-4 / 3.8e-3
                bad.append((key, en, es))
            n += 1
    print("G16: %d parameter tensors, worst |norm| error %.2e, worst sampled-entry error %.2e (of max)" % (n, worst_n, worst_s))
    assert not bad, bad
    assert n == golden_meta()["n_param_tensors_neus_train_golden"]


def test_neus_render_stratified_jitter_and_inference_mode():
    """perturb > 0 (renderer.py:369-378) runs and stays a valid placement; under no_grad nothing is attached."""
    from iron_amd.renderer import NeuSRenderer
    g13 = golden("g13_neus.npz")
    nets = {k: v.cuda() for k, v in _stage1_nets().items()}
    r = NeuSRenderer(nets["nerf"], nets["sdf_network"], nets["deviation_network"], nets["color_network"], n_samples=64, n_importance=64,
                     n_outside=32, up_sample_steps=4, perturb=1.0)
    args = [t(g13[k]).cuda() for k in ("rays_o", "rays_d", "near", "far")]
    torch.manual_seed(3)
    a = r.render(*args, cos_anneal_ratio=0.5)
    b = r.render(*args, cos_anneal_ratio=0.5)
    assert a["color_fine"].requires_grad and a["weights"].shape == (96, 160)
    assert float((a["color_fine"] - b["color_fine"]).abs().max()) > 0.0  # two different jitters
    assert float(a["weight_sum"].max()) <= 1.0 + 1e-4 and bool(torch.isfinite(a["color_fine"]).all())
    with torch.no_grad():
        c = r.render(*args, perturb_overwrite=0, cos_anneal_ratio=0.5)
    assert not c["color_fine"].requires_grad and not c["gradient_error"].requires_grad


@pytest.mark.parametrize("kind", ["smooth_dielectric", "thin_dielectric", "smooth_conductor", "rough_conductor"])
def test_simple_head_backward_vs_autograd(kind):
    """SmoothDielectric / ThinDielectric / SmoothConductorCoLoc / RoughConductorCoLoc (renderer_ggx.py:149-395) under autograd."""
    from iron_amd import renderer_ggx as G
    from oracle import iron_ref as R
    cls = {"smooth_dielectric": G.SmoothDielectricRenderer, "thin_dielectric": G.ThinDielectricRenderer,
           "smooth_conductor": G.SmoothConductorCoLocRenderer, "rough_conductor": G.RoughConductorCoLocRenderer}[kind]
    head = cls(use_cuda=True) if kind.endswith("dielectric") else cls(ior_path="./resource/ior", use_cuda=True)
    ofn = getattr(R, kind)
    gen = torch.Generator().manual_seed(51)
    n = 2051
    nrm = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1)
    vd = torch.nn.functional.normalize(nrm + 0.9 * torch.randn(n, 3, generator=gen), dim=-1)
    ins = {"light": torch.tensor(21.0), "distance": torch.rand(n, 1, generator=gen) * 2 + 0.5, "normal": nrm, "viewdir": vd,
           "kd": torch.rand(n, 3, generator=gen), "ks": torch.rand(n, 3, generator=gen) * 0.5, "alpha": torch.rand(n, 1, generator=gen) * 0.6 + 0.01}
    ups = [torch.randn(n, 3, generator=gen) for _ in range(3)]

    def run(dev, fn):
        v = {k: x.clone().to(dev).requires_grad_(True) for k, x in ins.items()}
        out = fn(v)
        sum((out[k] * u.to(dev)).sum() for k, u in zip(("diffuse_rgb", "specular_rgb", "rgb"), ups)).backward()
        return {k: (x.grad.detach().cpu().numpy() if x.grad is not None else np.zeros(tuple(x.shape), np.float32)) for k, x in v.items()}

    ref = run("cpu", lambda v: ofn(v["light"], v["distance"], v["normal"], v["viewdir"], v["kd"], v["ks"], v["alpha"]))
    got = run("cuda", lambda v: head(v["light"], v["distance"], v["normal"], v["viewdir"], v["kd"], v["ks"], v["alpha"]))
    for k in ins:
        if float(np.abs(ref[k]).max()) == 0.0:
            assert float(np.abs(got[k]).max()) == 0.0, k
            continue
        r = _rel(got[k], ref[k])
        print("%s d/d%s rel-L2 %.2e" % (kind, k, r))
        assert r <= 1e-4, (kind, k, r)
