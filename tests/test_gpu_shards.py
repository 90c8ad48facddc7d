"""GPU: BASELINE config C4's structure on one card -- a view's rays tile-sharded over 8 ranks (iron_amd.sharding), every
shard run in turn through the SAME phase methods ShardedRenderer.render() uses, the exchanges done in memory.  The
assembled image must be bit-equal to render_camera's, at the C4 per-view size (1600x1600) and at the headline 800x800,
with and without the whole-image passes (hole filling on every rank, silhouette edges on rank 0); each shard's device
time is recorded, and T(one frame) / max_r T(shard r) -- the strong-scaling factor load balance allows -- is printed."""
import json
import os

import numpy as np
import pytest
import torch

from iron_amd import scenes
from iron_amd.raytracer import Camera, RayTracer, render_camera
from iron_amd.renderer_ggx import GGXColocatedRenderer
from iron_amd.rendering_func import make_render_fn
from iron_amd.sharding import RECORD, render_emulated

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scene(name="S0"):
    nets = {k: v.cuda() for k, v in scenes.build_networks(name).items()}
    return nets, make_render_fn(GGXColocatedRenderer(use_cuda=True))


def _frame_ms(f, reps=2):
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def _check_equal(out, ref, v=0):
    for k, _ in RECORD:
        a, b = out[k][v], ref[k]
        assert a.shape == b.shape, (k, a.shape, b.shape)
        assert torch.equal(a, b), (k, float((a.float() - b.float()).abs().max()))


def _record(name, payload):
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "shard_scaling_%s.json" % name), "w") as f:
        json.dump(payload, f, indent=1)


@pytest.mark.parametrize("res,yaw", [(1600, 45.0), (800, 0.0)])
def test_eight_tile_shards_equal_render_camera_and_balance(res, yaw):
    """C4 (per-view size 1600x1600, an orbit view) and C1's size: 8 tile-shards == render_camera bit for bit; the
    full-size invariants hold on the assembled image; the slowest shard bounds the 8-GPU step."""
    nets, fn = _scene("S0")
    sdf = nets["sdf_network"]
    K, W2C = scenes.fixture_camera_matrices(res, res, yaw_deg=yaw)
    cam = Camera(res, res, K.cuda(), W2C.cuda())
    ref = render_camera(cam, sdf, RayTracer(), nets, fn, fill_holes=False, handle_edges=False)
    t_frame = _frame_ms(lambda: render_camera(cam, sdf, RayTracer(), nets, fn, fill_holes=False, handle_edges=False))
    render_emulated(8, [cam], sdf, nets, fn, RayTracer)  # warm the per-shard launches
    runs = [render_emulated(8, [cam], sdf, nets, fn, RayTracer) for _ in range(3)]
    out = runs[-1][0]
    ms = [min(r[1][k] for r in runs) for k in range(8)]   # per shard: best of 3 (an allocator stall inflates a single run 3-10x)
    ms_asm = min(r[2] for r in runs)
    _check_equal(out, ref)
    # invariants of the assembled image (the ones test_fullsize_invariants holds the unsharded frame to)
    conv = out["convergent_mask"][0]
    assert conv.shape == (res, res) and conv.dtype == torch.bool
    s = out["sdf"][0][conv].abs()
    assert float(s.max()) <= 2e-4
    assert torch.equal(sdf.sdf(out["points"][0][conv])[:, 0], out["sdf"][0][conv])
    assert torch.all(out["depth"][0][~conv] == 0)
    for k in ("color", "diffuse_color", "specular_color", "diffuse_albedo", "specular_albedo", "normal"):
        assert torch.all(out[k][0][~conv] == 0) and torch.isfinite(out[k][0]).all(), k
    assert float((out["normal"][0][conv].norm(dim=-1) - 1).abs().max()) <= 1e-5
    col = out["color"][0][conv]
    np.testing.assert_allclose(col.cpu().numpy(), (out["diffuse_color"][0] + out["specular_color"][0])[conv].cpu().numpy(), rtol=1e-6, atol=1e-8)
    # load balance: interleaved 32x32 tiles keep the slowest shard close to the mean
    factor = t_frame / max(ms)
    print("%dx%d: frame %.2f ms; shards %s ms (max %.2f, mean %.2f); assemble %.2f ms; predicted 8-GPU strong scaling %.2fx "
          "(kernels only; + all-reduce, gather and assemble on rank 0)" % (res, res, t_frame, [round(x, 2) for x in ms], max(ms),
                                                                          sum(ms) / 8, ms_asm, factor))
    _record("%d" % res, {"res": res, "world": 8, "frame_ms": t_frame, "shard_ms": ms, "assemble_ms": ms_asm,
                         "predicted_strong_scaling_kernels_only": factor, "hits": int(conv.sum())})
    assert max(ms) <= 1.35 * (sum(ms) / 8)
    # 640 k rays over 8 shards leave each persistent kernel ~2.4 waves of workgroups: the per-launch tails (a ray's 17
    # sequential evaluations) do not shrink with the shard, so the factor at 800x800 is well below the 1600x1600 one
    assert factor >= (5.0 if res == 1600 else 3.0)


def test_shards_with_hole_filling_and_edges_equal_render_camera():
    """The whole-image passes in the sharded form (hole filling on every rank from the all-gathered trace records,
    silhouette edge sampling on rank 0 after the gather) reproduce render_camera(fill_holes=True, handle_edges=True)."""
    nets, fn = _scene("S1")
    sdf = nets["sdf_network"]
    res = 400
    K, W2C = scenes.fixture_camera_matrices(res, res, yaw_deg=90.0)
    cam = Camera(res, res, K.cuda(), W2C.cuda())
    for fill, edges in ((True, False), (False, True), (True, True)):
        ref = render_camera(cam, sdf, RayTracer(), nets, fn, fill_holes=fill, handle_edges=edges)
        for world in (3, 8):
            out, _, _ = render_emulated(world, [cam], sdf, nets, fn, RayTracer, fill_holes=fill, handle_edges=edges)
            _check_equal(out, ref)
            if edges:
                assert torch.equal(out["edge_mask"][0], ref["edge_mask"])
                assert torch.equal(out["edge_pixel_idx"][0], ref["edge_pixel_idx"])
                assert torch.equal(out["uv"][0], ref["uv"])
                assert int(ref["edge_mask"].sum()) > 100
    # a multi-view batch (C4 renders 8 views per call): view-major records, chunk tables spaced per view
    cams = [Camera(200, 200, *(m.cuda() for m in scenes.fixture_camera_matrices(200, 200, yaw_deg=45.0 * v))) for v in range(3)]
    out, _, _ = render_emulated(4, cams, sdf, nets, fn, RayTracer, fill_holes=True, handle_edges=False)
    for v, c in enumerate(cams):
        _check_equal(out, render_camera(c, sdf, RayTracer(), nets, fn, fill_holes=True, handle_edges=False), v)
