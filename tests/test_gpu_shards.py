"""GPU: BASELINE config C4's structure on one card -- a view's rays tile-sharded over 8 ranks (iron_amd.sharding), every
shard run in turn through the SAME phase methods ShardedRenderer.render() uses, the exchanges done in memory.  The
assembled image must be bit-equal to render_camera's, at the C4 per-view size (1600x1600) and at the headline 800x800,
with and without the whole-image passes (hole filling on every rank, silhouette edges on rank 0), and for the full C4 call
(8 views x 1600x1600 in one render).  Parity and properties only: timings live in bench.py / tools/shard_scaling.py."""
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


def _check_equal(out, ref, v=0):
    for k, _ in RECORD:
        a, b = out[k][v], ref[k]
        assert a.shape == b.shape, (k, a.shape, b.shape)
        assert torch.equal(a, b), (k, float((a.float() - b.float()).abs().max()))


def _invariants(out, v, sdf, res):
    """The full-size invariants (tests/test_gpu_fullsize.py) on view v of an assembled sharded result."""
    conv = out["convergent_mask"][v]
    assert conv.shape == (res, res) and conv.dtype == torch.bool
    s = out["sdf"][v][conv].abs()
    assert float(s.max()) <= 2e-4
    assert torch.equal(sdf.sdf(out["points"][v][conv])[:, 0], out["sdf"][v][conv])
    assert torch.all(out["depth"][v][~conv] == 0)
    for k in ("color", "diffuse_color", "specular_color", "diffuse_albedo", "specular_albedo", "normal"):
        assert torch.all(out[k][v][~conv] == 0) and torch.isfinite(out[k][v]).all(), k
    assert float((out["normal"][v][conv].norm(dim=-1) - 1).abs().max()) <= 1e-5
    col = out["color"][v][conv]
    np.testing.assert_allclose(col.cpu().numpy(), (out["diffuse_color"][v] + out["specular_color"][v])[conv].cpu().numpy(), rtol=1e-6, atol=1e-8)
    return int(conv.sum())


@pytest.mark.parametrize("res,yaw", [(1600, 45.0), (800, 0.0)])
def test_eight_tile_shards_equal_render_camera(res, yaw):
    """C4's per-view size (1600x1600, an orbit view) and C1's size: 8 tile-shards == render_camera bit for bit; the full-size
    invariants hold on the assembled image.  (Per-shard timings and the strong-scaling prediction are measured by bench.py's
    `predicted_strong_scaling` and tools/shard_scaling.py, not asserted here.)"""
    nets, fn = _scene("S0")
    sdf = nets["sdf_network"]
    K, W2C = scenes.fixture_camera_matrices(res, res, yaw_deg=yaw)
    cam = Camera(res, res, K.cuda(), W2C.cuda())
    ref = render_camera(cam, sdf, RayTracer(), nets, fn, fill_holes=False, handle_edges=False)
    out, ms, _ = render_emulated(8, [cam], sdf, nets, fn, RayTracer)
    _check_equal(out, ref)
    assert _invariants(out, 0, sdf, res) == int(ref["convergent_mask"].sum())
    assert len(ms) == 8 and all(m > 0 for m in ms)


def test_c4_eight_views_1600_over_eight_shards():
    """BASELINE config C4 as specified: ONE call renders 8 views of 1600x1600 (the fixture pose orbited by k x 45 degrees, the
    camera loop of utils/process_routine.py:11-35), every view's rays tile-sharded over 8 ranks -- played through the rank-local
    phase methods of ShardedRenderer on one card.  20.5 M rays per call: every view is held to the full-size invariants, views 0
    and 5 are bit-equal to render_camera, and the view-major record layout keeps the views apart (each view's hit count is its
    own single-view count; opposite views of the symmetric sphere scene do not have to agree, so they are all checked)."""
    nets, fn = _scene("S0")
    sdf = nets["sdf_network"]
    res = 1600
    cams = [Camera(res, res, *(m.cuda() for m in scenes.fixture_camera_matrices(res, res, yaw_deg=45.0 * v))) for v in range(8)]
    out, ms, _ = render_emulated(8, cams, sdf, nets, fn, RayTracer)
    assert out["color"].shape == (8, res, res, 3) and out["convergent_mask"].shape == (8, res, res)
    hits = [_invariants(out, v, sdf, res) for v in range(8)]
    assert min(hits) > 0.3 * res * res, hits
    for v in (0, 5):
        ref = render_camera(cams[v], sdf, RayTracer(), nets, fn, fill_holes=False, handle_edges=False)
        _check_equal(out, ref, v)
        assert hits[v] == int(ref["convergent_mask"].sum())
        del ref
    # every view went through every shard: 8 shards x 8 views, no shard idle
    assert len(ms) == 8 and min(ms) > 0.25 * max(ms), ms


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
