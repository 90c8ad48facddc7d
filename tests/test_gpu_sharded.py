"""GPU: the tile-sharded multi-rank path equals the single-process path BIT FOR BIT (rays are independent and
the chunk-global bisection count is exchanged), rehearsed with 2 and 3 ranks on one card over gloo."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 3])
def test_multi_rank_equals_single_process(world):
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(29500 + world), os.path.join(ROOT, "tests", "run_sharded_check.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "SHARDED_CHECK OK" in r.stdout, r.stdout[-3000:]
    assert "SHARDED_EDGES_CHECK OK" in r.stdout, r.stdout[-3000:]


def test_world1_sharded_renderer_equals_render_camera():
    from iron_amd import scenes
    from iron_amd.raytracer import Camera, RayTracer, render_camera
    from iron_amd.renderer_ggx import GGXColocatedRenderer
    from iron_amd.rendering_func import make_render_fn
    from iron_amd.sharding import ShardedRenderer
    nets = {k: v.cuda() for k, v in scenes.build_networks("S0").items()}
    fn = make_render_fn(GGXColocatedRenderer(use_cuda=True))
    K, W2C = scenes.fixture_camera_matrices(100, 100)
    cam = Camera(100, 100, K.cuda(), W2C.cuda())
    out = ShardedRenderer(nets["sdf_network"], nets, RayTracer(), fn, tile=32, chunk=50000).render([cam])
    ref = render_camera(cam, nets["sdf_network"], RayTracer(), nets, fn, fill_holes=False, handle_edges=False)
    for k in ("convergent_mask", "distance", "depth", "color", "normal", "specular_roughness", "points"):
        assert np.array_equal(out[k][0].cpu().numpy(), ref[k].cpu().numpy()), k
