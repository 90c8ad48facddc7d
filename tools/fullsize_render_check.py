"""One-off full-size forward check against the oracle (minutes of host time: not in the suite): render_camera on SCENE at
SIZE x SIZE, mask flips and colour / normal agreement with oracle.iron_ref.render_camera on the GPU box's host cores.
    python tools/fullsize_render_check.py [SCENE=S1] [SIZE=800] [edges]
With `edges`: fill_holes=True, handle_edges=True (closing, sobel, silhouette walk, side-ray blend) on both sides."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import oracle_scene  # noqa: E402
from iron_amd import scenes  # noqa: E402
from iron_amd.raytracer import Camera, RayTracer, render_camera  # noqa: E402
from iron_amd.renderer_ggx import GGXColocatedRenderer  # noqa: E402
from iron_amd.rendering_func import make_render_fn  # noqa: E402
from oracle import iron_ref as R  # noqa: E402

torch.set_grad_enabled(False)
scene = sys.argv[1] if len(sys.argv) > 1 else "S1"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 800
edges = len(sys.argv) > 3 and sys.argv[3] == "edges"
torch.set_num_threads(min(16, os.cpu_count() or 8))
cpu_nets = scenes.build_networks(scene)
K, W2C = scenes.fixture_camera_matrices(size, size)
nets = {k: m.cuda() for k, m in scenes.build_networks(scene).items()}
res = render_camera(Camera(size, size, K.cuda(), W2C.cuda()), nets["sdf_network"], RayTracer(), nets, make_render_fn(GGXColocatedRenderer(use_cuda=True)),
                    fill_holes=edges, handle_edges=edges)
torch.cuda.synchronize()
print("product done; oracle: %s %dx%d on %d threads ..." % (scene, size, size, torch.get_num_threads()), flush=True)
t0 = time.time()
sc = oracle_scene(cpu_nets)
cam = R.CameraSpec(size, size, K, W2C)
if edges:
    tr = R.raytrace_camera_full(sc, cam, max_num_rays=50000, fill_holes=True, detect_edges=True)
    print("oracle: traced + edge walk after %.1f s" % (time.time() - t0), flush=True)
    R.render_normal_and_color(sc, tr)
    if tr["edge_mask"].sum() > 0:
        R.render_edge_pixels(sc, tr, cam)
    print("oracle: shaded + edge pixels after %.1f s" % (time.time() - t0), flush=True)
    em, rem = res["edge_mask"].cpu().numpy(), tr["edge_mask"].numpy()
    common = em & rem
    ec = np.abs(res["color"].cpu().numpy() - tr["color"].numpy())[common].max(axis=-1)
    print("edge pixels: product %d, oracle %d, symmetric difference %d; blended colour |d| median %.2e p90 %.2e" % (
        int(em.sum()), int(rem.sum()), int((em != rem).sum()), float(np.median(ec)), float(np.percentile(ec, 90))), flush=True)
else:
    tr = R.raytrace_camera(sc, cam, max_num_rays=50000)
    print("oracle: traced after %.1f s" % (time.time() - t0), flush=True)
    R.render_normal_and_color(sc, tr)
    print("oracle: shaded after %.1f s" % (time.time() - t0), flush=True)
conv, rconv = res["convergent_mask"].cpu().numpy(), tr["convergent_mask"].numpy()
both = conv & rconv
col, rcol = res["color"].cpu().numpy()[both].astype(np.float64), tr["color"].numpy()[both].astype(np.float64)
dn = np.abs(res["normal"].cpu().numpy() - tr["normal"].numpy())[both]
dd = np.abs(res["distance"].cpu().numpy() - tr["distance"].numpy())[both]
print("%s %dx%d: hits %d (oracle %d), mask flips %d, colour rel-L2 %.2e (p99 |d| %.2e, max %.2e), normal max|d| %.2e, distance max|d| %.2e" % (
    scene, size, size, int(conv.sum()), int(rconv.sum()), int((conv != rconv).sum()), np.linalg.norm(col - rcol) / np.linalg.norm(rcol),
    np.percentile(np.abs(col - rcol), 99), np.abs(col - rcol).max(), dn.max(), dd.max()))
