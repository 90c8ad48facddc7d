"""Diagnostic (not collected): the CPU oracle on THIS host against the G15 fixture (recorded from the reference on the build
container's CPU) -- which pixels differ, by how much, and are they edge pixels."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from iron_amd import scenes  # noqa: E402
from oracle import iron_ref as R  # noqa: E402
from oracle import train_ref as T  # noqa: E402
from _util import cpu_sd, golden_meta, tables  # noqa: E402

NETS = ("sdf_network", "diffuse_albedo_network", "specular_albedo_network", "specular_roughness_network")
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "g15_train_edges_S1.npz")))
size = int(g["W"])
mt, md = tables()
nets = scenes.build_networks("S1")
sd = {k: T.leaf_state(cpu_sd(nets[k])) for k in NETS}
sc = R.Scene(sd["sdf_network"], R.SDFSpec(), {k: (sd[k], R.GGX_SPECS[k]) for k in R.GGX_SPECS}, golden_meta()["light"], mt, md)
for threads in (8, 16, 1):
    torch.set_num_threads(threads)
    res = T.render_camera_edges_train(sc, R.CameraSpec(size, size, torch.from_numpy(g["K"]), torch.from_numpy(g["W2C"])),
                                      torch.from_numpy(g["depth_edge_mask_input"]))
    col = res["color"].detach().numpy()
    d = np.abs(col - g["color"]).max(axis=-1)
    idx = np.argsort(-d.reshape(-1))[:5]
    print("threads %2d: edge mask equal %s, conv mask equal %s; colour max|d| %.3e; top pixels %s (edge? %s) |d| %s" % (
        threads, np.array_equal(res["edge_mask"].numpy(), g["edge_mask"]), np.array_equal(res["convergent_mask"].numpy(), g["convergent_mask"]),
        d.max(), idx.tolist(), g["edge_mask"].reshape(-1)[idx].tolist(), d.reshape(-1)[idx].round(6).tolist()), flush=True)
