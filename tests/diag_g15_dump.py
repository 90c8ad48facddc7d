"""Diagnostic (not collected).  `dump`: run the CPU oracle on this host for the G15 setting, loss = term B (edge colour), and
save, for every get_all call (main pixels, edge points, positive / negative side rays), the query points, the sdf values, and
dB/d(sdf), dB/d(gradient), dB/d(feature) norms per point.  `compare`: the same from the HIP path, compared call by call with
a dump made on another host.    python tests/diag_g15_dump.py dump|compare"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from iron_amd import scenes  # noqa: E402

g = dict(np.load(os.path.join(ROOT, "tests", "golden", "g15_train_edges_S1.npz")))
size = int(g["W"])
K, W2C = torch.from_numpy(g["K"]), torch.from_numpy(g["W2C"])
dem = torch.from_numpy(g["depth_edge_mask_input"])
wt = torch.from_numpy(g["loss_weights"])
em = torch.from_numpy(g["edge_mask"]).bool()
PATH = os.path.join(ROOT, "tests", "golden", "_diag_g15_host.npz")
NETS = ("sdf_network", "diffuse_albedo_network", "specular_albedo_network", "specular_roughness_network")
calls = []


def record(x, y, feat, grad):
    for t in (y, feat, grad):
        t.retain_grad()
    calls.append((x.detach(), y, feat, grad))


def summarize():
    out = {}
    for i, (x, y, feat, grad) in enumerate(calls):
        z = lambda t: torch.zeros_like(t) if t.grad is None else t.grad
        out["x%d" % i] = x.cpu().numpy()
        out["y%d" % i] = y.detach().cpu().numpy()
        out["dy%d" % i] = z(y).cpu().numpy().reshape(-1)
        out["dg%d" % i] = z(grad).cpu().numpy()
        out["df%d" % i] = z(feat).norm(dim=-1).cpu().numpy()
    return out


if sys.argv[1] == "dump":
    from oracle import iron_ref as R
    from oracle import train_ref as T
    from _util import cpu_sd, golden_meta, tables
    mt, md = tables()
    nets = scenes.build_networks("S1")
    sd = {k: T.leaf_state(cpu_sd(nets[k])) for k in NETS}
    sc = R.Scene(sd["sdf_network"], R.SDFSpec(), {k: (sd[k], R.GGX_SPECS[k]) for k in R.GGX_SPECS}, golden_meta()["light"], mt, md)
    orig = T.sdf_get_all_train

    def wrapped(sd_, spec, x):
        y, feat, grad = orig(sd_, spec, x)
        record(x, y, feat, grad)
        return y, feat, grad
    T.sdf_get_all_train = wrapped
    torch.set_num_threads(8)
    res = T.render_camera_edges_train(sc, R.CameraSpec(size, size, K, W2C), dem)
    ((res["color"] * wt)[em]).sum().backward()
    out = summarize()
    out["gsdf"] = torch.cat([sd["sdf_network"][k].grad.reshape(-1) for k in sd["sdf_network"] if sd["sdf_network"][k].grad is not None]).numpy()
    np.savez_compressed(PATH, **out)
    print("dumped", len(calls), "get_all calls:", [c[0].shape[0] for c in calls])
else:
    from iron_amd.raytracer import Camera, RayTracer, render_camera
    from iron_amd.renderer_ggx import GGXColocatedRenderer
    from iron_amd.rendering_func import make_render_fn
    ref = dict(np.load(PATH))
    nets = {k: v.cuda() for k, v in scenes.build_networks("S1").items()}
    net = nets["sdf_network"]
    orig = net.get_all

    def wrapped(x, is_training=False):
        y, feat, grad = orig(x, is_training=is_training)
        if is_training:
            record(x, y, feat, grad)
        return y, feat, grad
    net.get_all = wrapped
    cam = Camera(size, size, K.cuda(), W2C.cuda())
    res = render_camera(cam, net, RayTracer(), nets, make_render_fn(GGXColocatedRenderer(use_cuda=True)), fill_holes=False,
                        handle_edges=True, is_training=True, depth_edge_mask=dem.cuda())
    ((res["color"] * wt.cuda())[em.cuda()]).sum().backward()
    mine = summarize()
    print("HIP get_all calls:", [c[0].shape[0] for c in calls], " reference dump:", [ref["x%d" % i].shape[0] for i in range(4) if "x%d" % i in ref])
    # HIP traces both side batches in one call: split it to line up with the dump's (pos, neg)
    n_edge = ref["x1"].shape[0]
    hip = {0: 0, 1: 1}
    parts = {}
    for k in ("x", "y", "dy", "dg", "df"):
        for i in (0, 1):
            parts["%s%d" % (k, i)] = mine["%s%d" % (k, i)]
    # side rays: only the convergent ones are shaded; HIP's order = pos hits then neg hits
    n_pos = ref["x2"].shape[0]
    for k in ("x", "y", "dy", "dg", "df"):
        parts[k + "2"], parts[k + "3"] = mine[k + "2"][:n_pos], mine[k + "2"][n_pos:]
    for i, label in enumerate(("main pixels", "edge points", "pos side hits", "neg side hits")):
        if parts["x%d" % i].shape != ref["x%d" % i].shape:
            print(label, "shape mismatch", parts["x%d" % i].shape, ref["x%d" % i].shape)
            continue
        dx = np.linalg.norm(parts["x%d" % i] - ref["x%d" % i], axis=-1)
        line = "%-14s n %5d  |dx| max %.2e" % (label, dx.size, dx.max() if dx.size else 0)
        for k in ("dy", "dg", "df"):
            a, b = parts["%s%d" % (k, i)], ref["%s%d" % (k, i)]
            a, b = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
            den = np.linalg.norm(b) + 1e-30
            per = np.linalg.norm(a - b, axis=-1)
            line += "   %s rel %.2e (|ref| %.2e, worst point %d: %.2e of |ref|)" % (k, np.linalg.norm(a - b) / den, den, int(per.argmax()) if per.size else -1,
                                                                                (per.max() / den) if per.size else 0)
        print(line)
    gs = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu().numpy()
    print("SDF parameter gradient of term B: rel-L2 %.3e, |hip| %.5e |ref| %.5e" % (np.linalg.norm(gs - ref["gsdf"]) / np.linalg.norm(ref["gsdf"]),
                                                                               np.linalg.norm(gs), np.linalg.norm(ref["gsdf"])))
