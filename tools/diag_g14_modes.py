"""G14 training render under IRON_GETALL=rev|fwd (two processes): which gradient entries move, and is it a ReLU kink flip?"""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1:
    import torch
    from _util import golden, t
    from iron_amd import scenes
    from iron_amd.raytracer import Camera, RayTracer, render_camera
    from iron_amd.renderer_ggx import GGXColocatedRenderer
    from iron_amd.rendering_func import make_render_fn
    g = golden("g14_train_S1_c32.npz")
    nets = {k: v.cuda() for k, v in scenes.build_networks("S1").items()}
    cam = Camera(int(g["W"]), int(g["H"]), t(g["K"]).cuda(), t(g["W2C"]).cuda())
    res = render_camera(cam, nets["sdf_network"], RayTracer(), nets, make_render_fn(GGXColocatedRenderer(use_cuda=True)), fill_holes=False, handle_edges=False, is_training=True)
    wt = t(g["loss_weights"]).cuda()
    loss = (res["color"] * wt).sum() + 0.1 * (res["normal"] * wt).sum()
    loss.backward()
    out = {"normal": res["normal"].detach().cpu().numpy()}
    for name in ("sdf_network", "diffuse_albedo_network", "specular_albedo_network", "specular_roughness_network"):
        for pname, p in nets[name].named_parameters():
            out["%s/%s" % (name, pname)] = p.grad.cpu().numpy()
    np.savez(sys.argv[1], **out)
    sys.exit(0)
for mode in ("rev", "fwd"):
    subprocess.check_call([sys.executable, __file__, "/tmp/g14_%s.npz" % mode], env=dict(os.environ, IRON_GETALL=mode))
a, b = np.load("/tmp/g14_rev.npz"), np.load("/tmp/g14_fwd.npz")
from _util import golden
g = golden("g14_train_S1_c32.npz")
print("normal max|d| rev-fwd %.3e" % np.abs(a["normal"] - b["normal"]).max())
for k in a.files:
    if k == "normal":
        continue
    d = np.abs(a[k].astype(np.float64) - b[k])
    ref_n = float(g["gnorm:" + k])
    ea = abs(np.linalg.norm(a[k].astype(np.float64)) - ref_n) / ref_n
    eb = abs(np.linalg.norm(b[k].astype(np.float64)) - ref_n) / ref_n
    if max(ea, eb) > 1e-4:
        flat = d.reshape(d.shape[0], -1).max(axis=1) if d.ndim > 1 else d
        top = np.argsort(-flat)[:4]
        print("%-48s norm err rev %.2e fwd %.2e | max|d| %.2e of max %.2e; rows with largest change %s (%s)" % (k, ea, eb, d.max(), np.abs(b[k]).max(), top.tolist(), ["%.1e" % flat[i] for i in top]))
