"""Child process of tests/test_gpu_f32core.py: the exact-fp32 MFMA core (IRON_MLP_CORE=f32, selected once per process
when the library decides its backend) against the reference goldens."""
import os
import sys


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
assert os.environ.get("IRON_MLP_CORE") == "f32"

import torch  # noqa: E402

torch.set_grad_enabled(False)  # the inference kernels (under grad mode the same calls return attached tensors)
from iron_amd import scenes  # noqa: E402
from iron_amd.raytracer import Camera, RayTracer, render_camera  # noqa: E402
from iron_amd.renderer_ggx import GGXColocatedRenderer  # noqa: E402
from iron_amd.rendering_func import make_render_fn  # noqa: E402
from _util import golden, rel_l2, t  # noqa: E402

nets = {k: v.cuda() for k, v in scenes.build_networks("S1").items()}
g = golden("g2_sdf.npz")
x = t(g["x"]).cuda()
y = nets["sdf_network"].sdf(x)[:, 0].cpu().numpy()
r_sdf = rel_l2(y, g["sdf"])
_, feat, grad = nets["sdf_network"].get_all(x, is_training=False)
r_grad = rel_l2(grad.cpu().numpy(), g["getall_grad"])
assert r_sdf <= 1e-5 and r_grad <= 2e-5, (r_sdf, r_grad)

g = golden("g67_S1_c0.npz")
cam = Camera(int(g["W"]), int(g["H"]), t(g["K"]).cuda(), t(g["W2C"]).cuda())
fn = make_render_fn(GGXColocatedRenderer(use_cuda=True))
res = render_camera(cam, nets["sdf_network"], RayTracer(), nets, fn, fill_holes=False, handle_edges=False)
conv = res["convergent_mask"].cpu().numpy()
flips = int((conv != g["convergent_mask"]).sum())
both = conv & g["convergent_mask"]
r_col = rel_l2(res["color"].cpu().numpy()[both], g["color"][both])
assert flips == 0 and r_col <= 1e-4, (flips, r_col)
# stage-1 networks on the exact core (k_material<10, 4, ..., skip>, k_nerf): the h2 kernels are the default elsewhere
from iron_amd.fields import NeRF, RenderingNetwork, SDFNetwork, SingleVarianceNetwork  # noqa: E402

torch.manual_seed(0)
SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5, scale=1.0, geometric_init=True, weight_norm=True)
cnet = RenderingNetwork(d_feature=256, mode="idr", d_in=9, d_out=3, d_hidden=256, n_layers=8, skip_in=[4], weight_norm=True,
                        multires=10, multires_view=4, squeeze_out=True).cuda()
nerf = NeRF(D=8, d_in=4, d_in_view=3, W=256, multires=10, multires_view=4, output_ch=4, skips=[4], use_viewdirs=True).cuda()
g = golden("g13_neus.npz")
co = cnet(t(g["color_pts"]).cuda(), t(g["color_nrm"]).cuda(), t(g["color_view"]).cuda(), t(g["color_feat"]).cuda())
na, nc = nerf(t(g["nerf_pts"]).cuda(), t(g["nerf_views"]).cuda())
r_c1, r_na, r_nc = rel_l2(co.cpu().numpy(), g["color_out"]), rel_l2(na.cpu().numpy(), g["nerf_alpha"]), rel_l2(nc.cpu().numpy(), g["nerf_rgb"])
assert r_c1 <= 1e-5 and r_na <= 1e-5 and r_nc <= 1e-5, (r_c1, r_na, r_nc)
print("F32CORE_CHECK OK sdf %.2e grad %.2e colour %.2e flips %d" % (r_sdf, r_grad, r_col, flips))
