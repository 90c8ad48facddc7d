"""Accuracy of the HIP SDF forward vs the CPU oracle (fp32) and an fp64 evaluation of the same weights."""
import os, sys
import torch
torch.set_grad_enabled(False)  # measurement / inspection of the inference kernels: nothing is attached
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from iron_amd import scenes
from oracle import iron_ref as R
from _util import oracle_scene
for scene in ("S0", "S1"):
    nets = scenes.build_networks(scene)
    sc = oracle_scene(nets)
    g = torch.Generator().manual_seed(3)
    x = torch.rand(8192, 3, generator=g) * 2 - 1
    ref = R.sdf_forward(sc.sdf_sd, sc.sdf_spec, x)
    sd64 = {k: v.double() for k, v in sc.sdf_sd.items()}
    ref64 = R.sdf_forward(sd64, sc.sdf_spec, x.double())
    gnet = nets["sdf_network"].cuda()
    out = gnet(x.cuda()).cpu()
    only = gnet.sdf(x.cuda()).cpu()[:, 0]
    def stats(a, b):
        a = a.double(); b = b.double()
        return "max|d| %.3e  relL2 %.3e" % ((a - b).abs().max().item(), ((a - b).norm() / b.norm()).item())
    print(scene, "sdf   hip~ref32:", stats(out[:, 0], ref[:, 0]), "| hip~ref64:", stats(out[:, 0], ref64[:, 0]), "| ref32~ref64:", stats(ref[:, 0], ref64[:, 0]))
    print(scene, ".sdf() hip~ref32:", stats(only, ref[:, 0]), "| hip~ref64:", stats(only, ref64[:, 0]), "  (core:", os.environ.get("IRON_MLP_CORE", "h2 (default)"), ")")
    print(scene, "feat  hip~ref32:", stats(out[:, 1:], ref[:, 1:]), "| hip~ref64:", stats(out[:, 1:], ref64[:, 1:]), "| ref32~ref64:", stats(ref[:, 1:], ref64[:, 1:]))
