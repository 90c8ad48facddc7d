"""Child process of tests/test_gpu_w16.py: the experimental "w16" core (csrc/w16.hip; IRON_MLP_CORE=w16 routes the batched SDF value
query through it) against the reference golden G2."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
assert os.environ.get("IRON_MLP_CORE") == "w16"

import torch  # noqa: E402

torch.set_grad_enabled(False)
from iron_amd import scenes  # noqa: E402
from _util import golden, rel_l2, t  # noqa: E402

net = scenes.build_networks("S1")["sdf_network"].cuda()
g = golden("g2_sdf.npz")
x = t(g["x"]).cuda()
y = net.sdf(x)[:, 0].cpu().numpy()
r = rel_l2(y, g["sdf"])
assert r <= 2e-6, r
# ragged sizes around the 128-point group and the 16-point wave tile
big = torch.rand(1000, 3, device="cuda") * 2 - 1
full = net.sdf(big)[:, 0]
for n in (1, 15, 16, 17, 127, 128, 129, 513):
    part = net.sdf(big[:n].contiguous())[:, 0]
    assert torch.equal(part, full[:n]), n
print("W16_CHECK OK sdf %.2e" % r)
