"""w16 core prototype: accuracy against the exact-fp32 core and timing against the h2 core (child processes: the core is chosen per process)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, time, torch
sys.path.insert(0, %r)
torch.set_grad_enabled(False)
from iron_amd import scenes
torch.manual_seed(0)
net = scenes.build_networks("S1")["sdf_network"].cuda()
n = 1 << 22
x = torch.rand(n, 3, device="cuda") * 2 - 1
y = net.sdf(x); torch.cuda.synchronize()
ts = []
for r in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): net.sdf(x)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 3 * 1e3)
torch.save(y[:1 << 18].cpu(), sys.argv[1])
print("RESULT", min(ts), sorted(ts)[2])
''' % ROOT
out = {}
for core in ("f32", "h2", "w16"):
    f = "/tmp/y_%s.pt" % core
    r = subprocess.run([sys.executable, "-c", CODE, f], env=dict(os.environ, IRON_MLP_CORE=core), capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    print(core, line[0] if line else ("FAILED\n" + r.stdout[-800:] + r.stderr[-1500:]), flush=True)
import torch
ref = torch.load("/tmp/y_f32.pt").double()
for core in ("h2", "w16"):
    if os.path.exists("/tmp/y_%s.pt" % core):
        y = torch.load("/tmp/y_%s.pt" % core).double()
        print(core, "rel-L2 vs exact-fp32 core %.3e  max|d| %.3e" % (float((y - ref).norm() / ref.norm()), float((y - ref).abs().max())))
