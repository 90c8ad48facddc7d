import os, subprocess, sys
ROOT = "/root/repo"
CODE = r'''
import sys, time, torch
sys.path.insert(0, "/root/repo")
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
print("RESULT", min(ts), sorted(ts)[2])
'''
for v in ("", "1", "2", "3", "4"):
    env = dict(os.environ, IRON_MLP_CORE="w16")
    if v: env["IRON_HIP_LIB"] = ROOT + "/variants/libiron_w16v%s.so" % v
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=300)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    print("variant", v or "0", line[0] if line else ("FAILED\n" + r.stdout[-800:] + r.stderr[-1500:]), flush=True)
