"""Timing of iron_sdf_get_all (k_sdf_grad_h2) for the product library and ablation builds under variants/ (garbage results, timing only)."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, time, torch
sys.path.insert(0, %r)
torch.set_grad_enabled(False)
from iron_amd import scenes
torch.manual_seed(0)
net = scenes.build_networks("S1")["sdf_network"].cuda()
n = 1 << 19
x = torch.rand(n, 3, device="cuda") * 2 - 1
net.get_all(x, is_training=False); torch.cuda.synchronize()
ts = []
for r in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): net.get_all(x, is_training=False)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 3 * 1e3)
print("RESULT", min(ts), sorted(ts)[2])
''' % ROOT
for lib in [""] + sorted(glob.glob(os.path.join(ROOT, "variants", "*.so"))):
    env = dict(os.environ)
    if lib: env["IRON_HIP_LIB"] = lib
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=300)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    print(os.path.basename(lib) or "product", line[0] if line else ("FAILED\n" + r.stdout[-800:] + r.stderr[-1500:]), flush=True)
