"""Kernel A/B harness.

  python tools/variants.py build NAME [hipcc flags...]   (here: cross-compiles into iron_amd/csrc/build/variants/)
  python tools/variants.py run [--n N] [--rounds R] [--trace]   (GPU box: every built variant, interleaved rounds)

Each variant is the whole library compiled with extra flags / -D switches; `run` loads them one per subprocess through
IRON_HIP_LIB, interleaves the rounds (clock drift hits all arms alike) and prints median / min per arm plus the SDF
rel-L2 of each arm against the exact-fp32 core of the default build.
"""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "iron_amd", "csrc", "build", "variants")


def build(name, flags):
    from iron_amd import build as B
    os.makedirs(os.path.join(VDIR, name), exist_ok=True)
    hipcc = B._hipcc()
    only = [x for x in os.environ.get("VARIANT_ONLY", "").split(",") if x]  # restrict the flags to these sources
    objs, procs = [], []

    def start(src, fl):
        obj = os.path.join(VDIR, name, src.replace(".hip", ".o"))
        return obj, subprocess.Popen([hipcc] + B.BASE_FLAGS + fl + ["-c", os.path.join(B.CSRC, src), "-o", obj],
                                     stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)

    for src in B.SOURCES:   # every unit gets the product's flag set; the variant's flags go to all units or to VARIANT_ONLY's
        fl = list(B.OPTIONAL_FLAGS) + list(B.SOURCE_FLAGS.get(src, [])) + (list(flags) if (not only or src in only) else [])
        obj, p = start(src, fl)
        objs.append(obj)
        procs.append((src, fl, p))
    for src, fl, p in procs:
        out = p.communicate()[0]
        if p.returncode != 0 and "-amdgpu-mfma-vgpr-form=1" in fl:  # same fallback as iron_amd/build.py
            print("  %s: hipcc failed with the vgpr-form option, recompiling without" % src)
            fl2 = [f for i, f in enumerate(fl) if f != "-amdgpu-mfma-vgpr-form=1" and not (f == "-mllvm" and fl[i + 1] == "-amdgpu-mfma-vgpr-form=1")]
            _, p2 = start(src, fl2)
            out = p2.communicate()[0]
            if p2.returncode != 0:
                sys.exit(out)
        elif p.returncode != 0:
            sys.exit(out)
    lib = os.path.join(VDIR, "libiron_hip_%s.so" % name)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    json.dump({"flags": list(flags)}, open(os.path.join(VDIR, name, "flags.json"), "w"))
    print("built", lib)


WORKER = r'''
import json, os, sys, time, torch
sys.path.insert(0, %(root)r)
from iron_amd import scenes
n = %(n)d
torch.manual_seed(0)
net = scenes.build_networks("S1")["sdf_network"].cuda()
x = torch.rand(n, 3, device="cuda") * 2 - 1
y = net.sdf(x); torch.cuda.synchronize()
out = {}
ts = []
for r in range(%(iters)d):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): net.sdf(x)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 3 * 1e3)
out["mlp_ms"] = ts
torch.save(y[:65536].cpu(), %(ydump)r)
if %(trace)d:
    from iron_amd.raytracer import Camera, RayTracer, render_camera
    from iron_amd.renderer_ggx import GGXColocatedRenderer
    from iron_amd.rendering_func import make_render_fn
    nets = {k: v.cuda() for k, v in scenes.build_networks("S0").items()}
    K, W2C = scenes.fixture_camera_matrices(800, 800)
    cam = Camera(800, 800, K.cuda(), W2C.cuda())
    fn = make_render_fn(GGXColocatedRenderer(use_cuda=True)); tr = RayTracer()
    f = lambda: render_camera(cam, nets["sdf_network"], tr, nets, fn, fill_holes=False, handle_edges=False)
    f(); torch.cuda.synchronize()
    fs = []
    for r in range(%(iters)d):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); f(); torch.cuda.synchronize()
        fs.append((time.perf_counter() - t0) / 2 * 1e3)
    out["frame_ms"] = fs
print("RESULT " + json.dumps(out))
'''


def run(n, rounds, trace):
    import torch
    arms = {"default": os.path.join(ROOT, "iron_amd", "csrc", "libiron_hip.so")}
    if os.path.isdir(VDIR):
        for f in sorted(os.listdir(VDIR)):
            if f.startswith("libiron_hip_") and f.endswith(".so") and "stamp" not in f:  # stamp builds: tools/stamps.py only
                arms[f[len("libiron_hip_"):-3]] = os.path.join(VDIR, f)
    res = {a: {"mlp_ms": [], "frame_ms": []} for a in arms}
    tmp = os.path.join(ROOT, "gpurun_out")
    os.makedirs(tmp, exist_ok=True)

    def one(arm, lib, env_extra, ydump):
        env = dict(os.environ, IRON_HIP_LIB=lib, **env_extra)
        code = WORKER % dict(root=ROOT, n=n, iters=3, ydump=ydump, trace=int(trace))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        for line in r.stdout.splitlines():
            if line.startswith("RESULT "):
                return json.loads(line[7:])
        print("arm %s failed:\n%s\n%s" % (arm, r.stdout[-2000:], r.stderr[-2000:]))
        return None

    ref = os.path.join(tmp, "y_f32core.pt")
    one("f32core", arms["default"], {"IRON_MLP_CORE": "f32"}, ref)
    yref = torch.load(ref).double()
    for rd in range(rounds):
        for a, lib in arms.items():
            yd = os.path.join(tmp, "y_%s.pt" % a)
            o = one(a, lib, {}, yd)
            if o is None:
                continue
            res[a]["mlp_ms"] += o["mlp_ms"]
            res[a]["frame_ms"] += o.get("frame_ms", [])
            y = torch.load(yd).double()
            res[a]["rel_l2_vs_f32core"] = float((y - yref).norm() / yref.norm())
    for a, r in res.items():
        if not r["mlp_ms"]:
            continue
        line = "%-14s mlp %d pts: median %.3f ms  min %.3f" % (a, n, statistics.median(r["mlp_ms"]), min(r["mlp_ms"]))
        if r["frame_ms"]:
            line += "   frame 800^2: median %.2f ms  min %.2f" % (statistics.median(r["frame_ms"]), min(r["frame_ms"]))
        line += "   rel-L2 vs f32 core %.2e" % r.get("rel_l2_vs_f32core", float("nan"))
        print(line, flush=True)


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "build":
        build(sys.argv[2], sys.argv[3:])
    elif len(sys.argv) >= 2 and sys.argv[1] == "run":
        import argparse
        ap = argparse.ArgumentParser()
        ap.add_argument("cmd")
        ap.add_argument("--n", type=int, default=1 << 22)
        ap.add_argument("--rounds", type=int, default=3)
        ap.add_argument("--trace", action="store_true")
        a = ap.parse_args()
        run(a.n, a.rounds, a.trace)
    else:
        sys.exit(__doc__)
