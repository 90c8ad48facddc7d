"""bench.py (C1, no extras) once per library under iron_amd/csrc/build/variants/, interleaved: A/B of whole-frame effects."""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sorted(glob.glob(os.path.join(ROOT, "iron_amd", "csrc", "build", "variants", "*.so")))
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ, IRON_HIP_LIB=lib)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-extras", "--no-cpu-baseline", "--steps", "8"], env=env,
                           capture_output=True, text=True, timeout=600)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print(os.path.basename(lib), "%.3f Mrays/s %.2f ms" % (d["value"], d["ms_per_step"]),
                  {k: round(v["ms_avg"], 2) for k, v in d["kernels"].items()}, "executed", d["frame"]["E_executed"], "hits", d["frame"]["H_device"], flush=True)
        except Exception:
            print(os.path.basename(lib), "FAILED", r.stdout[-500:], r.stderr[-1500:], flush=True)
