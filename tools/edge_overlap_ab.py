"""A/B of the edge overlap (bench.py --edges) over IRON_EDGE_SIDE_CUS / IRON_EDGE_OVERLAP settings, interleaved rounds."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
arms = [("off", {"IRON_EDGE_OVERLAP": "0"})] + [("side%d" % n, {"IRON_EDGE_SIDE_CUS": str(n)}) for n in (32, 48, 64, 80)]
res = {a: [] for a, _ in arms}
for rnd in range(3):
    for name, env in arms:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--edges", "--no-extras", "--no-cpu-baseline", "--steps", "8"],
                           env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        try:
            res[name].append(json.loads(r.stdout.strip().splitlines()[-1])["ms_per_step"])
        except Exception:
            print(name, "FAILED", r.stderr[-800:])
for name, v in res.items():
    print("%-8s" % name, " ".join("%.2f" % x for x in v), " min %.2f" % min(v) if v else "")
