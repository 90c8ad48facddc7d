"""Counts the reference algorithm's work (E = SDF evaluations, H = hits) with the CPU oracle
for the bench workloads and writes tests/golden/work_counts.json.  Build-container tool (minutes of CPU)."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from iron_amd import scenes
from oracle import iron_ref as R
from _util import oracle_scene
torch.set_num_threads(8)
out_path = os.path.join(ROOT, "tests", "golden", "work_counts.json")
out = json.load(open(out_path)) if os.path.exists(out_path) else {}
for scene, res in [("S0", 800), ("S1", 800), ("S0", 200), ("S1", 200)]:
    key = "%s_%d" % (scene, res)
    if key in out:
        continue
    sc = oracle_scene(scenes.build_networks(scene))
    K, W2C = scenes.fixture_camera_matrices(res, res)
    st = {}
    t0 = time.time()
    tr = R.raytrace_camera(sc, R.CameraSpec(res, res, K, W2C), max_num_rays=50000, stats=st)
    out[key] = {"rays": res * res, "E": sc.counter.evals, "H": int(tr["convergent_mask"].sum()),
                "n_sampler": st["n_sampler"], "bisect_iters": st["bisect_iters"], "oracle_trace_seconds_8threads": time.time() - t0}
    print(key, out[key], flush=True)
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
