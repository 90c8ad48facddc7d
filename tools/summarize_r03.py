"""Copies the round-3 measurement pass (tools/profile_r03.sh, gpurun_out/r03/) into profiles/: the three bench lines and the
iron kernels' rows of each rocprofv3 --kernel-trace --stats summary (torch / runtime kernels above 0.5 % are kept too)."""
import csv, glob, json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "r03")
out = os.path.join(ROOT, "profiles")
names = {"c1": "r03_bench_1gpu.json", "c1_edges": "r03_bench_1gpu_edges.json", "c2": "r03_bench_c2_neus_forward.json", "c3": "r03_bench_c3_train_step.json"}
for k, n in names.items():
    p = os.path.join(src, "bench_%s.json" % k)
    if os.path.exists(p):
        line = open(p).read().strip().splitlines()[-1]
        json.loads(line)
        open(os.path.join(out, n), "w").write(line + "\n")
stats = {"c1": "r03_h2_kernel_stats.csv", "c2": "r03_c2_kernel_stats.csv", "c3": "r03_c3_train_step_kernel_stats.csv"}
for k, n in stats.items():
    cands = glob.glob(os.path.join(src, "trace_%s" % k, "**", "*kernel_stats.csv"), recursive=True)
    if not cands:
        continue
    rows = list(csv.DictReader(open(cands[0])))
    with open(os.path.join(out, n), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            if "iron" in r["Name"] or float(r["Percentage"]) >= 0.5:
                w.writerow([r[c] for c in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")])
    print(n, len(rows), "kernels;", "Cijk rows:", sum("Cijk" in r["Name"] for r in rows))
