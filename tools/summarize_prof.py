"""Condenses the rocprofv3 CSVs under gpurun_out/prof (tools/profile.sh) into the committed summaries:
   profiles/<tag>_kernel_stats.csv   (iron kernels only, from --kernel-trace --stats)
   profiles/<tag>_pmc_summary.csv    (per-kernel means of every collected counter + derived figures)
   profiles/hbm_traffic.json         (HBM bytes per launch: (2*FETCH_SIZE + WRITE_SIZE) KB, the gfx950
                                      FETCH_SIZE half-count correction of MI355X_MICROARCH.md applied)
"""
import collections, csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)

def short(name):
    n = name.split("(")[0].replace("void ", "").replace("iron::", "")
    return n

rows = list(csv.DictReader(open(os.path.join(src, "trace", "r01_kernel_stats.csv"))))
with open(os.path.join(out, tag + "_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_ms", "avg_ms", "percent", "min_ms", "max_ms"])
    for r in rows:
        if "iron::" not in r["Name"]:
            continue
        w.writerow([short(r["Name"]), r["Calls"], "%.3f" % (float(r["TotalDurationNs"]) / 1e6), "%.4f" % (float(r["AverageNs"]) / 1e6),
                    r["Percentage"], "%.4f" % (float(r["MinNs"]) / 1e6), "%.4f" % (float(r["MaxNs"]) / 1e6)])

agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in ("pmc_fetch", "pmc_write", "pmc_mfma"):
    p = os.path.join(src, d, "r01_counter_collection.csv")
    if not os.path.exists(p):
        continue
    for r in csv.DictReader(open(p)):
        if "iron::" not in r["Kernel_Name"]:
            continue
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if d == "pmc_mfma" and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
traffic = {}
with open(os.path.join(out, tag + "_pmc_summary.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches", "FETCH_SIZE_KB", "WRITE_SIZE_KB", "hbm_MB_per_launch(2*FETCH+WRITE)", "GRBM_GUI_ACTIVE(sum 8 XCD)",
                "eff_clock_GHz", "SQ_VALU_MFMA_BUSY_CYCLES", "mfma_busy_frac(of 1024 SIMD x cycles)", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"])
    for k, d in sorted(agg.items()):
        m = {c: sum(v) / len(v) for c, v in d.items()}
        fetch, write = m.get("FETCH_SIZE", 0.0), m.get("WRITE_SIZE", 0.0)
        hbm = (2 * fetch + write) * 1024
        cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        ms = sum(dur[k]) / len(dur[k]) if dur[k] else 0.0
        clock = cyc / (ms * 1e6) if ms else 0.0
        busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        frac = busy / (1024 * cyc) if cyc else 0.0
        w.writerow([k, len(d.get("FETCH_SIZE", d.get("GRBM_GUI_ACTIVE", []))), "%.1f" % fetch, "%.1f" % write, "%.2f" % (hbm / 1e6), "%.4g" % m.get("GRBM_GUI_ACTIVE", 0),
                    "%.3f" % clock, "%.4g" % busy, "%.3f" % frac, "%.4g" % m.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0), "%.4g" % m.get("SQ_WAVE_CYCLES", 0), "%.4g" % m.get("SQ_BUSY_CYCLES", 0)])
        base = k.split("<")[0].strip('"')  # template arguments (backend) and the _h2 suffix name the same bench kernel class
        key = {"k_sphere": "sphere", "k_sampler": "sampler", "k_bisect_a": "bisect_a", "k_bisect_b": "bisect_b", "k_sdf_grad": "sdf_grad",
               "k_sdf_grad_h2": "sdf_grad", "k_ggx_shade": "ggx"}.get(base, "material" if base.startswith("k_material") else None)
        if key and (key != "material" or key not in traffic):
            traffic[key] = {"bytes_per_launch": hbm, "fetch_size_kb": fetch, "write_size_kb": write, "mfma_busy_frac": frac, "eff_clock_ghz": clock,
                            "source": "profiles/%s_pmc_summary.csv" % tag}
json.dump(traffic, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1, sort_keys=True)
print(open(os.path.join(out, tag + "_pmc_summary.csv")).read())
print(open(os.path.join(out, tag + "_kernel_stats.csv")).read())
