"""gpurun_out/r03_pmc (tools/pmc_r03.sh) -> profiles/r03_pmc_summary.csv and profiles/hbm_traffic.json.
HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB: on gfx950 FETCH_SIZE tallies a wide coalesced read at half its bytes
(MI355X_MICROARCH.md, HBM / rocprofv3 section); WRITE_SIZE is exact for streaming stores."""
import collections, csv, glob, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "r03_pmc")
out = os.path.join(ROOT, "profiles")


def short(name):
    return name.split("(")[0].replace("void ", "").replace("iron::", "").replace("iron_train::", "")


rows_out, traffic = [], {}
for wl in ("c1", "c2", "c3"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = glob.glob(os.path.join(src, "%s_%s" % (wl, ctr), "**", "*counter_collection.csv"), recursive=True)
        if not fs:
            continue
        for r in csv.DictReader(open(fs[0])):
            if "iron" not in r["Kernel_Name"] or r["Counter_Name"] != ctr:
                continue
            agg[short(r["Kernel_Name"])][ctr].append(float(r["Counter_Value"]))
    for k, d in sorted(agg.items()):
        f = sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1)
        w = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1)
        # the largest launches of a kernel are the workload's (the same kernel also runs on small side batches)
        fmax = max(d["FETCH_SIZE"]) if d["FETCH_SIZE"] else 0.0
        wmax = max(d["WRITE_SIZE"]) if d["WRITE_SIZE"] else 0.0
        rows_out.append([wl, k, len(d["FETCH_SIZE"]), "%.1f" % f, "%.1f" % w, "%.2f" % ((2 * f + w) * 1024 / 1e6), "%.2f" % ((2 * fmax + wmax) * 1024 / 1e6)])
        base = k.split("<")[0]
        key = None
        if wl == "c1":
            key = {"k_sphere": "sphere", "k_sampler": "sampler", "k_bisect_a": "bisect_a", "k_bisect_b": "bisect_b", "k_sdf_grad_h2": "sdf_grad", "k_sdf_getall_rev_h2": "sdf_grad",
                   "k_ggx_shade": "ggx"}.get(base, "material" if base.startswith("k_material") and "material" not in traffic else None)
        elif wl == "c2" and base in ("k_sdf_grad_h2", "k_sdf_getall_rev_h2"):
            key = "c2_sdf_grad"
        elif wl == "c3" and k.startswith("k_gemm_rows<2, 0>"):
            key = "c3_gemm_rows_probe"
        if key:
            use_max = key in ("c2_sdf_grad", "c3_gemm_rows_probe")
            ff, ww = (fmax, wmax) if use_max else (f, w)
            traffic[key] = {"bytes_per_launch": (2 * ff + ww) * 1024, "fetch_size_kb": ff, "write_size_kb": ww, "launches": len(d["FETCH_SIZE"]),
                            "source": "profiles/r03_pmc_summary.csv (%s passes, %s launches)" % (wl, "largest" if use_max else "mean of all")}
with open(os.path.join(out, "r03_pmc_summary.csv"), "w", newline="") as fcsv:
    w = csv.writer(fcsv)
    w.writerow(["workload", "kernel", "launches", "FETCH_SIZE_KB(mean)", "WRITE_SIZE_KB(mean)", "hbm_MB_per_launch(2*FETCH+WRITE, mean)", "hbm_MB(largest launch)"])
    w.writerows(rows_out)
old = {}
tp = os.path.join(out, "hbm_traffic.json")
if os.path.exists(tp):
    old = json.load(open(tp))
old.update(traffic)
json.dump(old, open(tp, "w"), indent=1, sort_keys=True)
for r in rows_out:
    print(r)
