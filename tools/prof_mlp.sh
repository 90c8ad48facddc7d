#!/bin/bash
# PMC passes on the batched SDF-MLP micro-benchmark (tools/bench_mlp.py); IRON_MLP_CORE selects the core.
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/prof_mlp"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/tools/bench_mlp.py --n 2097152 --iters 2"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d "$OUT/p1" -o m -- python3 $ARGS > "$OUT/p1.log" 2>&1 || { tail -5 "$OUT/p1.log"; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/p2" -o m -- python3 $ARGS > "$OUT/p2.log" 2>&1 || { tail -5 "$OUT/p2.log"; exit 1; }
python3 - <<PY
import csv, collections
for d in ("p1","p2"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open("$OUT/%s/m_counter_collection.csv" % d)):
        if "k_sdf_values" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print("%-28s %.4g" % (k, sum(v)/len(v)))
PY
