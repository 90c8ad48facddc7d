#!/bin/bash
# I-cache / instruction-fetch PMC passes on the batched SDF-MLP kernel (tools/bench_mlp.py).
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/prof_icache"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/tools/bench_mlp.py --n 2097152 --iters 2"
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_IFETCH_LEVEL SQ_IFETCH SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/p$i" -o m -- python3 $ARGS > "$OUT/p$i.log" 2>&1 || echo "pass $i failed: $(tail -2 $OUT/p$i.log)"
done
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_sdf_values" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print("%-32s %.5g" % (k, sum(v)/len(v)))
PY
