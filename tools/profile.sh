#!/bin/bash
# rocprofv3 passes for bench.py (run on the GPU box from the repo root):
#   1. --kernel-trace --stats              per-kernel device time (profiles/*_kernel_stats.csv)
#   2..4 separate --pmc passes             HBM traffic (FETCH_SIZE / WRITE_SIZE) and MFMA activity
# Counters are collected in their own runs, never together with sys/runtime traces.
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/prof"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
STEPS="${STEPS:-5}"
ARGS="$ROOT/bench.py --steps $STEPS --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o r01 -- python3 $ARGS > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
echo "trace pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o r01 -- python3 $ARGS > "$OUT/pmc_fetch.log" 2>&1 || { tail -5 "$OUT/pmc_fetch.log"; exit 1; }
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o r01 -- python3 $ARGS > "$OUT/pmc_write.log" 2>&1 || { tail -5 "$OUT/pmc_write.log"; exit 1; }
echo "write pass done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_mfma" -o r01 -- python3 $ARGS > "$OUT/pmc_mfma.log" 2>&1 || { tail -5 "$OUT/pmc_mfma.log"; exit 1; }
echo "mfma pass done"
find "$OUT" -name "*.csv" | head -40
