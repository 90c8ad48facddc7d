#!/bin/bash
# HBM traffic counters of the three bench workloads (run on the GPU box from the repo root).  One counter per pass (FETCH_SIZE and
# WRITE_SIZE do not fit one pass), --kernel-trace only beside them, the program itself behind `--`.
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03_pmc"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for wl in c1 c2 c3; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    if [ "$wl" = c1 ]; then ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extras"; else ARGS="--workload $wl --steps 3 --warmup 1 --no-cpu-baseline"; fi
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$OUT/${wl}_${ctr}" -o p -- python3 "$ROOT/bench.py" $ARGS > "$OUT/${wl}_${ctr}.log" 2>&1 || { tail -5 "$OUT/${wl}_${ctr}.log"; exit 1; }
    echo "$wl $ctr done"
  done
done
