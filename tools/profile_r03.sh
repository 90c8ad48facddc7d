#!/bin/bash
# Round-3 measurement pass (run on the GPU box from the repo root): the three bench lines, then rocprofv3 --kernel-trace --stats of
# the same commands (no counters in these passes).  Outputs under gpurun_out/r03/; tools/summarize_r03.py copies the summaries.
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
python3 bench.py > "$OUT/bench_c1.json" 2> "$OUT/bench_c1.err" || { tail -5 "$OUT/bench_c1.err"; exit 1; }
echo "c1 bench done"
python3 bench.py --workload c2 > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err" || { tail -5 "$OUT/bench_c2.err"; exit 1; }
echo "c2 bench done"
python3 bench.py --workload c3 > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err" || { tail -5 "$OUT/bench_c3.err"; exit 1; }
echo "c3 bench done"
python3 bench.py --edges --no-extras > "$OUT/bench_c1_edges.json" 2> "$OUT/bench_c1_edges.err" || { tail -5 "$OUT/bench_c1_edges.err"; exit 1; }
echo "c1 --edges bench done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c1" -o c1 -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras > "$OUT/trace_c1.log" 2>&1 || { tail -5 "$OUT/trace_c1.log"; exit 1; }
echo "c1 trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c2" -o c2 -- python3 "$ROOT/bench.py" --workload c2 --no-cpu-baseline > "$OUT/trace_c2.log" 2>&1 || { tail -5 "$OUT/trace_c2.log"; exit 1; }
echo "c2 trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c3" -o c3 -- python3 "$ROOT/bench.py" --workload c3 --no-cpu-baseline > "$OUT/trace_c3.log" 2>&1 || { tail -5 "$OUT/trace_c3.log"; exit 1; }
echo "c3 trace done"
find "$OUT" -name "*kernel_stats.csv"
