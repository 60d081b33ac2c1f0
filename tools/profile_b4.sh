#!/bin/bash
# Kernel trace of BASELINE configs[0]'s shape on the GPU (B 4, 50 classes): per-kernel stats + the timeline of one step's main queue.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_b4; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t --output-format csv -- python3 "$ROOT/bench.py" --batch 4 --classes 50 --steps 30 --warmup 5 --no-cpu-baseline --no-parity-mode --no-profile > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
cp "$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)" "$OUT/kernel_stats.csv"
python3 "$ROOT/tools/step_timeline.py" "$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)" 35 > "$OUT/timeline.md"
tail -1 "$OUT/trace.log" | cut -c1-200; tail -1 "$OUT/timeline.md"
rm -rf "$OUT/trace"
