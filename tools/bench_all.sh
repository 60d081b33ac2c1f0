#!/bin/bash
# Every workload of DESIGN.md 5 on one box, one JSON / text line each into gpurun_out/bench_all.log (GPU box: bash tools/bench_all.sh)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out; LOG=gpurun_out/bench_all.log; : > $LOG
run() { echo "## $*" >> $LOG; "$@" 2>&1 | tail -1 >> $LOG || exit 1; }
run python bench.py
run python bench.py --dtype fp16 --no-cpu-baseline --no-parity-mode
run python bench.py --dtype fp32 --no-cpu-baseline --no-parity-mode
run python bench.py --classes 1000 --no-cpu-baseline --no-parity-mode
run python bench.py --arch vit_l14_336 --batch 128 --classes 1000 --steps 8 --warmup 3 --no-cpu-baseline --no-parity-mode
run python bench.py --batch 4 --classes 50 --steps 50 --warmup 10 --no-cpu-baseline --no-parity-mode --no-profile
run python tools/cocoop_bench.py
run python tools/plugin_bench.py
