#!/bin/bash
# Every tracked profile of a round on one box (GPU box: bash tools/profile_all.sh r04): kernel trace + the two PMC passes per workload
# (tools/profile_round.sh), then the B 4 / C 50 and CoCoOp kernel traces.  Copy gpurun_out/prof_<tag>*/ summaries into profiles/ afterwards
# (tools/profiles_commit.py).
set -o pipefail
R=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
bash tools/profile_round.sh ${R} || exit 1
bash tools/profile_round.sh ${R}_fp32 --dtype fp32 || exit 1
bash tools/profile_round.sh ${R}_fp16 --dtype fp16 || exit 1
bash tools/profile_round.sh ${R}_c1000 --classes 1000 || exit 1
bash tools/profile_round.sh ${R}_vitl --arch vit_l14_336 --batch 128 --classes 1000 || exit 1
echo all profiles done
# kernel traces (no PMC passes) of the reference's own training batch (B 4, 50 classes) and of CoCoOp: gpurun_out/prof_<R>_b4 / _cocoop
for t in b4 cocoop; do
  OUT=$ROOT/gpurun_out/prof_${R}_$t; rm -rf "$OUT"; mkdir -p "$OUT"
  cd /tmp && export TMPDIR=/tmp
  if [ $t = b4 ]; then
    rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t --output-format csv -- python3 "$ROOT/bench.py" --batch 4 --classes 50 --steps 100 --warmup 10 --no-cpu-baseline --no-parity-mode --no-profile > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
  else
    rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t --output-format csv -- python3 "$ROOT/tools/cocoop_bench.py" > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
  fi
  cp "$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)" "$OUT/kernel_stats.csv"; rm -rf "$OUT/trace"; cd "$ROOT"
  tail -1 "$OUT/trace.log" | cut -c1-160
done
