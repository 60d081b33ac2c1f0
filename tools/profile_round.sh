#!/bin/bash
# Profile the headline bench command: one kernel-trace pass and two PMC passes (FETCH_SIZE, WRITE_SIZE -- separate runs, never
# combined with tracing), then join them (tools/prof_join.py).  Usage on the GPU box:  bash tools/profile_round.sh <tag> [extra bench args]
set -o pipefail
TAG=${1:-r02}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
TS=8; TW=2; PS=2; PW=1
HB=$(( TS < 5 ? TS : 5 )); [ "$HB" -lt 2 ] && HB=2   # bench.py appends this many steps for the HBM-kernel classes (events on LN / attention)
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t --output-format csv -- python3 "$ROOT/bench.py" --steps $TS --warmup $TW --no-cpu-baseline --no-parity-mode "$@" > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o f --output-format csv -- python3 "$ROOT/bench.py" --steps $PS --warmup $PW --no-cpu-baseline --no-parity-mode --no-profile "$@" > "$OUT/fetch.log" 2>&1 || { tail -5 "$OUT/fetch.log"; exit 1; }
rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o w --output-format csv -- python3 "$ROOT/bench.py" --steps $PS --warmup $PW --no-cpu-baseline --no-parity-mode --no-profile "$@" > "$OUT/write.log" 2>&1 || { tail -5 "$OUT/write.log"; exit 1; }
T=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1); F=$(find "$OUT/fetch" -name "*counter_collection.csv" | head -1); W=$(find "$OUT/write" -name "*counter_collection.csv" | head -1)
STAMP=$(cd "$ROOT" && python3 -c "from mudpt_amd import build; print(build.source_hash())")
python3 "$ROOT/tools/prof_join.py" --trace "$T" --fetch "$F" --write "$W" --steps $((TS + TW + HB)) --pmc-steps $((PS + PW)) --stamp "$STAMP" \
    --title "HBM-side bytes per kernel joined with the kernel trace ($TAG)" \
    --note "Command: \`python3 bench.py --steps $TS --warmup $TW --no-cpu-baseline --no-parity-mode $*\` under \`rocprofv3 --kernel-trace --stats\` ($((TS + TW + HB)) steps in the trace: warm-up, timed, and the $HB steps of the HBM-kernel pass); PMC passes: the same with \`--steps $PS --warmup $PW --no-profile\` under \`--pmc FETCH_SIZE\` / \`--pmc WRITE_SIZE\`." \
    --out-md "$OUT/bytes_per_step.md" --out-json "$OUT/bytes_per_step.json" > "$OUT/join.log" 2>&1 || { tail -5 "$OUT/join.log"; exit 1; }
cp "$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)" "$OUT/kernel_stats.csv"
tail -1 "$OUT/trace.log"
# keep what travels back small: the per-dispatch CSVs are tens of MB
rm -rf "$OUT/trace" "$OUT/fetch" "$OUT/write"
