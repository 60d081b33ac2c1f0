#!/bin/bash
# SQ / GRBM and LDS counters of the headline step, two separate --pmc passes (never combined with tracing beyond --kernel-trace):
# GPU box: bash tools/profile_sq.sh r04  ->  gpurun_out/prof_<R>_sq/pmc_sq.md (copy to profiles/<R>_pmc_sq.md)
set -o pipefail
R=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_${R}_sq; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/sq" -o sq --output-format csv -- \
    python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-parity-mode > "$OUT/sq.log" 2>&1 || { tail -5 "$OUT/sq.log"; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT -d "$OUT/lds" -o lds --output-format csv -- \
    python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-parity-mode > "$OUT/lds.log" 2>&1 || { tail -5 "$OUT/lds.log"; exit 1; }
python3 "$ROOT/tools/pmc_sq_summary.py" "$(find "$OUT/sq" -name "*counter_collection.csv" | head -1)" "$OUT/pmc_sq.md" "$(find "$OUT/lds" -name "*counter_collection.csv" | head -1)"
rm -rf "$OUT/sq" "$OUT/lds"
head -12 "$OUT/pmc_sq.md" | cut -c1-200
