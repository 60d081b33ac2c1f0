#!/bin/bash
# Per-kernel times of tools/attn_bench.py (rocprofv3 kernel trace); run on the GPU box: bash tools/attn_prof.sh [--long]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT="$ROOT/gpurun_out/prof_attn"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT" -o attn --output-format csv -- python3 "$ROOT/tools/attn_bench.py" "$@" > "$OUT/run.log" 2>&1 || { tail -5 "$OUT/run.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:9.1f} us')
PY
