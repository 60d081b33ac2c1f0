#!/bin/bash
# Every tracked profile of a round on one box (GPU box: bash tools/profile_all.sh r04): kernel trace + the two PMC passes per workload
# (tools/profile_round.sh), B 4 / C 50 and CoCoOp kernel traces.  Copy gpurun_out/prof_<tag>*/ summaries into profiles/ afterwards
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
