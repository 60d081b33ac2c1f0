"""gpurun_out/prof_<tag>/ (tools/profile_round.sh / tools/profile_all.sh on the GPU box) -> the summaries tracked under profiles/:
<tag>_kernel_stats.md (rocprofv3 --kernel-trace --stats), <tag>_bytes_per_step.{md,json} (the PMC join).
    python tools/profiles_commit.py r04 "headline: BASELINE configs[1], ViT-B/16 B 256 C 11 bf16" [r04_fp32 "..." ...]"""
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    for tag, what in zip(args[0::2], args[1::2]):
        src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
        rnd = re.match(r"r(\d+)", tag).group(1).lstrip("0")
        ms = ""
        log = os.path.join(src, "trace.log")
        if os.path.exists(log):
            line = next((l for l in open(log) if l.startswith("{")), None)
            if line:
                ms = f"; ms per step under the tracer: {json.loads(line)['ms_per_step']}"
        note = (f"Round {rnd}, end state.  15 steps in the trace (2 warm-up + 8 timed + 5 steps of the HBM-kernel pass){ms} (the tracer and the event pairs on every launch "
                f"of the HBM pass slow the step by a few %; unprofiled numbers are in DESIGN.md 5).  Command: `bash tools/profile_round.sh {tag}` "
                "(tools/profile_round.sh shows the bench.py arguments).") if os.path.exists(os.path.join(src, "bytes_per_step.json")) else \
               f"Round {rnd}, end state.  Command: see tools/profile_all.sh / DESIGN.md 5."
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "prof_summary.py"), os.path.join(src, "kernel_stats.csv"),
                               os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.md"), f"rocprofv3 --kernel-trace --stats: {what}", note])
        for ext in ("md", "json"):
            f = os.path.join(src, f"bytes_per_step.{ext}")
            if os.path.exists(f):
                shutil.copy(f, os.path.join(ROOT, "profiles", f"{tag}_bytes_per_step.{ext}"))
        print("profiles/" + tag)


if __name__ == "__main__":
    main()
