"""Count instructions by kind inside every loop of one kernel of a hipcc -S listing (where does a loop's issue budget go?).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip -S --cuda-device-only mudpt_amd/csrc/attention.hip -o /tmp/attn.s
    python tools/isa_loops.py /tmp/attn.s _ZN5mudpt21attn_bwd_sweep_kernelINS_4BF16ELi7EEEvNS_8AttnArgsEPKv
"""
import re
import sys
from collections import Counter

txt = open(sys.argv[1]).read().split("\n")
name = sys.argv[2]
start = next(i for i, l in enumerate(txt) if l.startswith(name + ":"))
end = next(i for i in range(start, len(txt)) if "s_endpgm" in txt[i])
lines = txt[start:end + 1]
labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
loops = []
for i, l in enumerate(lines):
    m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
print(f"{name}: {len(lines)} lines, loops {loops}")
for a, b in loops:
    c, detail = Counter(), Counter()
    for l in lines[a:b + 1]:
        l = l.strip()
        if not l or l[0] in ";." or l.endswith(":"):
            continue
        op = l.split()[0]
        if op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1; detail[op] += 1
        elif op.startswith(("v_exp", "v_rcp", "v_log", "v_rsq", "v_sqrt")):
            c["valu transcendental"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1; detail[op] += 1
        elif op.startswith("s_waitcnt"):
            c["s_waitcnt"] += 1
        elif op.startswith(("buffer_", "global_", "flat_", "scratch_")):
            c["vmem"] += 1; detail[op] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    print(f"--- loop at lines {a}..{b}: " + ", ".join(f"{k} {v}" for k, v in c.most_common()))
    print("    " + ", ".join(f"{k} {v}" for k, v in detail.most_common(24)))
