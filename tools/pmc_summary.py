"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as the guide requires) per kernel.

    python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.md> <out.json>
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM): reads are doubled.  Infinity-Cache hits are counted as fetches, so "traffic" is an
upper bound of HBM bytes."""
import collections
import csv
import json
import sys


def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            a = agg[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg


f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
rows = []
for k in sorted(f, key=lambda k: -f[k][1]):
    n, v = f[k]
    wn, wv = w.get(k, [0, 0.0])
    rows.append((k, n, 2 * v / n * 1024, wv / max(wn, 1) * 1024))
with open(sys.argv[3], "w") as out:
    out.write("# HBM-side traffic per kernel launch from PMC counters (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)\n\n"
              "Command of both passes: `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile` (B = 256, bf16).\n"
              "Reads = 2 x FETCH_SIZE (gfx950 under-reports wide coalesced reads by half), writes = WRITE_SIZE; KiB -> bytes.\n"
              "Infinity-Cache hits count as fetches: an upper bound of the HBM bytes.\n\n| kernel | launches | read MB / launch | write MB / launch |\n|---|---:|---:|---:|\n")
    for k, n, rd, wr in rows[:24]:
        out.write(f"| `{k[:100]}` | {n} | {rd / 1e6:.1f} | {wr / 1e6:.1f} |\n")
pp = [r for r in rows if "gemm_pp_kernel" in r[0]]
n = sum(r[1] for r in pp)
summary = {"kernel": "gemm_pp_kernel", "launches": n, "read_bytes_per_launch": sum(r[1] * r[2] for r in pp) / n,
           "write_bytes_per_launch": sum(r[1] * r[3] for r in pp) / n}
summary["traffic_bytes_per_launch"] = summary["read_bytes_per_launch"] + summary["write_bytes_per_launch"]
json.dump(summary, open(sys.argv[4], "w"), indent=1)
print(summary)
