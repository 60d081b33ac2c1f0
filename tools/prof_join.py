"""Join a rocprofv3 kernel trace with two PMC passes into ONE per-kernel table: duration, launches per step, HBM-side bytes, GB/s.

    python tools/prof_join.py --trace <kernel_trace.csv> --fetch <counter_collection.csv> --write <counter_collection.csv> \
        --steps <steps in the trace> --pmc-steps <steps in each PMC pass> --out-md profiles/rNN_bytes_per_step.md --out-json profiles/rNN_bytes_per_step.json

The three runs are separate invocations of the same bench.py command (the guide: counters in their own passes, never with tracing).
Dispatches are grouped by (kernel name, grid size): the same LayerNorm / attention kernel serves the vision tower (big grids) and the
text tower (small ones), and only like is averaged with like.  Reads = 2 x FETCH_SIZE (gfx950 reports half the bytes of wide coalesced
reads, MI355X_MICROARCH.md HBM section), writes = WRITE_SIZE, both KiB; Infinity-Cache hits count as fetches, so the byte columns are
the traffic that left the L2s -- an upper bound of HBM bytes.  The last table gives, per hardware queue, the kernel time and the idle
time between consecutive kernels (launch gaps) per step."""
import argparse
import collections
import csv
import json
import re


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("mudpt::", "")
    return re.sub(r"\(.*$", "", name)


def load_trace(path):
    g = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
        a = g[k]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3  # us
    return g


def queue_gaps(path, steps):
    """Per hardware queue: kernel time and the idle time between consecutive kernels of that queue (gaps > 200 us are step boundaries /
    host synchronisation and are not counted), per step.  The queue with the most kernel time is the main stream."""
    q = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        q[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    out = []
    for qid, ev in q.items():
        ev.sort()
        busy = sum(e - s for s, e in ev) / 1e6
        gaps = [(ev[i + 1][0] - ev[i][1]) / 1e3 for i in range(len(ev) - 1)]
        small = [g for g in gaps if 0 < g <= 200.0]
        out.append(dict(queue=qid, launches_per_step=len(ev) / steps, busy_ms_per_step=busy / steps, gap_ms_per_step=sum(small) / 1e3 / steps,
                        mean_gap_us=sum(small) / max(len(small), 1)))
    return sorted(out, key=lambda d: -d["busy_ms_per_step"])


def load_pmc(path, counter):
    g = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = g[(short(r["Kernel_Name"]), int(r["Grid_Size"]))]
        a[0] += 1
        a[1] += float(r["Counter_Value"]) * 1024.0
    return g


CLASSES = [  # (class, kernel-name regex, minimum grid size: the vision tower's launches)
    ("gemm_pp", r"^gemm_pp_kernel", 0),
    ("ln_fwd", r"^ln_fwd_kernel<.*false>", 1 << 20),
    ("ln_bwd", r"^ln_bwd_kernel<.*false>", 1 << 20),
    ("attn_fwd", r"^attn_fwd_kernel", 0),
    ("attn_bwd", r"^attn_bwd_sweep_kernel<[^,]*, 7", 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace", required=True)
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--steps", type=int, required=True, help="bench steps (warm-up included) in the trace run")
    ap.add_argument("--pmc-steps", type=int, required=True, help="bench steps (warm-up included) in each PMC run")
    ap.add_argument("--title", default="HBM-side bytes per kernel, joined with the kernel trace")
    ap.add_argument("--note", default="")
    ap.add_argument("--stamp", default="", help="mudpt_amd.build.source_hash() of the library that was profiled")
    ap.add_argument("--out-md", required=True)
    ap.add_argument("--out-json", required=True)
    a = ap.parse_args()
    tr, fe, wr = load_trace(a.trace), load_pmc(a.fetch, "FETCH_SIZE"), load_pmc(a.write, "WRITE_SIZE")
    rows = []
    for k, (n, us) in tr.items():
        f, w = fe.get(k), wr.get(k)
        rd = 2.0 * f[1] / f[0] if f else float("nan")
        wt = w[1] / w[0] if w else float("nan")
        rows.append(dict(kernel=k[0], grid=k[1], per_step=n / a.steps, avg_us=us / n, ms_per_step=us / a.steps / 1e3, read=rd, write=wt))
    rows.sort(key=lambda r: -r["ms_per_step"])
    tot_ms = sum(r["ms_per_step"] for r in rows)
    tot_gb = sum((r["read"] + r["write"]) * r["per_step"] for r in rows if r["read"] == r["read"] and r["write"] == r["write"]) / 1e9
    out = {"steps": a.steps, "kernel_ms_per_step": tot_ms, "hbm_side_gb_per_step": tot_gb, "classes": {}, "queues": queue_gaps(a.trace, a.steps)}
    for cls, pat, min_grid in CLASSES:
        sel = [r for r in rows if re.search(pat, r["kernel"]) and r["grid"] >= min_grid and r["read"] == r["read"]]
        if not sel:
            continue
        n = sum(r["per_step"] for r in sel)
        out["classes"][cls] = {
            "launches_per_step": n, "avg_us": sum(r["avg_us"] * r["per_step"] for r in sel) / n,
            "read_bytes_per_launch": sum(r["read"] * r["per_step"] for r in sel) / n,
            "write_bytes_per_launch": sum(r["write"] * r["per_step"] for r in sel) / n,
        }
        c = out["classes"][cls]
        c["traffic_bytes_per_launch"] = c["read_bytes_per_launch"] + c["write_bytes_per_launch"]
        c["gb_per_s"] = c["traffic_bytes_per_launch"] / c["avg_us"] / 1e3
    with open(a.out_md, "w") as f:
        f.write(f"# {a.title}\n\n{a.note}\n\nTrace: {a.steps} steps; PMC passes: {a.pmc_steps} steps each (separate runs of the same command).  "
                "read = 2 x FETCH_SIZE, write = WRITE_SIZE (bytes that left the L2s; Infinity-Cache hits included).\n\n"
                "| kernel | grid | launches / step | avg us | ms / step | read MB | write MB | GB/s (read + write) |\n|---|---:|---:|---:|---:|---:|---:|---:|\n")
        for r in rows[:32]:
            gbs = (r["read"] + r["write"]) / r["avg_us"] / 1e3
            f.write(f"| `{r['kernel'][:70]}` | {r['grid']} | {r['per_step']:.1f} | {r['avg_us']:.1f} | {r['ms_per_step']:.3f} | {r['read'] / 1e6:.1f} | {r['write'] / 1e6:.1f} | {gbs:.0f} |\n")
        f.write(f"\nAll kernels: {tot_ms:.2f} ms of kernel time per step (both streams), {tot_gb:.1f} GB of HBM-side traffic per step.\n\n"
                "| class (vision-tower launches) | launches / step | avg us | read MB | write MB | GB/s |\n|---|---:|---:|---:|---:|---:|\n")
        for cls, c in out["classes"].items():
            f.write(f"| {cls} | {c['launches_per_step']:.1f} | {c['avg_us']:.1f} | {c['read_bytes_per_launch'] / 1e6:.1f} | {c['write_bytes_per_launch'] / 1e6:.1f} | {c['gb_per_s']:.0f} |\n")
        f.write("\n| hardware queue | launches / step | kernel ms / step | idle between kernels ms / step | mean gap us |\n|---|---:|---:|---:|---:|\n")
        for qd in out["queues"]:
            f.write(f"| {qd['queue']} | {qd['launches_per_step']:.1f} | {qd['busy_ms_per_step']:.2f} | {qd['gap_ms_per_step']:.2f} | {qd['mean_gap_us']:.1f} |\n")
    out["source_hash"] = a.stamp
    json.dump(out, open(a.out_json, "w"), indent=1)
    print(json.dumps(out["classes"], indent=1))


if __name__ == "__main__":
    main()
