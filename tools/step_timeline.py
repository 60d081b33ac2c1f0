"""Timeline of ONE step on the busiest hardware queue, from a rocprofv3 kernel trace: per launch its duration and the idle gap before it.
    python tools/step_timeline.py <kernel_trace.csv> <steps in the trace> [step index]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name).replace("mudpt::", "")
    return re.sub(r"\(.*$", "", name)[:70]


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    which = int(sys.argv[3]) if len(sys.argv) > 3 else steps - 2
    rows = list(csv.DictReader(open(path)))
    byq = {}
    for r in rows:
        byq.setdefault(r["Queue_Id"], []).append(r)
    main_q = max(byq.values(), key=lambda v: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in v))
    main_q.sort(key=lambda r: int(r["Start_Timestamp"]))
    per = len(main_q) // steps
    seg = main_q[which * per:(which + 1) * per]
    prev = None
    busy = gap = 0.0
    print("| # | kernel | grid | us | gap before us |\n|---:|---|---:|---:|---:|")
    for i, r in enumerate(seg):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        g = (s - prev) / 1e3 if prev is not None else 0.0
        busy += (e - s) / 1e3
        gap += max(g, 0.0)
        print(f"| {i} | `{short(r['Kernel_Name'])}` | {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))} | {(e - s) / 1e3:.1f} | {g:.1f} |")
        prev = e
    print(f"\n{len(seg)} launches, {busy / 1e3:.3f} ms in kernels, {gap / 1e3:.3f} ms between them")


if __name__ == "__main__":
    main()
