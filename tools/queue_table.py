"""Per hardware queue: kernel time per step by (kernel, grid), from a rocprofv3 kernel trace.  The queue with the most kernel time is the main
stream (the vision tower and everything serial with it); the other is the text tower's side stream.
    python tools/queue_table.py <kernel_trace.csv> <steps in the trace> [rows per queue]"""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name).replace("mudpt::", "")
    return re.sub(r"\(.*$", "", name)


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    q = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(path)):
        k = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
        a = q[r["Queue_Id"]][k]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for qid, tab in sorted(q.items(), key=lambda kv: -sum(v[1] for v in kv[1].values())):
        tot = sum(v[1] for v in tab.values()) / steps / 1e3
        print(f"\n## queue {qid}: {tot:.2f} ms of kernel time per step, {sum(v[0] for v in tab.values()) / steps:.1f} launches per step\n")
        print("| kernel | grid | launches / step | avg us | ms / step |\n|---|---:|---:|---:|---:|")
        for (name, grid), (n, us) in sorted(tab.items(), key=lambda kv: -kv[1][1])[:top]:
            print(f"| `{name[:80]}` | {grid} | {n / steps:.1f} | {us / n:.1f} | {us / steps / 1e3:.3f} |")


if __name__ == "__main__":
    main()
