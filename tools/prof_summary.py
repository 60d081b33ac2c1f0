"""Turn a rocprofv3 --kernel-trace --stats kernel_stats.csv into the markdown summary committed under profiles/."""
import csv
import sys

src, dst, title = sys.argv[1], sys.argv[2], sys.argv[3]
note = sys.argv[4] if len(sys.argv) > 4 else ""
rows = list(csv.DictReader(open(src)))
with open(dst, "w") as f:
    f.write(f"# {title}\n\n{note}\n\n| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
    for r in rows[:30]:
        f.write(f"| `{r['Name'][:110]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |\n")
    g = [r for r in rows if "gemm" in r["Name"] and "sgemm" not in r["Name"]]
    if g:
        tot, n = sum(float(r["TotalDurationNs"]) for r in g), sum(int(r["Calls"]) for r in g)
        f.write(f"\nAll MFMA GEMM kernel instantiations: {n} launches, {tot / 1e6:.2f} ms, average {tot / n / 1e3:.1f} us per launch.\n")
    pp = [r for r in rows if "gemm_pp_kernel" in r["Name"]]
    if pp:
        tot, n = sum(float(r["TotalDurationNs"]) for r in pp), sum(int(r["Calls"]) for r in pp)
        f.write(f"\ngemm_pp_kernel (the kernel of bench.py's roofline object), all epilogues: {n} launches, {tot / 1e6:.2f} ms, average {tot / n / 1e3:.1f} us per launch.\n")
    f.write(f"\nAll kernels: {sum(float(r['TotalDurationNs']) for r in rows) / 1e6:.2f} ms.\n")
