"""Summarise a rocprofv3 --pmc pass of SQ / GRBM counters per kernel (averages per launch).

    python tools/pmc_sq_summary.py <counter_collection.csv> <out.md>
Derived columns: wave-state fractions (SQ_WAIT_ANY = parked at s_waitcnt / s_barrier, SQ_WAIT_INST_ANY = issue stall on a dependency or a
busy pipe, SQ_ACTIVE_INST_ANY = issuing; fractions of SQ_WAVE_CYCLES); MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES (matrix-pipe busy
cycles summed over the SIMDs: 16 per v_mfma_f32_16x16x32, checked against the launches' MFMA counts) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs),
i.e. the fraction of the launch's shader cycles in which a SIMD's matrix pipe is busy."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
seen = set()
for r in rows:
    k = r["Kernel_Name"]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"])
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = ["# SQ / GRBM counters per kernel (rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
       "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE)", "",
       "Command: `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile` (B = 256, bf16).  Averages per launch; profiled passes run "
       "at a lower clock than unprofiled ones (MI355X_MICROARCH.md, DVFS item 2).", "",
       "| kernel | launches | avg us | parked (WAIT_ANY) | issue stall (WAIT_INST_ANY) | issuing (ACTIVE_INST_ANY) | MFMA pipe utilisation |",
       "|---|---:|---:|---:|---:|---:|---:|"]
order = sorted(agg, key=lambda k: -sum(dur[k]))
for k in order[:16]:
    m = {c: sum(v) / len(v) for c, v in agg[k].items()}
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    us = sum(dur[k]) / len(dur[k])
    mf = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * m["GRBM_GUI_ACTIVE"] / 8) if m.get("GRBM_GUI_ACTIVE") else 0
    out.append(f"| `{k[:90]}` | {len(dur[k])} | {us:.1f} | {m.get('SQ_WAIT_ANY', 0) / wc:.2f} | {m.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} | "
               f"{m.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f} | {mf:.2f} |")
# optional second table: LDS counters of a separate pass (python tools/pmc_sq_summary.py <sq.csv> <out.md> <lds.csv>):
# SQ_ACTIVE_INST_LDS / SQ_WAVE_CYCLES = share of wave time spent issuing LDS instructions; SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = share
# of the LDS pipe's active cycles lost to bank conflicts; SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES = issue stalls on the LDS queue
if len(sys.argv) > 3:
    rows2 = list(csv.DictReader(open(sys.argv[3])))
    agg2 = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows2:
        agg2[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out += ["", "LDS counters (separate pass: --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT):", "",
            "| kernel | LDS instructions per launch | issuing LDS (ACTIVE_INST_LDS / WAVE_CYCLES) | stalled on the LDS queue (WAIT_INST_LDS / WAVE_CYCLES) | LDS pipe active (LDS_IDX_ACTIVE / BUSY_CYCLES-equivalent) | bank-conflict share of LDS-active cycles |",
            "|---|---:|---:|---:|---:|---:|"]
    for k in order[:16]:
        if k not in agg2:
            continue
        m = {c: sum(v) / len(v) for c, v in agg2[k].items()}
        wc = m.get("SQ_WAVE_CYCLES", 0) or 1
        act = m.get("SQ_LDS_IDX_ACTIVE", 0) or 1
        out.append(f"| `{k[:90]}` | {m.get('SQ_INSTS_LDS', 0):.0f} | {m.get('SQ_ACTIVE_INST_LDS', 0) / wc:.2f} | {m.get('SQ_WAIT_INST_LDS', 0) / wc:.2f} | "
                   f"{m.get('SQ_LDS_IDX_ACTIVE', 0):.3g} | {m.get('SQ_LDS_BANK_CONFLICT', 0) / act:.2f} |")
open(sys.argv[2], "w").write("\n".join(out) + "\n")
print("\n".join(out[5:]))
