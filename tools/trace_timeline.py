"""Timeline of the last launches from a rocprofv3 --kernel-trace CSV: python tools/trace_timeline.py <kernel_trace.csv> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_step", "k_obs", "k_regen", "k_prep"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sel = rows[-400:-400 + nshow] if len(rows) > 400 else rows[:nshow]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    nm = next(k for k in ("k_step", "k_obs", "k_regen", "k_prep") if k in r["Kernel_Name"])
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-8s start %9.1f end %9.1f dur %7.1f us" % (nm, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
