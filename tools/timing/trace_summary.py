"""Per-kernel summary of a rocprofv3 kernel trace, optionally only the last `--tail N` dispatches (one steady-state repetition):
    python tools/timing/trace_summary.py <..._kernel_trace.csv> [--tail N]
prints dispatches in time order with start offsets, so that gaps between dependent launches show."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
if "--tail" in sys.argv:
    rows = rows[-int(sys.argv[sys.argv.index("--tail") + 1]):]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    nm = r["Kernel_Name"].replace("mzk::", "").replace("void ", "").split("(")[0]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +gap %6.1f  dur %8.1f us  grid %8s x %-5s wg %-5s %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Grid_Size_X"], r.get("Grid_Size_Y", ""), r["Workgroup_Size_X"], nm[:90]))
    prev_end = e
print("span %.1f us" % ((prev_end - t0) / 1e3))
