"""Print the last N rows of a rocprofv3 --kernel-trace CSV as (start offset us, duration us, gap to the previous kernel
us, name): python tools/trace_tail.py <dir> [N]"""
import csv, glob, sys
paths = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 160
rows = []
for p in paths:
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f}  {r['Kernel_Name'][:90]}")
    prev_end = e
