"""kernel timeline out of a rocprofv3 results .db: python tools/timeline.py db [first_index] [count]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(cur.execute(f"select k.kernel_name, d.start, d.end, d.queue_id from {kd} d join {ks} k on d.kernel_id=k.id order by d.start"))


def short(n):
    m = re.search(r"_ZN6hiprec\d+(\w+?)(I|E)", n)
    return (m.group(1) + ("<flush>" if "Lb1E" in n else "")) if m else n[:30]


t0 = rows[0][1]
seq = [(short(n), (s - t0) / 1e3, (e - s) / 1e3, q) for n, s, e, q in rows]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 60
if first < 0:
    idx = [i for i, x in enumerate(seq) if x[0].startswith("stage_epoch")]
    print("stage launches:", len(idx), [round(seq[i][2], 1) for i in idx])
    first = idx[-first] - 3
prev_end = None
for x in seq[first:first + cnt]:
    gap = "" if prev_end is None else f" gap {x[1] - prev_end:6.1f}"
    print("%-30s start %10.1f us dur %7.1f q%s" % x + gap)
    if x[3] == seq[first][3] or prev_end is None:
        prev_end = x[1] + x[2]
