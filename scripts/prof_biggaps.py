"""All GPU-idle gaps above a threshold over a whole rocprofv3 kernel trace (rocpd sqlite), with the kernels either
side, the dispatch index and the queue ids.   python scripts/prof_biggaps.py results.db [min_ms]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
thr = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 5e6
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else "0")
rows = c.execute(f"select start, end, name, {q} from kernels order by start").fetchall()
t0 = rows[0][0]
cur_end, cur = rows[0][1], rows[0]
for i, r in enumerate(rows[1:], 1):
    s, e = r[0], r[1]
    if s - cur_end > thr:
        print(f"gap {(s - cur_end) / 1e6:7.2f} ms at t={(cur_end - t0) / 1e6:9.2f} ms  dispatch #{i}  q{cur[3]}->q{r[3]}  "
              f"after {cur[2][:34]} | before {r[2][:34]}")
    if e > cur_end:
        cur_end, cur = e, r
print("columns:", cols)
