"""List the largest GPU-idle gaps inside one steady-state bench step of a rocprofv3 kernel trace (rocpd sqlite).

    python scripts/prof_gaps.py /tmp/pr/r_results.db [step_index]
"""
import sqlite3
import sys

db = sys.argv[1]
step = int(sys.argv[2]) if len(sys.argv) > 2 else 4
c = sqlite3.connect(db)
ends = [r[0] for r in c.execute("select end from kernels where name like 'postprocess_words_kernel%' order by end")]
a, b = ends[step - 1], ends[step]
rows = c.execute("select start, end, name from kernels where start >= ? and end <= ? order by start", (a, b)).fetchall()
gaps = []
cur_end, cur_name = a, "postprocess_words_kernel (previous step)"
for s, e, n in rows:
    if s > cur_end:
        gaps.append((s - cur_end, cur_name, n, (cur_end - a) / 1e6))
    if e > cur_end:
        cur_end, cur_name = e, n
tot = sum(g[0] for g in gaps)
print(f"step {step}: wall {(b - a) / 1e6:.2f} ms, {len(rows)} kernels, idle {tot / 1e6:.2f} ms in {len(gaps)} gaps")
for g, before, after, t in sorted(gaps, reverse=True)[:25]:
    print(f"{g / 1e3:8.1f} us at t={t:6.2f} ms  after {before[:48]:48s} before {after[:48]}")
# idle per phase: bucket by the kernel that follows
import collections
by = collections.Counter()
for g, before, after, t in gaps:
    by[after[:40]] += g
print("idle by following kernel:")
for k, v in by.most_common(12):
    print(f"  {v / 1e3:8.1f} us  {k}")
