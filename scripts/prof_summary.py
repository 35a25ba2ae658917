"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the text table kept under profiles/.

    python scripts/prof_summary.py gpurun_out/prof_r1/r1_results.db [steps] > profiles/r01_kernel_stats.txt
"""
import sqlite3
import sys

db = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
c = sqlite3.connect(db)
n_pp = c.execute("select count(*) from kernels where name like 'postprocess_words_kernel%'").fetchone()[0]
if steps <= 0:
    steps = max(n_pp, 1)            # every bench step ends with one postprocess_words_kernel launch
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                 "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
# GPU-busy (union of kernel intervals) vs wall per bench step; a step ends with postprocess_words_kernel
iv = c.execute("select start, end from kernels order by start").fetchall()
ends = [r[0] for r in c.execute("select end from kernels where name like 'postprocess_words_kernel%' order by end")]
step_lines = []
for a, b in zip(ends[:-1], ends[1:]):
    busy, cur_s, cur_e = 0, None, None
    for s_, e_ in iv:
        if e_ <= a or s_ >= b:
            continue
        s_, e_ = max(s_, a), min(e_, b)
        if cur_e is None or s_ > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s_, e_
        else:
            cur_e = max(cur_e, e_)
    if cur_e is not None:
        busy += cur_e - cur_s
    step_lines.append(f"{(b - a) / 1e6:.2f}/{busy / 1e6:.2f}")
label = sys.argv[3] if len(sys.argv) > 3 else "python bench.py --steps 5 --warmup 2 --no-cpu-baseline"
print(f"# rocprofv3 --kernel-trace --stats summary of: {label}")
print(f"# source db: {db}; {steps} bench steps in the trace (pipeline fill + warmup + timed + 3 metered; = postprocess_words_kernel launches)")
print(f"# total kernel time {tot / 1e6:.2f} ms  ({tot / 1e6 / steps:.2f} ms per step)")
print(f"# per step wall/GPU-busy ms (union of kernel intervals between consecutive postprocess_words ends): {' '.join(step_lines)}")
print(f"{'Name':72s} {'Calls':>7s} {'TotalDurationNs':>16s} {'AverageNs':>12s} {'MinNs':>10s} {'MaxNs':>10s} {'Percentage':>10s}")
for name, n, s, a, mn, mx in rows[:60]:
    print(f"{name[:72]:72s} {n:7d} {s:16d} {a:12.1f} {mn:10d} {mx:10d} {100.0 * s / tot:10.2f}")
