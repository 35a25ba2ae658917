import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
thr = 10e6
names = [r[0] for r in c.execute("select name from sqlite_master where type in ('view','table')")]
print([n for n in names if not n.startswith('rocpd_')][:40])
rcols = [r[1] for r in c.execute("pragma table_info(regions)")]
print("regions cols", rcols)
rows = c.execute("select start, end, name from kernels order by start").fetchall()
t0 = rows[0][0]
gaps = []
cur_end = rows[0][1]
for r in rows[1:]:
    if r[0] - cur_end > thr and (cur_end - t0) > 2.0e9:
        gaps.append((cur_end, r[0]))
    cur_end = max(cur_end, r[1])
for a, b in gaps[:6]:
    print(f"--- gap {(b-a)/1e6:.1f} ms at t={(a-t0)/1e6:.1f}")
    for s, e, n, tid in c.execute("select start, end, name, tid from regions where end > ? and start < ? order by start", (a - 2e6, b + 1e6)):
        if e - s > 0.3e6:
            print(f"   api {n:40s} tid {tid} start {(s-a)/1e6:8.2f} ms  dur {(e-s)/1e6:8.2f} ms")
    n_api = c.execute("select count(*) from regions where start > ? and start < ?", (a, b)).fetchone()[0]
    print("   api calls started inside gap:", n_api)
