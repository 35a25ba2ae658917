"""Per-SHAPE kernel durations out of a rocprofv3 kernel trace (rocpd sqlite): launches of the kernels whose name contains
`pattern`, grouped by grid size (= workgroups x 256 threads, i.e. by layer shape).

    python scripts/prof_by_grid.py /tmp/pr_default/r_results.db conv3x3_wino43_f32 "label" > gpurun_out/wino43_by_shape_default.txt

Used to compare a layer's duration INSIDE the step (two steps in flight, two-stream local extractor: its partial last round of
workgroups can be filled by other kernels) with its duration on the serial meter (VERDICT r3 #5c)."""
import sqlite3
import sys

db, pat = sys.argv[1], sys.argv[2]
label = sys.argv[3] if len(sys.argv) > 3 else ""
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
gx = next((k for k in cols if k.lower() in ("grid_x", "grid_size_x", "grid_size")), None) or \
    next((k for k in cols if "grid" in k.lower() and k.lower().endswith("x")), None) or next((k for k in cols if "grid" in k.lower()), None)
wx = next((k for k in cols if k.lower() in ("workgroup_x", "workgroup_size_x", "workgroup_size")), None) or \
    next((k for k in cols if "workgroup" in k.lower() and k.lower().endswith("x")), None)
print(f"# {label}")
print(f"# source db: {db}; kernels matching '{pat}', grouped by ({gx}" + (f" / {wx}" if wx else "") + ")")
if gx is None:
    print("# no grid column in the kernels view: columns = " + ", ".join(cols))
    sys.exit(0)
sel = f"{gx}" + (f", {wx}" if wx else ", 256")
rows = c.execute(f"select name, {sel}, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels "
                 f"where name like ? group by name, {gx} order by 8 desc", (f"%{pat}%",)).fetchall()
print(f"{'kernel':44s} {'workgroups':>10s} {'calls':>6s} {'avg us':>9s} {'min us':>9s} {'max us':>9s} {'total ms':>9s}")
for name, g, w, n, avg, mn, mx, tot in rows:
    short = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:44]
    print(f"{short:44s} {int(g) // max(int(w), 1):10d} {n:6d} {avg / 1e3:9.1f} {mn / 1e3:9.1f} {mx / 1e3:9.1f} {tot / 1e6:9.2f}")
