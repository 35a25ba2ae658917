"""Summarise a rocprofv3 --pmc run (rocpd sqlite) per kernel name: mean counter value per dispatch.
Run ON the GPU box (the raw .db is too large to carry back):
    python scripts/pmc_summary.py /tmp/pmc_X/pmc_results.db > gpurun_out/pmc_X.json
"""
import json
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(pmc_events)")]
out = {"db": db, "pmc_events_columns": cols}
try:
    # view pmc_events carries the kernel/dispatch ids and counter name/value
    q = ("select name, counter_name, count(*), sum(counter_value), avg(counter_value), sum(duration) "
         "from pmc_events group by name, counter_name order by 4 desc")
    rows = c.execute(q).fetchall()
    out["per_kernel"] = [{"kernel": r[0][:90], "counter": r[1], "samples": r[2], "sum": r[3], "mean_per_dispatch": r[4],
                          "sum_duration_ns": r[5]} for r in rows[:240]]
    nd = c.execute("select name, count(*), sum(end-start) from kernels group by name order by 3 desc").fetchall()
    out["dispatches"] = [{"kernel": r[0][:90], "n": r[1], "total_ns": r[2]} for r in nd[:80]]
except Exception as e:  # schema differs: dump what is there
    out["error"] = repr(e)
    out["sample"] = [list(map(str, r)) for r in c.execute("select * from pmc_events limit 5").fetchall()]
print(json.dumps(out, indent=1))
