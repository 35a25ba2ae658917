"""Package power and shader clock during bench.py's step loop with the 1x1 layers on the fp32 MFMA (GLASS_PW_SPLIT=0) and on the
bf16-split kernel (9): the sampler of scripts/power_clock_log.py around a child bench.py.   python scripts/exp_split_bench_power.py"""
import os, sys
sys.argv = [sys.argv[0], "3"]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import io, contextlib
import power_clock_log as P   # noqa: E402

for sp in ("0", "9", "0", "9"):
    os.environ["GLASS_PW_SPLIT"] = sp
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        P.leg_bench(f"bench step loop, GLASS_PW_SPLIT={sp}", ["--steps", "400", "--warmup", "3", "--no-extras", "--no-cpu-baseline"])
    rows = []
    val = None
    for ln in buf.getvalue().splitlines():
        if ln.startswith("=="):
            import re
            m = re.search(r'"value": ([\d.]+)', ln)
            val = float(m.group(1)) if m else None
        parts = ln.split()
        if len(parts) == 3:
            try:
                rows.append(tuple(float(x) for x in parts))
            except ValueError:
                pass
    hot = [r for r in rows if r[1] > 900.0]           # the timed loop: the stretch above 900 W
    if hot:
        print(f"GLASS_PW_SPLIT={sp}: value {val:.1f} images/s; {len(hot)} samples above 900 W: power mean {sum(r[1] for r in hot) / len(hot):.0f} W (max {max(r[1] for r in hot):.0f}), "
              f"sclk mean {sum(r[2] for r in hot) / len(hot):.0f} MHz (min {min(r[2] for r in hot):.0f})")
    else:
        print(f"GLASS_PW_SPLIT={sp}: value {val}; no samples above 900 W of {len(rows)}")
