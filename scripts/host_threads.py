"""Which host threads burn CPU during a bench run?  Runs bench.main() and prints per-thread user+system seconds
(/proc/self/task/*/stat) and the cgroup throttling counters.   python scripts/host_threads.py [bench.py args]"""
import glob
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                              # noqa: E402


def cgstat():
    try:
        return dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
    except OSError:
        return {}


c0, t0 = cgstat(), time.time()
bench.main()
c1 = cgstat()
tick = os.sysconf("SC_CLK_TCK")
rows = []
for p in glob.glob("/proc/self/task/*/stat"):
    try:
        s = open(p).read()
    except OSError:
        continue
    comm = s[s.index("(") + 1:s.rindex(")")]
    f = s[s.rindex(")") + 2:].split()
    rows.append(((int(f[11]) + int(f[12])) / tick, comm, p.split("/")[4]))
rows.sort(reverse=True)
print(f"wall {time.time() - t0:.1f} s, {len(rows)} live threads, cpu.max {open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else '-'}", file=sys.stderr)
for cpu, comm, tid in rows[:12]:
    print(f"  {cpu:8.2f} s  {comm:20s} tid {tid}", file=sys.stderr)
print(f"  ... sum of all live threads {sum(r[0] for r in rows):.1f} s", file=sys.stderr)
if c0:
    print("  cgroup: " + ", ".join(f"{k} +{int(c1[k]) - int(c0[k])}" for k in ("usage_usec", "nr_periods", "nr_throttled", "throttled_usec")),
          file=sys.stderr)
