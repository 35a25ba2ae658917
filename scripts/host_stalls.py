"""Where does the host stall?  Wraps every libglass_hip entry point with a timer and runs bench.py's main():
prints the C calls (and the python gaps between consecutive C calls) that took longer than THRESH ms.
usage: python scripts/host_stalls.py [bench.py args]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                              # noqa: E402  (sets sys.path for glass_amd)
from glass_amd import _lib                                                # noqa: E402

THRESH = float(os.environ.get("STALL_MS", "3"))
real = _lib.lib()
state = {"last_end": time.perf_counter(), "last_name": "-", "n": 0}
log = []


class Proxy:
    def __getattr__(self, name):
        fn = getattr(real, name)

        def timed(*a):
            t0 = time.perf_counter()
            gap = (t0 - state["last_end"]) * 1e3
            if gap > THRESH:
                log.append((state["n"], "python gap", gap, f"{state['last_name']} -> {name}"))
            r = fn(*a)
            t1 = time.perf_counter()
            if (t1 - t0) * 1e3 > THRESH:
                log.append((state["n"], "C call", (t1 - t0) * 1e3, name))
            state["last_end"] = t1
            state["last_name"] = name
            state["n"] += 1
            return r
        timed.restype = getattr(fn, "restype", None)
        setattr(self, name, timed)
        return timed


_lib._LIB = Proxy()
bench.main()
for n, kind, ms, what in log:
    print(f"call#{n:6d}  {kind:10s} {ms:8.2f} ms  {what}", file=sys.stderr)
