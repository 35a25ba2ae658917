"""Host-side thread hygiene of the serving loop.

The product's host work is python glue around kernel launches; torch's intra-op (OpenMP) pool is sized from the
*visible* cores (256 on an MI355X host), and its workers spin after every parallel region. Inside a container with a
CFS quota (the GPU boxes run `cpu.max = 1600000 100000`, i.e. 16 CPUs) that spinning exhausts the quota and the kernel
then freezes *every* thread of the cgroup - including the launching thread - for the rest of each 100 ms period:
20-60 ms GPU-idle holes, 100 ms apart (scripts/prof_biggaps.py, scripts/host_threads.py). Capping the pool removes
them (fp16 conv mode 238 -> 334 images/s; the fp32 mode is GPU-bound and hides most of it behind its queue).
"""
from __future__ import annotations

import math
import os

import torch

DEFAULT_HOST_THREADS = 2


def usable_cpus() -> int:
    """cores this process may really use: min(affinity mask, cgroup CFS quota / period)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:                                    # pragma: no cover (non-linux)
        n = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda s: s.split()),):
        try:
            quota, period = parse(open(path).read())
            if quota != "max":
                n = min(n, max(1, math.ceil(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    try:                                                      # cgroup v1
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            n = min(n, max(1, math.ceil(q / p)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def limit_host_threads(n: int | None = None) -> int:
    """cap torch's intra-op pool at `n`; returns the previous setting. Default ($GLASS_HOST_THREADS overrides):
    min(current setting, 2, usable_cpus() / ranks on this node) - it never raises a limit the launcher already set
    (torchrun exports OMP_NUM_THREADS=1) and shares the quota between the node's ranks (LOCAL_WORLD_SIZE).
    Call before the first CPU tensor op where possible (the pool is created lazily at that size)."""
    prev = torch.get_num_threads()
    if n is None:
        n = int(os.environ.get("GLASS_HOST_THREADS", "0"))
        if n <= 0:
            ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
            n = min(prev, DEFAULT_HOST_THREADS, max(1, usable_cpus() // ranks))
    n = max(1, int(n))
    if n != prev:
        torch.set_num_threads(n)
    return prev
