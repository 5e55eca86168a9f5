"""What the host gives this process: the CPU quota of its cgroup, how long the kernel has throttled it, the cores it has
used.  A real-time process whose threads are runnable but have no CPU is late for reasons no kernel launch explains --
the status record says so (rcf_host_*), next to the pump's own account of its late wake-ups (rcf_pump_stats_t.runq_*)."""
from __future__ import annotations

import os
import time


def cgroup_cpu():
    """{'quota_cores': float | None, 'nr_throttled': int, 'throttled_ms': float, 'usage_s': float} of this process's CPU
    cgroup (v2, then v1); fields that cannot be read are absent"""
    out = {}
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat", "/sys/fs/cgroup/cpu,cpuacct/cpu.stat"):
        try:
            raw = {}
            with open(path) as fh:
                for line in fh:
                    k, v = line.split()
                    raw[k] = int(v)
        except Exception:
            continue
        if "nr_throttled" in raw:
            out["nr_throttled"] = raw["nr_throttled"]
        if "throttled_usec" in raw:
            out["throttled_ms"] = raw["throttled_usec"] / 1e3
        elif "throttled_time" in raw:
            out["throttled_ms"] = raw["throttled_time"] / 1e6
        if "usage_usec" in raw:
            out["usage_s"] = raw["usage_usec"] / 1e6
        break
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()
        out["quota_cores"] = None if q == "max" else int(q) / int(per)
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
                q = int(fh.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                per = int(fh.read())
            out["quota_cores"] = None if q < 0 else q / per
        except Exception:
            pass
    return out


class CpuMeter:
    """cores used by this process / by its whole cgroup since the previous call"""

    def __init__(self):
        self._t = time.time()
        self._p = time.process_time()
        self._c = cgroup_cpu()

    def sample(self):
        now, p, c = time.time(), time.process_time(), cgroup_cpu()
        dt = max(now - self._t, 1e-9)
        out = {"rcf_host_process_cores": (p - self._p) / dt}
        if "usage_s" in c and "usage_s" in self._c:
            out["rcf_host_cgroup_cores"] = (c["usage_s"] - self._c["usage_s"]) / dt
        if c.get("quota_cores") is not None:
            out["rcf_host_cpu_quota_cores"] = c["quota_cores"]
        if "throttled_ms" in c:
            out["rcf_host_throttled_ms"] = c["throttled_ms"]
            out["rcf_host_nr_throttled"] = c.get("nr_throttled", 0)
        out["rcf_host_cpus_allowed"] = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
        self._t, self._p, self._c = now, p, c
        return out
