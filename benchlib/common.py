"""Constants and small helpers every leg of bench.py shares."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_PY = os.path.join(ROOT, "bench.py")

FS = 20e6
NB = 256
N_ACTIVE = 32
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
FP32_MATRIX_PEAK_TF = 157.3    # MI355X_MICROARCH.md: FP32 matrix = FP32 vector peak
SCAN_CEILING_NOTE = ("a 1M-point FFT cannot live in LDS: the four-step form moves 8+8 B (columns) + 8+4 B (rows) "
                     "+ 4+4 B (running sum) = 36 B/sample against 12 B algorithmic, at the 5.5 TB/s a plain copy "
                     "sustains on this chip: ceiling 12/36 x 5.5/8 = 0.23 of the HBM peak (DESIGN 4.4)")


def proto_taps(native, fs=FS, nb=NB):
    # SURVEY 8(d) cfg2 prototype by the reference's own low_pass_2 rule: fc = 0.4 bin, tw = 0.2 bin,
    # 60 dB, Blackman-Harris -> 3491 taps (13.6 per branch) at 256 bins, 6981 at 512
    bw = fs / nb
    return native.design_low_pass_2(1.0, fs, 0.4 * bw, 0.2 * bw, 60.0, native.WIN_BLACKMAN_HARRIS)


def read_sclk_mhz():
    """current shader clock from sysfs (pp_dpm_sclk marks the active level with '*').  A box exposes every GPU of the
    node there, idle ones included, and nothing maps a HIP device to its card index without the PCI bus id: the busy
    GPU is the one with the highest current clock, so report the maximum.  None when nothing is exposed."""
    import glob
    best = None
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            for line in open(f):
                if "*" in line:
                    v = float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
                    best = v if best is None else max(best, v)
        except Exception:
            continue
    return best


def cgroup_cpu_stat():
    """(nr_throttled, throttled_usec, usage_usec, quota cores or None) of this process's CPU cgroup: a paced run on a host
    whose container is throttled by its CFS quota misses deadlines that are not the GPU's"""
    out = {}
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            for line in open(path):
                k, v = line.split()
                out[k] = int(v)
            break
        except Exception:
            continue
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else int(q) / int(per)
    except Exception:
        pass
    return out.get("nr_throttled"), out.get("throttled_usec", out.get("throttled_time")), out.get("usage_usec"), quota


def cpu_busy_sample(seconds=0.3):
    """{cpu: busy fraction over `seconds`} of the CPUs this process may run on, from /proc/stat: EVERYONE's load on them (the
    GPU boxes are shared; the container has a CPU quota but no CPUs of its own)"""
    def snap():
        out = {}
        try:
            for line in open("/proc/stat"):
                if line.startswith("cpu") and line[3].isdigit():
                    f = line.split()
                    v = [int(x) for x in f[1:9]]
                    out[int(f[0][3:])] = (v[3] + v[4], sum(v))
        except Exception:
            pass
        return out
    a = snap()
    time.sleep(seconds)
    b = snap()
    busy = {}
    for c in os.sched_getaffinity(0):
        if c in a and c in b and b[c][1] > a[c][1]:
            busy[c] = 1.0 - (b[c][0] - a[c][0]) / float(b[c][1] - a[c][1])
    return busy


def idlest_cpus(n, busy=None):
    """n CPUs of the affinity mask for latency-critical threads: the least busy ones, at most one per physical core (a
    sibling hardware thread that is busy is someone else's work on the same core).  [] when /proc/stat says nothing."""
    busy = cpu_busy_sample() if busy is None else busy
    if not busy:
        return []
    core_of = {}
    for c in busy:
        try:
            core_of[c] = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip()
        except OSError:
            core_of[c] = str(c)
    load_of_core = {}
    for c, b in busy.items():
        load_of_core[core_of[c]] = load_of_core.get(core_of[c], 0.0) + b
    picked, seen = [], set()
    for c in sorted(busy, key=lambda c_: (load_of_core[core_of[c_]], busy[c_], c_)):
        if core_of[c] not in seen:
            seen.add(core_of[c])
            picked.append(c)
        if len(picked) == n:
            break
    return picked
