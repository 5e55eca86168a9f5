"""roofline.traffic measured live: two rocprofv3 --pmc child passes over a short headline-only run of bench.py."""
import os
import sys

from .common import BENCH_PY

def measure_traffic_live(cfg5, block, kernel_substr, timeout_s=150):
    """HBM bytes per filterbank launch from the PMC counters, measured in THIS run on THIS box: two separate rocprofv3
    passes (FETCH_SIZE, then WRITE_SIZE -- they do not fit one pass: MI355X_MICROARCH.md) over a short headline-only child
    run of this script, counters averaged per dispatch of the kernel; KiB -> bytes, FETCH x 2 on gfx950 (same guide).
    Counter passes carry --kernel-trace only.  None when rocprofv3 is not there or a pass fails (the line then falls back
    to the dated file under profiles/ and says so)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rp is None:
        return None
    if any(k.startswith("ROCPROF") for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None                                          # this run is itself being profiled: no profiler inside a profiler
    out = {}
    child = [sys.executable, BENCH_PY, "--steps", "5", "--warmup", "1", "--block", str(block), "--no-extras",
             "--no-cpu-baseline", "--no-sustained", "--no-live-traffic", "--prewarm-seconds", "0.2"]
    if cfg5:
        child += ["--config", "cfg5"]
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="rcf_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp", RCF_BENCH_FULL="/dev/null")
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                env.pop(k, None)
            # (its own session: a pass that hangs -- it happened once in a profile run, ten minutes of nothing -- is killed
            # WITH the child run rocprofv3 started, so that no stray copy of this script shares the GPU with the legs below)
            pr = subprocess.Popen([rp, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--"] + child,
                                  cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                pr.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except OSError:
                    pass
                pr.wait()
                return None
            r = pr
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if kernel_substr in row["Kernel_Name"] and "true>" not in row["Kernel_Name"] and row["Counter_Name"] == counter:
                        vals.append(float(row["Counter_Value"]))
            if r.returncode != 0 or len(vals) < 3:
                return None
            vals = vals[1:]                                   # the first dispatch still sees zero history / cold caches
            out[counter] = (sum(vals) / len(vals), len(vals))
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch = out["FETCH_SIZE"][0] * 1024.0 * 2.0
    write = out["WRITE_SIZE"][0] * 1024.0
    return {"hbm_bytes_per_launch": fetch + write, "fetch_bytes_corrected_x2": fetch, "write_bytes": write,
            "dispatches_averaged": min(out["FETCH_SIZE"][1], out["WRITE_SIZE"][1])}
