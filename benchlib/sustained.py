"""The `sustained` leg: >= 2 s of back-to-back commits, the filterbank launch timed per window."""
import time

from .common import HBM_PEAK_GBS, read_sclk_mhz

def sustained_leg(fe, native, B, alg_bytes, seconds=2.0, window=100):
    """>= `seconds` of back-to-back commits of the resident block, the filterbank launch timed with HIP events on
    librcf's stream and read back every `window` launches (one stream sync per window: < 0.5 % of the time).
    The fraction that counts as sustained is the LAST window's."""
    fe.sync()
    fe.timing_enable(True, classes=[native.T_PFB])
    fe.timing_read(native.T_PFB)
    wins, clocks = [], []
    t0 = time.perf_counter()
    while True:
        for _ in range(window):
            fe.commit(B)
        clocks.append(read_sclk_mhz())                 # the GPU is still busy: the host runs <= 2 commits ahead
        ms, n = fe.timing_read(native.T_PFB)
        wins.append(ms / max(n, 1))
        if time.perf_counter() - t0 >= seconds and len(wins) >= 3:
            break
    wall = time.perf_counter() - t0
    fe.timing_enable(False)
    frac = lambda ms: alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    ck = [c for c in clocks if c]
    return {
        "seconds": wall, "launches": len(wins) * window, "window": window,
        "kernel_us_first_window": wins[0] * 1e3, "kernel_us_last_window": wins[-1] * 1e3,
        "kernel_us_slowest_window": max(wins) * 1e3, "kernel_us_fastest_window": min(wins) * 1e3,
        "frac_first_window": frac(wins[0]), "frac_last_window": frac(wins[-1]), "frac_slowest_window": frac(max(wins)),
        "wall_ms_per_step": wall / (len(wins) * window) * 1e3,
        "sclk_mhz_first": ck[0] if ck else None, "sclk_mhz_last": ck[-1] if ck else None,
        "kernel_us_by_window": [round(w * 1e3, 2) for w in wins[:: max(1, len(wins) // 32)]],
        "note": "HIP events on every filterbank launch, read back per window of %d launches; sclk from "
                "/sys/class/drm/card*/device/pp_dpm_sclk while the queue is full" % window,
    }
