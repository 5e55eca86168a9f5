"""The LAST stdout line of bench.py: a compact object (<= LIMIT bytes) with the contract keys, the roofline and
cpu_baseline objects and one-number summaries of every leg.  The full record (every point of every leg) goes to
bench_full.json -- the driver keeps 8 018 bytes of stdout, and round 5's 27.9 KB line came back unparsed."""
import json
import math
import os

from .common import ROOT

LIMIT = 7900          # bytes of the line incl. newline; the driver's stdout tail is 8 018


def sig(x, n=5):
    """floats to n significant digits (the contract's own numbers stay exact: see compact())"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        if x == 0 or not math.isfinite(x):
            return x if math.isfinite(x) else None
        return float("%.*g" % (n, x))
    if isinstance(x, dict):
        return {k: sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [sig(v, n) for v in x]
    return x


def pick(d, *keys, **renamed):
    """the named keys of d that exist (renamed: new_name=old_name)"""
    if not isinstance(d, dict):
        return d
    out = {k: d[k] for k in keys if k in d}
    out.update({new: d[old] for new, old in renamed.items() if old in d})
    return out


def _err(d):
    return {"error": str(d["error"])[:160]} if isinstance(d, dict) and "error" in d else None


def _realtime(rt):
    if not isinstance(rt, dict):
        return None
    if "error" in rt:
        return _err(rt)
    out = pick(rt, "pump_threads", "staggered", block_ms="block_ms")
    out["confirm_seconds"] = rt.get("seconds_of_the_confirmation_run_at_K_max")
    for shape in ("pfb256", "grid1600", "grid1600fm"):
        s = rt.get(shape)
        if not isinstance(s, dict):
            continue
        e = pick(s, "K_max", "K_max_first_attempt", "first_K_that_missed", "K_max_p99_under_5ms", "bins_per_front_end",
                 "demodulated_per_front_end", "fm_channels_sustained", "input_Msps_sustained")
        a = s.get("at_K_max")
        if isinstance(a, dict):
            e["at_K_max"] = pick(a, "seconds", "deadline_misses", "ring_overruns", "latency_ms_p50", "latency_ms_p99",
                                 "latency_ms_max", "gpu_busy_percent_est", "pcie_GBps_in", "confirmation_run")
            wl = a.get("why_late") or {}
            if wl:
                e["at_K_max"]["late_wakeups_ms"] = wl.get("late_wakeups_ms")
                e["at_K_max"]["of_them_on_a_run_queue_ms"] = wl.get("of_them_on_a_run_queue_ms")
            cg = a.get("host_cgroup") or {}
            e["at_K_max"].update(pick(cg, "throttled_ms", "cpu_cores_used_mean"))
            if cg.get("cpu_quota_cores") is not None:
                out["cpu_quota_cores"] = cg["cpu_quota_cores"]
        e["points"] = [[p.get("front_ends"), bool(p.get("ok")), p.get("deadline_misses"), p.get("latency_ms_p99")]
                       for p in s.get("points", [])]
        out["points_are"] = "[K, ok, misses, p99 ms] in run order"
        out[shape] = e
    return out


def _channels(ch):
    out = pick(ch, *[k for k in ch if not isinstance(ch[k], dict)])
    db = ch.get("direct_bank")
    if isinstance(db, dict):
        if "error" in db:
            out["direct_bank"] = _err(db)
        else:
            rt = [p for p in db.get("points", []) if p.get("real_time")]
            top = rt[-1] if rt else (db.get("points") or [{}])[-1]
            out["direct_bank"] = {"channels_run_in_real_time": db.get("channels_run_in_real_time"),
                                  "at_that_count": pick(top, "channels", "kernel_ms_per_block", "wall_ms_per_block",
                                                        "block_ms_of_signal", "tflops_fp32", "frac_of_fp32_matrix_peak",
                                                        "kernel"),
                                  "counts_run": [p.get("channels") for p in db.get("points", [])]}
    rg = ch.get("reference_grid_filterbank")
    if isinstance(rg, dict):
        if "error" in rg:
            out["reference_grid_filterbank"] = _err(rg)
        elif isinstance(rg.get("with_taps"), list):              # (already a compact line's summary: as it is)
            out["reference_grid_filterbank"] = rg
        else:
            e = pick(rg, "kernel", "pfb_ms_per_block", "reference_channels_per_frontend")
            e["frac"] = (rg.get("roofline") or {}).get("frac")
            e["sustained_frac_last_window"] = (rg.get("sustained") or {}).get("frac_last_window")
            e["with_taps"] = [[p.get("bins_tapped"), bool(p.get("discriminator_only")), p.get("pfb_ms_per_block"),
                               p.get("tap_finalize_ms_per_block")] for p in (rg.get("with_taps") or {}).get("points", [])]
            e["with_taps_are"] = "[bins tapped, fm only, bank ms, tap_finalize ms] per block"
            e["fused_discriminator"] = [[p.get("mode"), p.get("pfb_ms_per_block"), p.get("frac_of_hbm_peak")] if "error" not in p
                                        else [p.get("mode"), str(p["error"])[:80]]
                                        for p in (rg.get("fused_discriminator") or {}).get("points", [])]
            e["fused_discriminator_are"] = "[mode (2: fm ring only, 1: + bins ring), ms per block, frac of HBM peak]"
            e["grid_6k25"] = [pick(p, "bins", "decim", "pfb_ms_per_block", frac="frac_of_hbm_peak") for p in rg.get("grid_6k25", [])]
            out["reference_grid_filterbank"] = e
    return out


def _scan(s):
    if not isinstance(s, dict):
        return None
    if "error" in s:
        return _err(s)
    out = pick(s, "fft_logmag_ms", "moving_sum_ms", "peak_pick_ms_incl_readback", "peaks_found", "peak_indices",
               "input_Msamples_per_s", "scan_ms_max_over_ranks", "peaks_found_rank0", "realtime_factor_at_25Msps")
    r = s.get("roofline")
    if isinstance(r, dict):
        out["roofline"] = pick(r, "bound", "achieved", "peak", "unit", "frac", "frac_fft_pass_alone", "algorithmic_bytes")
    return out


def compact(full, full_path=None):
    """-> dict for the last line.  `value`, `ms_per_step` and the roofline's own arithmetic keep full precision (the driver
    and tests/test_bench_contract.py recompute them); everything else is rounded to 5 significant digits."""
    exact = pick(full, "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config")
    r = full.get("roofline") or {}
    roof = pick(r, "bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                "algorithmic_bytes_filterbank", "algorithmic_bytes_stage2_rider", "stage2_rides_in_this_launch",
                "avg_launch_ms", "launches", "timed_every", "avg_launch_ms_every_launch_pass", "launches_every_launch_pass",
                "frac_every_launch_pass", "frac_by_rank", "frac_slowest_rank")
    if r.get("traffic") and r.get("algorithmic_bytes_per_launch"):
        roof["traffic_over_algorithmic"] = sig(r["traffic"] / r["algorithmic_bytes_per_launch"])
    src = r.get("traffic_source") or ""
    roof["traffic_source"] = ("pmc passes of this run (FETCH x2 + WRITE)" if src.startswith("measured in this run") else
                              "profiles/ file of an earlier run" if src else None)
    if isinstance(r.get("filterbank_alone"), dict):
        roof["filterbank_alone"] = pick(r["filterbank_alone"], "avg_launch_ms", "launches", "frac")
    exact["roofline"] = roof
    rest = {}
    c = full.get("cpu_baseline")
    if isinstance(c, dict):
        cb = pick(c, "value", "unit", "cores", "kind", "sample", "box_physical_cores", "channels", "largest_cpu_realtime_channels",
                  "largest_is", "gpu_channels_run_in_real_time_over_cpu_realtime_channels")
        cb.update(pick(c.get("host_cgroup") or {}, "cpu_quota_cores", "throttled_periods", "throttled_ms"))
        ac = c.get("all_cores") or {}
        cb["realtime_channels"] = {
            "one_core": (c.get("single_channel_one_core") or {}).get("realtime_channels_per_core_at_20Msps"),
            "measured_reference_structure": (ac.get("reference_structure_measured") or {}).get("realtime_channels"),
            "measured_time_tiled": (ac.get("best_cpu_time_tiled_measured") or {}).get("realtime_channels"),
            "threads_x_one_core": (ac.get("formula_threads_x_single_core") or {}).get("realtime_channels"),
            "box_cores_x_one_core": (ac.get("survey_formula_box_cores_x_single_core") or {}).get("realtime_channels"),
            "measured_over_threads_formula": ac.get("measured_over_formula")}
        if isinstance(c.get("gpu_fm_parity_vs_oracle"), dict):
            cb["gpu_fm_parity_vs_oracle"] = pick(c["gpu_fm_parity_vs_oracle"], "ok", "worst_fm_rms_error", "tolerance",
                                                 "channels_checked")
        if isinstance(cb.get("sample"), str) and len(cb["sample"]) > 200:      # (the whole sentence is in the full record)
            cb["sample"] = cb["sample"][:197] + "..."
        rest["cpu_baseline"] = cb
    else:
        rest["cpu_baseline"] = None
    s = full.get("sustained")
    if isinstance(s, dict):
        rest["sustained"] = pick(s, "seconds", "launches", "frac_first_window", "frac_last_window", "frac_slowest_window",
                                 "kernel_us_last_window", "sclk_mhz_last", "frac_last_window_by_rank")
    rest.update(pick(full, "ms_per_step_by_rank", "rccl_ranks", "transport", "ranks_started_by", "rccl_proof", "numa",
                     "peaks_allgather_us"))
    if isinstance(full.get("numa_by_rank"), list):
        rest["numa_node_by_rank"] = [(n or {}).get("numa_node") if (n or {}).get("pinned") else None for n in full["numa_by_rank"]]
    if isinstance(full.get("peaks_allgather"), dict):
        rest["peaks_allgather"] = pick(full["peaks_allgather"], "transport", "values_gathered", "values_expected", "ranks",
                                       "peaks_by_rank", "ok")
    if isinstance(full.get("realtime_per_gpu"), dict):
        rest["realtime_per_gpu"] = pick(full["realtime_per_gpu"], "front_ends_per_gpu", "front_ends_per_gpu_asked",
                                        "cpu_quota_cores", "ok_by_rank", "latency_ms_p99_by_rank", "deadline_misses_by_rank",
                                        "front_ends_sustained_total", "fm_channels_sustained_total",
                                        "input_Msps_sustained_total", "errors")
    if isinstance(full.get("channels"), dict):
        rest["channels"] = _channels(full["channels"])
    for k in ("scan", "scan_ref"):
        if k in full:
            rest[k] = _scan(full[k])
    e2e = full.get("end_to_end")
    if isinstance(e2e, dict):
        rest["end_to_end"] = _err(e2e) or pick(e2e, "pinned_cf32_push_iq_Msps", "pinned_u8_push_raw_Msps", "block_samples")
    gc = full.get("group_capacity")
    if isinstance(gc, dict):
        rest["group_capacity"] = _err(gc) or {
            "front_ends": gc.get("front_ends"), "grouped_over_one_by_one": gc.get("grouped_over_one_by_one"),
            "grouped_filterbank_frac": (gc.get("grouped") or {}).get("filterbank_frac_of_hbm_peak"),
            "grouped_input_Msps": (gc.get("grouped") or {}).get("input_Msps")}
    if "realtime" in full:
        rest["realtime"] = _realtime(full["realtime"])
    dm = full.get("daemon")
    if isinstance(dm, dict):
        rest["daemon"] = _err(dm) or pick(dm, "sources", "channels", "block_ms", "seconds", "input_Msps", "blocks", "late_blocks",
                                         "overruns", "latency_ms_p99", "latency_ms_max", "channels_delivering",
                                         "channel_rate_min_sps", "channel_rate_max_sps", "egress_MBps", "egress_errors")
    cp = full.get("control_plane")
    if isinstance(cp, dict):
        rest["control_plane"] = _err(cp) or pick(cp, "n", "create_ms_median", "release_ms_median", "connect_channel_new_ms_mean")
    out = dict(exact)
    out.update(sig(rest))
    out["full_record"] = full_path or "bench_full.json"
    # the guard: whatever a leg grows into, the line stays under LIMIT -- least important summaries go first
    for k in ("control_plane", "group_capacity", "end_to_end", "numa", "numa_node_by_rank", "rccl_proof", "scan",
              "scan_ref", "daemon", "realtime", "channels", "sustained", "realtime_per_gpu", "peaks_allgather"):
        if len(json.dumps(out)) + 1 <= LIMIT:
            break
        if k in out:
            if k in ("realtime", "channels") and isinstance(out[k], dict):       # first without their point lists
                for v in out[k].values():
                    if isinstance(v, dict):
                        v.pop("points", None)
                        v.pop("with_taps", None)
                        v.pop("grid_6k25", None)
                if len(json.dumps(out)) + 1 <= LIMIT:
                    break
            out[k] = "dropped from the compact line (size): see " + out["full_record"]
    if len(json.dumps(out)) + 1 > LIMIT:
        out["cpu_baseline"] = pick(out.get("cpu_baseline") or {}, "value", "unit", "cores", "kind", "largest_cpu_realtime_channels")
        out["config"] = pick(out["config"], "workload", "block_samples")
    return out


def write_full(full):
    """the full record next to bench.py's caller: ./bench_full.json and, on a gpurun box, gpurun_out/bench_full.json (that
    directory is what comes back from the box).  -> the path the compact line names"""
    text = json.dumps(full)
    named = None
    if os.environ.get("RCF_BENCH_FULL"):                      # tests: exactly this file
        with open(os.environ["RCF_BENCH_FULL"], "w") as f:
            f.write(text + "\n")
        return os.environ["RCF_BENCH_FULL"]
    for d in (os.path.join(ROOT, "gpurun_out"), os.getcwd()):
        try:
            os.makedirs(d, exist_ok=True)
            p = os.path.join(d, "bench_full.json")
            with open(p, "w") as f:
                f.write(text + "\n")
            named = named or os.path.relpath(p, ROOT)
        except OSError:
            continue
    return named
