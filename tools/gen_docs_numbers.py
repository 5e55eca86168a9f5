#!/usr/bin/env python3
"""Every measured number DESIGN.md 5, README.md and profiles/README.md quote for the current round, generated from the
tracked files under profiles/ -- so that prose cannot drift from the evidence.

    python tools/gen_docs_numbers.py r04            rewrite the blocks between the markers in the three documents
    python tools/gen_docs_numbers.py r04 --check    exit 1 if a document's block differs from what the files give

Markers:  <!-- numbers:<name> <round> -->  ...  <!-- /numbers:<name> -->      names: measured, files
tests/test_docs_numbers.py runs the check on the newest round that has a bench line.
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def _j(name):
    f = os.path.join(P, name)
    if not os.path.exists(f):
        return None
    with open(f) as fh:
        text = fh.read().strip()
    try:
        return json.loads(text)
    except Exception:
        lines = [l for l in text.splitlines() if l.startswith("{")]
        return json.loads(lines[-1]) if lines else None


def _stats(name, pat):
    f = os.path.join(P, name)
    if not os.path.exists(f):
        return None
    for r in csv.DictReader(open(f)):
        if re.search(pat, r["Name"]):
            return {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) * 1e-3, "name": r["Name"]}
    return None


def f3(x):
    return "%.3f" % x


def latest_round():
    rs = sorted(int(re.search(r"r(\d+)_bench\.json", f).group(1)) for f in glob.glob(os.path.join(P, "r[0-9][0-9]_bench.json")))
    return "r%02d" % rs[-1] if rs else None


def measured_block(R):
    """the table of this round's numbers; every cell names the file (and key) it comes from"""
    L = []
    add = L.append
    b = _j("%s_bench.json" % R)
    if b is None:
        return "(no %s_bench.json under profiles/)" % R
    ro = b["roofline"]
    add("| what | value | where it comes from |")
    add("|---|---|---|")
    add("| headline `value` (BASELINE configs[1], N = 1) | %.0f Msamples/s, step %.4f ms | `%s_bench.json`: `value`, `ms_per_step` |"
        % (b["value"], b["ms_per_step"], R))
    add("| filterbank launch, HIP events in the timed region | %.1f µs over %d timed launches ⇒ %.0f GB/s = **%s** of 8 TB/s | `%s_bench.json`: `roofline.avg_launch_ms`, `.launches`, `.achieved`, `.frac` |"
        % (ro["avg_launch_ms"] * 1e3, ro["launches"], ro["achieved"], f3(ro["frac"]), R))
    if ro.get("stage2_rides_in_this_launch"):
        add("| … what that launch carries | filterbank %.1f MB + the previous block's stage-2 workgroups %.1f MB = %.1f MB algorithmic per launch | `%s_bench.json`: `roofline.algorithmic_bytes_filterbank`, `.algorithmic_bytes_stage2_rider`, `.algorithmic_bytes_per_launch` |"
            % (ro["algorithmic_bytes_filterbank"] / 1e6, ro["algorithmic_bytes_stage2_rider"] / 1e6, ro["algorithmic_bytes_per_launch"] / 1e6, R))
    fa = ro.get("filterbank_alone")
    if fa:
        add("| the filterbank alone (stage-2 lag off for a pass of its own, every launch timed) | %.1f µs over %d launches ⇒ **%s** (16 B × block / launch time: the figure of rounds 1–4) | `%s_bench.json`: `roofline.filterbank_alone` |"
            % (fa["avg_launch_ms"] * 1e3, fa["launches"], f3(fa["frac"]), R))
    if ro.get("avg_launch_ms_every_launch_pass"):
        add("| every launch timed (a pass of %d commits right behind the timed region, events on each launch) | %.1f µs ⇒ %s | `%s_bench.json`: `roofline.avg_launch_ms_every_launch_pass`, `.frac_every_launch_pass` |"
            % (ro["launches_every_launch_pass"], ro["avg_launch_ms_every_launch_pass"] * 1e3, f3(ro["frac_every_launch_pass"]), R))
    st = _stats("%s_bench_kernel_stats.csv" % R, r"pfb_kernel_os<256, 1, 14, 4, false>")
    hu = _j("%s_bench_head_under_rocprof.json" % R)
    if st:
        frac = ro.get("algorithmic_bytes_per_launch", 16.0 * b["config"]["block_samples"]) / (st["avg_us"] * 1e-6) / 1e9 / 8000.0
        add("| the same kernel by `rocprofv3 --kernel-trace --stats` | %.1f µs over %d launches ⇒ **%s** | `%s_bench_kernel_stats.csv`, row `pfb_kernel_os<256, 1, 14, 4, false>` |"
            % (st["avg_us"], st["calls"], f3(frac), R))
    sn = _stats("%s_bench_nolag_kernel_stats.csv" % R, r"pfb_kernel_os<256, 1, 14, 4, false>")
    if sn:
        add("| the filterbank ALONE by `rocprofv3 --kernel-trace --stats` (the same headline run with `RCF_S2_LAG=0`: no rider in the launch) | %.1f µs over %d launches ⇒ **%s** (16 B × block / launch time: the row of rounds 1–4) | `%s_bench_nolag_kernel_stats.csv`, row `pfb_kernel_os<256, 1, 14, 4, false>` |"
            % (sn["avg_us"], sn["calls"], f3(16.0 * b["config"]["block_samples"] / (sn["avg_us"] * 1e-6) / 1e9 / 8000.0), R))
    if hu:
        add("| … and the bench line printed in that profiled run | %.1f µs (`frac` %s) | `%s_bench_head_under_rocprof.json`: `roofline.avg_launch_ms` |"
            % (hu["roofline"]["avg_launch_ms"] * 1e3, f3(hu["roofline"]["frac"]), R))
    su = b.get("sustained")
    if su:
        add("| sustained leg, %.1f s / %d launches | first window %s, last %s, slowest %s of the peak | `%s_bench.json`: `sustained.frac_*_window` |"
            % (su["seconds"], su["launches"], f3(su["frac_first_window"]), f3(su["frac_last_window"]), f3(su["frac_slowest_window"]), R))
    s60 = _j("%s_sustained_60s.json" % R)
    if s60 and s60.get("sustained"):
        q = s60["sustained"]
        add("| the same, %.0f s / %d launches (`--sustained-seconds 60`) | first window %s, last %s, slowest %s; launch %.1f–%.1f µs over the sampled windows; wall %.4f ms per step | `%s_sustained_60s.json`: `sustained` |"
            % (q["seconds"], q["launches"], f3(q["frac_first_window"]), f3(q["frac_last_window"]), f3(q["frac_slowest_window"]),
               min(q["kernel_us_by_window"]), max(q["kernel_us_by_window"]), q["wall_ms_per_step"], R))
    if ro.get("traffic") and ro.get("traffic_fetch_x2_bytes"):
        add("| HBM traffic per launch, PMC child passes of the SAME run | FETCH×2 %.1f MB + WRITE %.1f MB = %.1f MB = %.3f × algorithmic | `%s_bench.json`: `roofline.traffic`, `.traffic_fetch_x2_bytes`, `.traffic_write_bytes` |"
            % (ro["traffic_fetch_x2_bytes"] / 1e6, ro["traffic_write_bytes"] / 1e6, ro["traffic"] / 1e6,
               ro["traffic"] / ro["algorithmic_bytes_per_launch"], R))
    tr = _j("pfb_traffic.json")
    if tr:
        add("| HBM traffic per 256-bin launch (PMC, separate passes) | FETCH×2 %.1f MB + WRITE %.1f MB = %.1f MB = %.3f × algorithmic | `pfb_traffic.json` (%s) |"
            % (tr["fetch_bytes_corrected_x2"] / 1e6, tr["write_bytes"] / 1e6, tr["hbm_bytes_per_launch"] / 1e6,
               tr["hbm_bytes_per_launch"] / tr["algorithmic_bytes_per_launch"], tr.get("measured", "?")))
    k = b.get("kernel_ms_per_step", {})
    if k:
        if ro.get("stage2_rides_in_this_launch"):
            add("| step = ONE launch: filterbank with the previous block's stage-2 riding in it | %.4f ms of kernel time per step (+ %.4f ms: the one stage-2 flush the timing read forces, spread over the steps) | `%s_bench.json`: `kernel_ms_per_step` |"
                % (k["pfb"], k.get("stage2_fir_with_fused_discriminator") or 0.0, R))
        else:
            add("| step = filterbank + stage-2 FIR with fused discriminator | %.4f + %.4f ms | `%s_bench.json`: `kernel_ms_per_step` |"
                % (k["pfb"], k["stage2_fir_with_fused_discriminator"], R))
    c5 = _j("%s_bench_cfg5.json" % R)
    if c5:
        add("| cfg5 (BASELINE configs[4] per GPU: 512 bins, 25 Msps) | %.0f Msamples/s, launch %.1f µs = **%s**, sustained %s | `%s_bench_cfg5.json`: `value`, `roofline`, `sustained.frac_last_window` |"
            % (c5["value"], c5["roofline"]["avg_launch_ms"] * 1e3, f3(c5["roofline"]["frac"]),
               f3(c5["sustained"]["frac_last_window"]) if c5.get("sustained") else "–", R))
        s5 = _stats("%s_bench_cfg5_kernel_stats.csv" % R, r"pfb_kernel_2b<256, 14")
        if s5:
            add("| the 512-bin kernel by rocprofv3 | %.1f µs over %d launches ⇒ **%s** | `%s_bench_cfg5_kernel_stats.csv`, row `pfb_kernel_2b<256, 14, 3, false>` |"
                % (s5["avg_us"], s5["calls"], f3(16.0 * c5["config"]["block_samples"] / (s5["avg_us"] * 1e-6) / 1e9 / 8000.0), R))
    for nb in (512, 1024):
        t = _j("%s_pfb%d_traffic.json" % (R, nb))
        if t and "fetch_x2_over_algorithmic_read" in t:
            add("| %d-bin bank, PMC passes of `tools/pfb_probe.py` | FETCH×2 = %.3f × algorithmic read, FETCH×2 + WRITE = %.3f × algorithmic; %.1f µs per launch under the counters | `%s_pfb%d_traffic.json` |"
                % (nb, t["fetch_x2_over_algorithmic_read"], t["hbm_bytes_over_algorithmic"],
                   t["pass3"]["kernel_us_in_this_pass"], R, nb))
    g = b.get("channels", {}).get("reference_grid_filterbank")
    if g:
        add("| 1600-bin reference-grid bank (every bin one `channel.py` channel) | %.4f ms per 2^25 block = **%s**; sustained %s | `%s_bench.json`: `channels.reference_grid_filterbank.roofline.frac`, `.sustained.frac_last_window` |"
            % (g["pfb_ms_per_block"], f3(g["roofline"]["frac"]), f3(g["sustained"]["frac_last_window"]), R))
        for x in g.get("grid_6k25", []):
            add("| 3200 bins, decim %d, %d taps | %.4f ms = %s | `%s_bench.json`: `…grid_6k25[]` |"
                % (x["decim"], x["taps"], x["pfb_ms_per_block"], f3(x["frac_of_hbm_peak"]), R))
        for pt in g.get("with_taps", {}).get("points", []):
            add("| … with %d bins tapped and demodulated%s | bank %.4f ms (%.2f × untapped), finalize %.4f ms | `…with_taps.points[]` |"
                % (pt["bins_tapped"], " (discriminator only: `rcf_chan_set_fm_only`)" if pt.get("discriminator_only") else "",
                   pt["pfb_ms_per_block"], pt["pfb_over_untapped"], pt["tap_finalize_ms_per_block"]))
    fd = (g or {}).get("fused_discriminator")
    if fd:
        for pt in fd.get("points", []):
            if "error" in pt:
                continue
            add("| … every bin demodulated INSIDE the bank's launch (`rcf_pfb_fm_enable(%d)`: %s) | %.4f ms per 2^25 block (%.2f × the untapped bank%s); %.0f MB algorithmic ⇒ %s of 8 TB/s | `…fused_discriminator.points[]` |"
                % (pt["mode"], pt["what"], pt["pfb_ms_per_block"], pt["over_untapped_bank"],
                   "; the bank + `tap_finalize` path: %.4f ms" % fd["two_kernel_path_ms_per_block"] if fd.get("two_kernel_path_ms_per_block") and pt["mode"] == 2 else "",
                   pt["algorithmic_bytes_per_launch"] / 1e6, f3(pt["frac_of_hbm_peak"])))
    fk = _stats("%s_bench_legs_kernel_stats.csv" % R, r"pfb5_fmlb_kernel<20, 4, 2, 2, false, 2>")
    if fk:
        add("| … the fused kernel by `rocprofv3 --kernel-trace --stats` (the untimed legs under the tracer: its warm-up and timed launches) | %.1f µs over %d launches ⇒ %s of 8 TB/s on 16 B × 2^25 | `%s_bench_legs_kernel_stats.csv`, row `pfb5_fmlb_kernel<20, 4, 2, 2, false, 2>` |"
            % (fk["avg_us"], fk["calls"], f3(16.0 * (1 << 25) / (fk["avg_us"] * 1e-6) / 1e9 / 8000.0), R))
    t32 = [(_j("%s_pfb3200_d1600_pmc.json" % R), 1600), (_j("%s_pfb3200_d800_pmc.json" % R), 800)]
    if all(t and "fetch_x2_over_algorithmic_read" in t for t, _ in t32):
        add("| 3200-bin banks, PMC passes of `tools/pfb_probe.py` | %s | `%s_pfb3200_d1600_pmc.json`, `%s_pfb3200_d800_pmc.json` |"
            % ("; ".join("decim %d: FETCH×2 = %.3f × algorithmic read, FETCH×2 + WRITE = %.3f × algorithmic" % (
                dd, t["fetch_x2_over_algorithmic_read"], t["hbm_bytes_over_algorithmic"]) for t, dd in t32), R, R))
    db = b.get("channels", {}).get("direct_bank")
    if db:
        top = db["points"][-1]
        add("| reference-shaped direct bank (2909-tap xlating FIR per channel, FP32 matrix cores) | %d channels run in real time; at %d: kernel %.1f ms / wall %.1f ms per %.1f ms block, %.1f TFLOP/s = %s of 157.3 | `%s_bench.json`: `channels.direct_bank` |"
            % (db["channels_run_in_real_time"], top["channels"], top["kernel_ms_per_block"], top["wall_ms_per_block"],
               top["block_ms_of_signal"], top["tflops_fp32"], f3(top["frac_of_fp32_matrix_peak"]), R))
    sc = b.get("scan")
    if sc:
        add("| scan (BASELINE configs[2]: N = 2^20 × 1000 frames, 100-frame sum) | FFT+log %.2f ms, running sum %.2f ms, pick %.2f ms; %.0f Msamples/s; **%s** of the HBM peak; %d peaks | `%s_bench.json`: `scan` |"
            % (sc["fft_logmag_ms"], sc["moving_sum_ms"], sc["peak_pick_ms_incl_readback"], sc["input_Msamples_per_s"],
               f3(sc["roofline"]["frac"]), sc["peaks_found"], R))
    sr = b.get("scan_ref")
    if sr and "roofline" in sr:
        add("| scan at the reference's own size (`fft_vector.py:31-60`: fs = 2.4 Msps, N = 16384, 1000 frames / 100-frame average) | FFT+log %.3f ms, running sum %.3f ms, pick %.3f ms; %.0f Msamples/s = %.0f × real time; %s of the HBM peak on 12 B/sample (FFT pass alone %s); %d peaks | `%s_bench.json`: `scan_ref` |"
            % (sr["fft_logmag_ms"], sr["moving_sum_ms"], sr["peak_pick_ms_incl_readback"], sr["input_Msamples_per_s"],
               sr.get("realtime_factor_at_2.4Msps", 0.0), f3(sr["roofline"]["frac"]), f3(sr["roofline"].get("frac_fft_pass_alone", 0.0)),
               sr["peaks_found"], R))
    dm = b.get("daemon")
    if dm and "error" not in dm:
        add("| **the product's data plane** (`rcf.receiver` + `rcf.dataplane.NativeDataPlane`: source rings → native pump → per-channel host rings → `socket.send`), %d × 20 Msps u8 sources, %d channels, %.1f s | %.0f Msamples/s in, %d blocks of %.1f ms: %d late, %d overruns, latency p99 %.2f / max %.2f ms; every channel delivered %.0f–%.0f samples/s (%d of %d), %.0f MB/s to the sockets | `%s_bench.json`: `daemon` |"
            % (dm["sources"], dm["channels"], dm["seconds"], dm["input_Msps"], dm["blocks"], dm["block_ms"], dm["late_blocks"],
               dm["overruns"], dm["latency_ms_p99"], dm["latency_ms_max"], dm["channel_rate_min_sps"], dm["channel_rate_max_sps"],
               dm["channels_delivering"], dm["channels"], dm["egress_MBps"], R))
    e = b.get("end_to_end")
    if e:
        add("| PCIe-inclusive ingest (never `value`) | cf32 %.0f Msamples/s, u8 %.0f Msamples/s | `%s_bench.json`: `end_to_end` |"
            % (e["pinned_cf32_push_iq_Msps"], e["pinned_u8_push_raw_Msps"], R))
    gc = b.get("group_capacity")
    if gc:
        a, o = gc["grouped"], gc["one_by_one"]
        add("| %d front-ends (256 bins + 32 FM each, %d-sample blocks resident) committed back to back | grouped: %.3f ms per group block = %.0f Msamples/s, ONE filterbank launch %.1f µs = **%s** of 8 TB/s, one stage-2 launch %.1f µs; one by one: %.3f ms, %.1f µs per launch = %s; grouped / one by one %.2f × | `%s_bench.json`: `group_capacity` |"
            % (gc["front_ends"], gc["block_samples"], a["wall_ms_per_group_block"], a["input_Msps"], a["filterbank_launch_us"],
               f3(a["filterbank_frac_of_hbm_peak"]), a["stage2_launch_us"], o["wall_ms_per_group_block"], o["filterbank_launch_us"],
               f3(o["filterbank_frac_of_hbm_peak"]), gc["grouped_over_one_by_one"], R))
    gs = _stats("%s_group_capacity_kernel_stats.csv" % R, r"pfb_group_kernel_os<256")
    if gs:
        add("| the grouped filterbank launch by `rocprofv3 --kernel-trace --stats` (`tools/group_probe.py`, 80 front-ends) | %.1f µs over %d launches ⇒ **%s** (16 B × 80 × 409600 samples per launch) | `%s_group_capacity_kernel_stats.csv`, row `pfb_group_kernel_os<256, …>` |"
            % (gs["avg_us"], gs["calls"], f3(16.0 * 80 * 409600 / (gs["avg_us"] * 1e-6) / 1e9 / 8000.0), R))
    rt = b.get("realtime")
    if rt:
        for shape, label in (("pfb256", "256-bin bank + 32 FM"), ("grid1600", "1600-bin reference-grid bank, 256 bins demodulated"),
                             ("grid1600fm", "1600-bin reference-grid bank with the discriminator of ALL 1600 bins fused into its launch, 256 bins delivered to host rings")):
            s = rt.get(shape)
            if not s:
                continue
            a = s.get("at_K_max") or {}
            add("| **paced real time**, %s, 20 Msps u8 per front-end, %.0f ms blocks, %d native pump threads, ONE attempt per point | K_max_first_attempt = **%d** front-ends%s (%d bins, %d demodulated channels, %.0f Msamples/s), first K that missed: %s; the %.0f s confirmation run at K_max: latency p50 %.2f / p99 %.2f / max %.2f ms, %d misses, %d overruns, %.1f front-ends per group block, GPU busy %.0f %%, PCIe in %.1f GB/s | `%s_bench.json`: `realtime.%s` |"
                % (label, a.get("block_ms", 0), rt.get("pump_threads", 0), s.get("K_max_first_attempt", s.get("K_max", 0)),
                   ("" if s.get("K_max_first_attempt", s.get("K_max")) == s.get("K_max") else
                    "; its confirmation run MISSED (%s), **K_max = %d** confirmed, largest K with p99 < 5 ms in every run: %s" % (
                        "; ".join("%d front-ends: %d misses, longest device wait %.1f ms" % (q["front_ends"], q.get("deadline_misses", 0), q.get("host_longest_device_wait_ms") or 0)
                                  for q in (s.get("points") or []) if q.get("confirmation_run") and not q.get("ok")),
                        s.get("K_max", 0), s.get("K_max_p99_under_5ms", "n/a"))),
                   s["channels_sustained"], s["fm_channels_sustained"], s["input_Msps_sustained"], s["first_K_that_missed"],
                   rt.get("seconds_of_the_confirmation_run_at_K_max", 0), a.get("latency_ms_p50") or 0, a.get("latency_ms_p99") or 0,
                   a.get("latency_ms_max") or 0, a.get("deadline_misses", 0), a.get("ring_overruns", 0),
                   a.get("front_ends_per_group_block_mean") or 0, a.get("gpu_busy_percent_est") or 0, a.get("pcie_GBps_in", 0), R, shape))
            pts = s.get("points") or []
            if pts:
                add("| … every point of that search (front-ends: p99 / max ms, misses) | %s | `realtime.%s.points[]` |"
                    % ("; ".join("%d: %.2f / %.2f, %d" % (q["front_ends"], q.get("latency_ms_p99") or 0, q.get("latency_ms_max") or 0,
                                                            q.get("deadline_misses", 0)) for q in pts), shape))
    for shape, kern in (("pfb256", "pfb_group_kernel_os<256"), ("grid1600", "pfb5_group_kernel<20, 4, 2, 2>")):
        ss = _j("%s_rt_%s_steady_state.json" % (R, shape))
        if not ss:
            continue
        e = next((v for k, v in ss["kernels"].items() if k.startswith(kern)), None)
        tot = sum(v["total_ms"] for v in ss["kernels"].values())
        if e:
            add("| kernel trace of the paced leg, %s, steady state (after %.1f s) | grouped filterbank: %d launches, mean %.1f µs (max %.0f µs), %.3f GB per launch ⇒ %.0f GB/s = **%s** of 8 TB/s; it is %.0f %% of the traced kernel time | `%s_rt_%s_steady_state.json` (`tools/rt_trace_outliers.py`), totals in `%s_rt_%s_kernel_stats.csv` |"
                % (shape, ss["steady_state_from_s_after_first_group_block"], e["launches"], e["avg_us"], e["max_us"],
                   e["algorithmic_GB_per_launch_mean"], e["achieved_GBps"], f3(e["frac_of_8TBps"]), 100.0 * e["total_ms"] / tot, R, shape, R, shape))
    cb = b.get("cpu_baseline")
    if cb and "all_cores" in cb:
        ac = cb["all_cores"]
        hc = cb.get("host_cgroup") or {}
        add("| CPU baseline (oracle C port; %d threads under a CPU quota of %s cores, box: %d physical cores; throttled %.0f ms during the leg) | one channel on one core %.1f × real time; reference structure on those threads %.0f real-time channels; threads × single-core %.0f; SURVEY formula (every physical core × single-core, an upper bound under the quota) %.0f; time-tiled best CPU %.0f | `%s_bench.json`: `cpu_baseline` |"
            % (cb["cores"], hc.get("cpu_quota_cores", "?"), cb.get("box_physical_cores", 0), hc.get("throttled_ms", 0.0),
               cb["single_channel_one_core"]["realtime_channels_per_core_at_20Msps"],
               ac["reference_structure_measured"]["realtime_channels"],
               (ac.get("formula_threads_x_single_core") or {}).get("realtime_channels", 0.0),
               ac["survey_formula_box_cores_x_single_core"]["realtime_channels"] if "survey_formula_box_cores_x_single_core" in ac
               else ac["survey_formula_cores_x_single_core"]["realtime_channels"],
               ac["best_cpu_time_tiled_measured"]["realtime_channels"], R))
        if "gpu_channels_run_in_real_time_over_cpu_realtime_channels" in cb:
            add("| GPU / CPU concurrent channels at 20 Msps | %.1f × (against the largest CPU figure, %.0f) | `cpu_baseline.gpu_channels_run_in_real_time_over_cpu_realtime_channels` |"
                % (cb["gpu_channels_run_in_real_time_over_cpu_realtime_channels"], cb["largest_cpu_realtime_channels"]))
        if "gpu_fm_parity_vs_oracle" in cb:
            pv = cb["gpu_fm_parity_vs_oracle"]
            add("| the timed configuration's own FM outputs vs the oracle | worst rms %.1e over %d channels (bar %.0e) | `cpu_baseline.gpu_fm_parity_vs_oracle` |"
                % (pv["worst_fm_rms_error"], pv["channels_checked"], pv["tolerance"]))
    cp = b.get("control_plane")
    if cp:
        add("| control plane, 100 × create / release through `frontend_connector` | create %.3f ms median, release %.3f ms, new channel %.3f ms | `%s_bench.json`: `control_plane` |"
            % (cp["create_ms_median"], cp["release_ms_median"], cp["connect_channel_new_ms_mean"], R))
    t2 = _j("%s_bench_2ranks_1gpu.json" % R)
    if t2:
        add("| `RCF_BENCH_DEVICE=0 python bench.py --gpus 2` (no launcher; both ranks on the one GPU of the box) | `n_gpus` %d, %s, transport %s, step by rank %s ms, peak gather %.0f µs | `%s_bench_2ranks_1gpu.json` |"
            % (t2["n_gpus"], t2["ranks_started_by"], t2["transport"], ", ".join("%.4f" % v for v in t2["ms_per_step_by_rank"]),
               t2.get("peaks_allgather_us") or 0, R))
    ub = _j("%s_unpinned_bounds.json" % R)
    if ub:
        l = ub["log2"]
        add("| unpinned GNU Radio details, worst case (CPU, `tools/unpinned_bounds.py`) | log2 ±%.0e per value: %d of %d index lists changed (N = 16384), %d of %d (N = 2^20); first index moves at ±%s; atan literals ±1 digit: fm ≤ %.1e; summation orders: ≤ %.1e between two float32 orders; rotator FMA: phase step ≤ %.1e rad, common phase ≤ %.1e rad after 10^6 outputs | `%s_unpinned_bounds.json` |"
            % (l["N=16384"]["per_value_error_log2_units"], l["N=16384"]["index_lists_changed"], l["N=16384"]["patterns"],
               l["N=1048576"]["index_lists_changed"], l["N=1048576"]["patterns"], l.get("smallest_error_that_moves_an_index_N=16384"),
               ub["atan_table"]["max_fm_change_p25_gain"],
               max(v["largest_between_two_float32_orders"] for v in ub["summation"].values()),
               ub["rotator_fma"]["max_step_difference_rad"], ub["rotator_fma"]["max_phase_difference"], R))
    if ub and "voice_chain" in ub:
        vc = ub["voice_chain"]
        add("| … voice chain (f-2) | fm_deemph evaluation order / float32 accumulator: audio moves ≤ %.1e rms; pm_remez grid density 32, 64 vs 16: ≤ %.1e; resampler taps ±1 ulp: %.1e (audio rms %.2f) | `%s_unpinned_bounds.json`: `voice_chain` |"
            % (max(vc["fm_deemph_evaluation_order_audio_rms_change"].values()), max(vc["pm_remez_grid_density_audio_rms_change_vs_16"].values()),
               vc["resampler_taps_pm_1ulp_audio_rms_change"], vc["audio_rms"], R))
    if ub and "pfb_routing" in ub:
        rows = ub["pfb_routing"]["rows"]
        worst = max(max(r["measured_float_fwT0"], r["measured_double_fwT0"]) / r["predicted_fm_error_without_margin"] for r in rows)
        add("| … pfb-mode routing budget | measured bin fm error / tap-leakage prediction ≤ %.2f over %d bins (float and double `fwT0` builds), against the margin of 2.5 the routing rule applies | `%s_unpinned_bounds.json`: `pfb_routing.rows` |"
            % (worst, len(rows), R))
    mp = _j("%s_hbm_mix_probe.json" % R)
    if mp:
        best = mp["best"]
        cells = []
        cells.append("256 bins (1 : 1) %s / %s = %.2f" % (f3(ro["frac"]), f3(best["1:1"]["frac_of_peak"]), ro["frac"] / best["1:1"]["frac_of_peak"]))
        if c5:
            cells.append("512 bins (1 : 1) %s / %s = %.2f" % (f3(c5["roofline"]["frac"]), f3(best["1:1"]["frac_of_peak"]),
                                                               c5["roofline"]["frac"] / best["1:1"]["frac_of_peak"]))
        if g:
            cells.append("1600 bins (1 : 2) %s / %s = %.2f" % (f3(g["roofline"]["frac"]), f3(best["1:2"]["frac_of_peak"]),
                                                                g["roofline"]["frac"] / best["1:2"]["frac_of_peak"]))
            for x in g.get("grid_6k25", []):
                mix = "1:2" if x["decim"] == 1600 else "1:4"
                cells.append("3200 bins, decim %d (%s) %s / %s = %.2f" % (x["decim"], mix.replace(":", " : "), f3(x["frac_of_hbm_peak"]),
                                                                          f3(best[mix]["frac_of_peak"]), x["frac_of_hbm_peak"] / best[mix]["frac_of_peak"]))
        add("| what the memory system sustains for the same read : write mix with NO arithmetic (`tools/hbm_mix_probe.hip`: coalesced 16-byte accesses, ping-pong 268 MB blocks, best variant per mix) | read only %s, 1 : 1 %s, 1 : 2 %s, 1 : 4 %s, write only %s of 8 TB/s ⇒ kernel / ceiling: %s | `%s_hbm_mix_probe.json`: `best` |"
            % (f3(best["1:0"]["frac_of_peak"]), f3(best["1:1"]["frac_of_peak"]), f3(best["1:2"]["frac_of_peak"]),
               f3(best["1:4"]["frac_of_peak"]), f3(best["0:1"]["frac_of_peak"]), "; ".join(cells), R))
    return "\n".join(L)


def summary_block(R):
    """a dozen lines for DESIGN.md 5 and README.md; the full table is profiles/README.md"""
    b = _j("%s_bench.json" % R)
    if b is None:
        return "(no %s_bench.json under profiles/)" % R
    ro = b["roofline"]
    L = ["Measured on one MI355X, round %s (`profiles/%s_bench.json`; every row of the full table in `profiles/README.md` names its file and key):" % (R[1:].lstrip("0"), R), ""]
    add = L.append
    add("* headline: **%.0f Msamples/s** of input IQ (256 bins + 32 FM channels per front-end, block 2^25), step %.4f ms; the filterbank launch %.1f µs ⇒ %.0f GB/s = **%s of the 8 TB/s HBM peak** (HIP events in the timed region), PMC traffic %.3f × algorithmic"
        % (b["value"], b["ms_per_step"], ro["avg_launch_ms"] * 1e3, ro["achieved"], f3(ro["frac"]),
           (ro.get("traffic") or 0) / ro["algorithmic_bytes_per_launch"] if ro.get("traffic") else float("nan")))
    st = _stats("%s_bench_kernel_stats.csv" % R, r"pfb_kernel_os<256, 1, 14, 4, false>")
    if st:
        add("* the same kernel by `rocprofv3 --kernel-trace --stats`: %.1f µs over %d launches ⇒ %s (`%s_bench_kernel_stats.csv`)"
            % (st["avg_us"], st["calls"], f3(ro["algorithmic_bytes_per_launch"] / (st["avg_us"] * 1e-6) / 1e9 / 8000.0), R))
    fa = ro.get("filterbank_alone")
    su = b.get("sustained")
    if fa and su:
        add("* the filterbank alone %s; sustained (%.1f s of back-to-back commits) last window %s" % (f3(fa["frac"]), su["seconds"], f3(su["frac_last_window"])))
    db = (b.get("channels") or {}).get("direct_bank")
    cb = b.get("cpu_baseline") or {}
    if db and "channels_run_in_real_time" in db:
        add("* reference-shaped direct bank (2909-tap xlating FIR /800 + discriminator per channel, FP32 matrix cores): **%d channels** opened and run in real time at 20 Msps%s"
            % (db["channels_run_in_real_time"],
               " = %.1f × the largest CPU figure (%.0f channels: every physical core of the box × the measured single-core rate)" % (
                   cb["gpu_channels_run_in_real_time_over_cpu_realtime_channels"], cb["largest_cpu_realtime_channels"])
               if "gpu_channels_run_in_real_time_over_cpu_realtime_channels" in cb else ""))
    g = (b.get("channels") or {}).get("reference_grid_filterbank")
    if g and "roofline" in g:
        line = "* 1600-bin reference-grid bank (every bin one `channel.py` channel): %.4f ms per 2^25 block = %s" % (g["pfb_ms_per_block"], f3(g["roofline"]["frac"]))
        fd = g.get("fused_discriminator") or {}
        p2 = next((p for p in fd.get("points", []) if p.get("mode") == 2 and "error" not in p), None)
        if p2:
            line += "; all 1600 bins demodulated inside the bank's launch: **%.4f ms**" % p2["pfb_ms_per_block"]
            if fd.get("two_kernel_path_ms_per_block"):
                line += " (bank + `tap_finalize`: %.4f ms)" % fd["two_kernel_path_ms_per_block"]
        add(line)
    sc, sr = b.get("scan"), b.get("scan_ref")
    if sc and "roofline" in sc:
        add("* scan: N = 2^20 × 1000 frames %s of the HBM peak on 12 B/sample%s" % (
            f3(sc["roofline"]["frac"]), "; the reference's own size (N = 16384) %.0f × real time" % sr["realtime_factor_at_2.4Msps"] if sr and "realtime_factor_at_2.4Msps" in sr else ""))
    rt = b.get("realtime") or {}
    cells = []
    for shape, label in (("pfb256", "256-bin bank + 32 FM"), ("grid1600", "1600-bin bank, 256 bins demodulated"),
                         ("grid1600fm", "1600-bin bank, ALL bins demodulated in its launch")):
        s_ = rt.get(shape)
        if s_ and "K_max" in s_:
            a = s_.get("at_K_max") or {}
            cells.append("%s: K_max **%d** front-ends = %d FM channels (p99 %.1f ms, %d misses in the confirmation run)" % (
                label, s_["K_max"], s_.get("fm_channels_sustained", 0), a.get("latency_ms_p99") or 0, a.get("deadline_misses", 0)))
    if cells:
        add("* paced real time (20 Msps u8 per front-end at wall-clock rate, native pumps): " + "; ".join(cells))
    dm = b.get("daemon")
    if dm and "error" not in dm:
        add("* the product's own data plane (`rcf.dataplane`): %d × 20 Msps sources, %d channels all delivered, %d late blocks of %d, latency p99 %.2f ms"
            % (dm["sources"], dm["channels"], dm["late_blocks"] + dm["overruns"], dm["blocks"], dm["latency_ms_p99"]))
    if "value" in cb:
        add("* CPU baseline (C restatement of the GNU Radio path, %d threads under the container's quota): %.1f real-time channels per core" % (
            cb.get("cores", 0), (cb.get("single_channel_one_core") or {}).get("realtime_channels_per_core_at_20Msps", 0.0)))
    add("* multi-GPU scaling: unmeasured (no 8-GPU node has run it yet)")
    return "\n".join(L)


def _pmc_txt(name):
    """{counter: average per dispatch} of a tools/pmc_sets.sh summary"""
    f = os.path.join(P, name)
    if not os.path.exists(f):
        return None
    out = {}
    for line in open(f):
        m = re.match(r"(\w+)\s+([0-9.]+)\s+\(n=", line)
        if m:
            out[m.group(1)] = float(m.group(2))
    return out or None


def limiter_block(R):
    """what bounds the 1600-bin bank and the bank with the discriminator fused in, from the counter passes"""
    L, seen = [], []
    for fname, label in (("%s_pfb1600_limiter_pmc.txt" % R, "`pfb5_kernel<20,4,2,2>` (1600 bins, D = 800, block 2^25)"),
                         ("%s_pfb1600_fused_limiter_pmc.txt" % R, "`pfb5_fmlb_kernel<20,4,2,2>` (the same bank, every bin demodulated, discriminator ring only)")):
        c = _pmc_txt(fname)
        if not c or "GRBM_GUI_ACTIVE" not in c:
            continue
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0                      # per XCD = the dispatch's duration in shader clocks
        simds, cus = 1024.0, 256.0
        seen.append((c, cyc))
        parts = ["%s, `%s`: %.0f k shader cycles per dispatch under the counters" % (label, fname, cyc / 1e3)]
        if "SQ_WAVE_CYCLES" in c:
            parts.append("%.1f waves resident per SIMD on average (`SQ_WAVE_CYCLES` × 4 / 1024 SIMDs / cycles; 3.75 = three workgroups of five waves per CU, the LDS limit)" % (c["SQ_WAVE_CYCLES"] * 4 / simds / cyc))
        if "SQ_ACTIVE_INST_VALU" in c:
            parts.append("the vector ALU issues in **%.0f %%** of the cycles (`SQ_ACTIVE_INST_VALU` × 4 / 1024 / cycles; %.1f M wave-instructions, `SQ_INSTS_VALU`)" % (
                100 * c["SQ_ACTIVE_INST_VALU"] * 4 / simds / cyc, c.get("SQ_INSTS_VALU", 0) / 1e6))
        if "SQ_INSTS_LDS" in c and "SQ_LDS_BANK_CONFLICT" in c:
            parts.append("LDS: %.1f M instructions, %.2f bank-conflict cycles per instruction, the LDS pipe busy in %.0f %% of a CU's cycles (`SQ_ACTIVE_INST_LDS` × 4 / 256 CUs / cycles)" % (
                c["SQ_INSTS_LDS"] / 1e6, c["SQ_LDS_BANK_CONFLICT"] / c["SQ_INSTS_LDS"], 100 * c.get("SQ_ACTIVE_INST_LDS", 0) * 4 / cus / cyc))
        if "SQ_WAIT_INST_ANY" in c and "SQ_WAVE_CYCLES" in c:
            parts.append("a resident wave waits on an outstanding instruction %.0f %% of its time (`SQ_WAIT_INST_ANY` / `SQ_WAVE_CYCLES`), %.0f %% on LDS alone (`SQ_WAIT_INST_LDS`)" % (
                100 * c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 100 * c.get("SQ_WAIT_INST_LDS", 0) / c["SQ_WAVE_CYCLES"]))
        if "SQ_INSTS_VMEM_RD" in c:
            parts.append("%.2f M vector-memory reads and %.2f M writes" % (c["SQ_INSTS_VMEM_RD"] / 1e6, c.get("SQ_INSTS_VMEM_WR", 0) / 1e6))
        L.append("* " + "; ".join(parts) + ".")
    if not L:
        return "(no limiter counter files for %s)" % R
    L.append("")
    def util(c, cyc):
        return 100 * c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024.0 / cyc
    c0, y0 = seen[0]
    text = ("Reading.  The bank alone saturates no unit: HBM at the fraction of the table above, the vector ALU issuing in %.0f %% of "
            "the cycles, the LDS pipe in about half of a CU's.  A chunk is 53.7 KB of LDS, which allows three workgroups of five "
            "waves per CU and no fourth: with fewer than four waves per SIMD there is not enough independent work to cover one "
            "phase's latencies (the window's DMA, two barrier-separated FFT passes, the copy-out) with another's -- a resident "
            "wave waits on its own outstanding instructions for about a third of its time." % util(c0, y0))
    if len(seen) == 2 and all("SQ_INSTS_VALU" in c for c, _ in seen):
        c1, y1 = seen[1]
        text += ("  With the discriminator fused in the kernel issues %.2f x the vector instructions of the bank (the discriminator "
                 "of a frame is as many VALU instructions as the bank's own arithmetic for it) and lasts %.2f x as long under the "
                 "counters; its vector ALU issues in %.0f %% of the cycles%s.  It moves a third of the bytes of the bank + "
                 "`tap_finalize` path and is not bounded by them." % (
                     c1["SQ_INSTS_VALU"] / c0["SQ_INSTS_VALU"], y1 / y0, util(c1, y1),
                     ": the copy-out's arithmetic fills the issue slots the bank leaves idle, and the vector ALU is what bounds this kernel"
                     if util(c1, y1) > 75 else ""))
    L.append(text)
    return "\n".join(L)


def files_block(R):
    """what is tracked for the round (profiles/README.md)"""
    L = ["| file | present |", "|---|---|"]
    for f in sorted(os.listdir(P)):
        if f.startswith(R + "_") or f in ("pfb_traffic.json", "pfb512_traffic.json"):
            L.append("| `%s` | %d bytes |" % (f, os.path.getsize(os.path.join(P, f))))
    return "\n".join(L)


BLOCKS = {"measured": measured_block, "summary": summary_block, "limiter": limiter_block, "files": files_block}
DOCS = ["DESIGN.md", "README.md", os.path.join("profiles", "README.md")]


def apply(R, check=False):
    bad = []
    for doc in DOCS:
        path = os.path.join(ROOT, doc)
        text = open(path).read()
        new = text
        for name, fn in BLOCKS.items():
            pat = re.compile(r"(<!-- numbers:%s )(r\d+)( -->\n)(.*?)(\n<!-- /numbers:%s -->)" % (name, name), re.S)
            if not pat.search(new):
                continue
            body = fn(R)
            new = pat.sub(lambda m: m.group(1) + R + m.group(3) + body + m.group(5), new)
        if new != text:
            if check:
                bad.append(doc)
            else:
                open(path, "w").write(new)
                print("rewrote", doc)
    return bad


if __name__ == "__main__":
    R = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else latest_round()
    bad = apply(R, check="--check" in sys.argv)
    if bad:
        print("out of date with profiles/:", ", ".join(bad))
        sys.exit(1)
