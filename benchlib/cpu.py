"""cpu_baseline: the oracle's C port of the reference path on the host cores, and the parity check of the
timed configuration's FM outputs.  The ONLY place bench.py touches oracle/ (as the timed baseline and as the checker)."""
import numpy as np

from .common import FS, NB, cgroup_cpu_stat

def box_physical_cores():
    """physical cores of the whole box (every online CPU's thread_siblings_list), whatever this process may run on"""
    import glob
    seen = set()
    for f in glob.glob("/sys/devices/system/cpu/cpu[0-9]*/topology/thread_siblings_list"):
        try:
            seen.add((open(f.replace("thread_siblings_list", "physical_package_id")).read().strip(), open(f).read().strip()))
        except OSError:
            continue
    return len(seen)


def cpu_baseline(tile, carriers, fm_check=None, signal_seconds=2.0, reps=3, FS=FS, chans_per_thread=2):
    """The reference's structure on the host cores: one 2909-tap xlating FIR (D=800) + discriminator per channel over
    the whole 20 Msps stream (rc_frontend/channel.py:31-38), oracle C port (oracle/rcf_oracle.c: ro_bank_bench, the same
    arithmetic as the oracle's channel bank -- tests/test_oracle_kat.py holds the two together bit for bit).  This leg
    is the only place bench.py touches oracle/: as the timed CPU baseline and as the checker of the GPU's FM outputs.

    SURVEY 8(d) asks for (i) one channel on one core and (ii) all cores busy.  "All cores" is what this process may
    really use: the physical cores of its affinity mask (bench.py has pinned itself to the GPU's NUMA node), capped by the
    container's CFS quota (cpu.max) minus one core for the interpreter and the HIP runtime's threads -- more pinned
    threads than the quota only measures the throttle.  The cgroup's throttle counters are read
    around the leg and reported.  `signal_seconds` of signal per channel (a 0.25 s periodic tile walked 8 times), every
    thread on its own first-touched copy of the stream (the reference hands every channel flowgraph its own copy too:
    zeromq.pub_sink -> sub_source, channel.py:29), only the filtering timed.  The GPU is compared with the LARGEST of:
    the measured reference structure, the measured time-tiled form ("best CPU": blocks outer, the thread's channels
    inner -- GNU Radio does not run this), and SURVEY's formula over EVERY physical core of the box x single-core rate."""
    from oracle import cbind as OC
    from oracle import grspec as G
    D, taps = G.channel_params(FS, 12500)
    n_tile = int(FS * 0.25) // D * D
    x = np.tile(tile, (n_tile + len(tile) - 1) // len(tile))[:n_tile]
    passes = max(1, int(round(signal_seconds * FS / n_tile)))
    signal_s = passes * n_tile / FS
    cores = OC.physical_cores()                       # one logical CPU per physical core of the affinity mask
    quota = cgroup_cpu_stat()[3]
    n_avail = max(1, len(cores)) if cores else max(1, OC.max_threads())
    n_thr = n_avail if quota is None else max(1, min(n_avail, int(quota) - 1))
    box_cores = box_physical_cores() or n_avail
    cpu_ids = cores[:n_thr] if cores else None
    cpt = chans_per_thread
    n_ch = n_thr * cpt
    # the 32 bench carriers, repeated on a 12.5 kHz raster
    offs = [carriers[i % len(carriers)]["f_off"] + 12500.0 * (i // len(carriers)) for i in range(n_ch)]
    comp = [OC.xlating_composite(taps, D, f, FS) for f in offs]
    ct = np.stack([c[0] for c in comp])
    inc = np.array([c[1] for c in comp], dtype=np.complex64)
    gains = np.full(n_ch, G.p25_fm_gain(25000.0), dtype=np.float32)
    med = lambda v: sorted(v)[len(v) // 2]
    OC.bank_bench(x, 1, D, ct, inc, gains, n_thr, cpt, cpu_ids)                      # warm: threads, pages, clocks
    cg0 = cgroup_cpu_stat()
    # (i) a single channel on one core == one of the reference's per-channel GNU Radio flowgraphs
    t1 = med([OC.bank_bench(x, passes, D, ct[:1], inc[:1], gains[:1], 1, 1, cpu_ids)[0] for _ in range(reps)])
    # (ii) every usable core busy with `cpt` channels, the reference's structure (each channel walks the whole stream)
    t_ref = med([OC.bank_bench(x, passes, D, ct, inc, gains, n_thr, cpt, cpu_ids)[0] for _ in range(reps)])
    # (iii) "best CPU": time-tiled -- 64-output blocks (51 200 samples = 410 KB: L2) outer, EIGHT channels per thread
    # inner, so the stream comes from DRAM once per eight channels
    cpt_t = 8
    offs_t = [carriers[i % len(carriers)]["f_off"] + 12500.0 * (i // len(carriers)) for i in range(n_thr * cpt_t)]
    comp_t = [OC.xlating_composite(taps, D, f, FS) for f in offs_t]
    ct_t = np.stack([c[0] for c in comp_t])
    inc_t = np.array([c[1] for c in comp_t], dtype=np.complex64)
    g_t = np.full(len(offs_t), G.p25_fm_gain(25000.0), dtype=np.float32)
    p_t = max(1, passes // 2)                             # half the signal, four times the channels: same work bound
    t_tiled = med([OC.bank_bench(x, p_t, D, ct_t, inc_t, g_t, n_thr, cpt_t, cpu_ids, tiled=True, tile_block=64 * D)[0]
                   for _ in range(reps)])
    bw = OC.read_bandwidth(64 << 20, 4, n_thr, cpu_ids)
    cg1 = cgroup_cpu_stat()
    delta = lambda i, scale=1.0: (cg1[i] - cg0[i]) / scale if cg0[i] is not None and cg1[i] is not None else None
    rt_single = signal_s / t1
    rt_ref = n_ch * signal_s / t_ref
    rt_tiled = n_thr * cpt_t * (p_t * n_tile / FS) / t_tiled
    rt_formula = n_thr * rt_single                        # over the threads this leg ran: what `measured` compares with
    rt_formula_box = box_cores * rt_single                # over every physical core of the box: the comparison figure
    traffic_ref = n_ch * passes * n_tile * 8.0 / t_ref
    largest = max(rt_ref, rt_tiled, rt_formula_box)
    out = {
        "value": signal_s * FS / t_ref / 1e6,
        "unit": "Msamples/s",
        "cores": n_thr,
        "cores_are": "pinned threads, one per physical core of this process's affinity mask (%d cores there), capped by the "
                     "container's CPU quota minus one; the box has %d physical cores / %d hardware threads"
                     % (n_avail, box_cores, OC.max_threads()),
        "box_physical_cores": box_cores,
        "host_cgroup": {"cpu_quota_cores": quota, "throttled_periods": delta(0), "throttled_ms": delta(1, 1e3),
                        "cpu_seconds_used": delta(2, 1e6)},
        "kind": "port",
        "sample": "%.2f s of the same %g Msps synthetic stream per channel (a %.2f s periodic tile x %d), %d concurrent "
                  "12.5 kHz channels = %d per thread (%d-tap xlating FIR /%d + discriminator each), median of %d; "
                  "CPU restatement of the reference's GNU Radio path (GNU Radio itself unavailable)"
                  % (signal_s, FS / 1e6, n_tile / FS, passes, n_ch, cpt, len(taps), D, reps),
        "channels": n_ch,
        "realtime_channels_at_20Msps": rt_ref,
        "single_channel_one_core": {"seconds_per_second_of_signal": t1 / signal_s, "Msamples_per_s": signal_s * FS / t1 / 1e6,
                                    "realtime_channels_per_core_at_20Msps": rt_single},
        "all_cores": {
            "threads": n_thr,
            "reference_structure_measured": {"realtime_channels": rt_ref, "seconds": t_ref,
                                             "stream_read_GBps": traffic_ref / 1e9,
                                             "what": "channel outer, whole stream per channel: how GNU Radio runs it"},
            "formula_threads_x_single_core": {"realtime_channels": rt_formula,
                                              "what": "the %d threads of this leg x 1 / per-channel real-time fraction" % n_thr},
            "survey_formula_box_cores_x_single_core": {"realtime_channels": rt_formula_box, "cores": box_cores,
                                                       "what": "SURVEY 8(d): EVERY physical core of the box x single-core "
                                                               "rate (not measurable under the quota: an upper bound)"},
            "best_cpu_time_tiled_measured": {"realtime_channels": rt_tiled, "seconds": t_tiled,
                                             "channels": n_thr * cpt_t, "signal_seconds_per_channel": p_t * n_tile / FS,
                                             "what": "NOT the reference's structure: 64-output time blocks outer, the "
                                                     "thread's %d channels inner, stream read from DRAM once per thread" % cpt_t},
            "measured_over_formula": rt_ref / rt_formula,
            "read_bandwidth_of_these_threads_GBps": bw / 1e9,
            "measured_vs_formula": "measured / (threads x single-core) = %.2f with %d threads reading %.1f GB/s of stream "
                                   "(the same threads sum private buffers at %.1f GB/s); throttled %s ms in the leg"
                                   % (rt_ref / rt_formula, n_thr, traffic_ref / 1e9, bw / 1e9, delta(1, 1e3)),
        },
        "largest_cpu_realtime_channels": largest,
        "largest_is": ["reference_structure_measured", "best_cpu_time_tiled_measured",
                       "survey_formula_box_cores_x_single_core"][[rt_ref, rt_tiled, rt_formula_box].index(largest)],
    }
    if fm_check is not None:
        out["gpu_fm_parity_vs_oracle"] = fm_parity(G, tile, fm_check)
    return out


def fm_parity(G, tile, chk):
    """The timed configuration's own outputs against the oracle: the last ~300 discriminator samples of each of the
    32 FM channels after the timed loop vs PFB bin (float64 exact-phase bank) -> stage-2 xlating FIR /3 ->
    quadrature_demod on the same tail of the stream.  The resident block is the 2^20-sample tile repeated, so
    the tail is reproducible on the host."""
    taps, L = chk["taps"], 1 << 18
    x = np.tile(tile, 2)[-L:] if L <= 2 * len(tile) else None
    n_frames = L // NB
    f_end = chk["total_in"] // NB                 # PFB frames produced so far; the tail is frames [f_end - n_frames, f_end)
    f0 = f_end - n_frames
    warm = (len(taps) + NB - 1) // NB + 1          # frames that still see the tail's zero history
    j0 = warm + (-(f0 + warm)) % 3                 # first clean frame on the stage-2 decimation grid (frame % 3 == 0)
    bin_rate = FS / NB
    D2, taps2 = G.channel_params(bin_rate, 12500)
    worst, rows = 0.0, 0
    for c, fm in zip(chk["carriers"], chk["fm"]):
        stage1 = G.xlating_fir_exact(x, NB, taps, c["bin"] * FS / NB, FS).astype(np.complex64)
        yo = G.xlating_fir_ccc(stage1[j0:], D2, taps2, c["delta"], bin_rate)
        fo = G.quadrature_demod_cf(yo, 1.0)
        k_last = (f_end - 1) // 3                  # absolute stage-2 index of the newest output
        k_first = (f0 + j0) // 3                   # absolute index of fo[0]
        n_cmp = min(300, len(fo) - 8)
        ref = fo[k_last - k_first - n_cmp + 1: k_last - k_first + 1]
        got = fm[-n_cmp:]
        e = float(np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2)))
        worst = max(worst, e)
        rows += 1
    return {"channels_checked": rows, "samples_per_channel": 300, "worst_fm_rms_error": worst,
            "tolerance": 1e-4, "ok": bool(worst < 1e-4)}
