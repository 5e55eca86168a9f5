"""The paced real-time leg: K independent 20 Msps front-ends fed at wall-clock rate by native pump threads."""
import os
import time

import numpy as np

from .common import FS, NB, cgroup_cpu_stat, cpu_busy_sample, idlest_cpus, proto_taps

def realtime_point(native, pool, K, shape, src, carriers, device, seconds, block_ms, n_pumps, stagger=True, window_ms=1.0):
    """K independent 20 Msps front-ends on ONE GPU (the reference's deployment shape: ten sources per host,
    configs/config_denver_dev_den817.py:25-118, all of them inside one receiver when no -i is given,
    rc_frontend/receiver.py:67-70), every one fed its own u8 stream -- what an SDR link delivers, 2 bytes per sample --
    at exactly 20 Msps of WALL-CLOCK time in blocks of `block_ms`.  The front-ends are shared out over `n_pumps` groups
    (rcf_group_*), each driven by ONE native thread (rcf_pump_*; no interpreter in the loop): whenever blocks of some of its
    members are complete the pump pushes them as one group block -- one conversion, one filterbank, one stage-2 / tap and
    one gather launch for all of them -- and every subscribed channel's discriminator output lands in its pinned host ring.
    Latency of a block = from the instant its last sample exists to its channels' outputs being in host memory.  A
    deadline is missed when that exceeds the block period; an overrun is a block the pump only got to more than one
    period after it was complete (the source's double buffer would have been overwritten)."""
    blk = int(round(FS * block_ms * 1e-3))
    period = blk / FS
    warm = max(2, int(round(1.0 / period)))               # the first second (lazy allocations, module loads, clocks): run, not judged
    n_blocks = max(4, int(round(seconds / period))) + warm
    t_setup = time.perf_counter()
    fes, chans = pool["fes"], pool["chans"]
    while len(fes) < K:
        if shape == "pfb256":
            fe = native.Frontend(FS, 0.0, device=device, block_capacity=blk, hist_capacity=1 << 14, out_capacity=1 << 12)
            fe.pfb_open(NB, NB, proto_taps(native))
            ids = [fe.pfb_chan_open(c["bin"] % NB, 12500, c["delta"]) for c in carriers]
        elif shape == "grid1600fm":
            # the same bank with the discriminator of EVERY bin fused into its launch (rcf_pfb_fm_enable, discriminator ring
            # only): all 1600 reference channels of the front-end are demodulated on the device, 256 of them are subscribed
            # (bins of the fused ring: rcf_pump_subscribe with RCF_SRC_PFB_BIN0 + bin) and land in host rings
            D, T = native.channel_params(FS, 12500)
            fe = native.Frontend(FS, 0.0, device=device, block_capacity=blk, hist_capacity=1 << 14, out_capacity=1 << 11)
            fe.pfb_open(1600, D, native.design_low_pass_2(1.0, FS, 6250.0, 6250.0, 20.0))
            fe.pfb_fm_enable(2, gr_phase=True)
            ids = [native.SRC_PFB_BIN0 + (7 + 6 * j) % 1600 for j in range(256)]
        else:                                            # the bank whose bins ARE the reference's channels + 256 of them tapped
            D, T = native.channel_params(FS, 12500)
            fe = native.Frontend(FS, 0.0, device=device, block_capacity=blk, hist_capacity=1 << 14, out_capacity=1 << 11)
            fe.pfb_open(1600, D, native.design_low_pass_2(1.0, FS, 6250.0, 6250.0, 20.0))
            ids = [fe.pfb_tap_open((7 + 6 * j) % 1600, gr_phase=True) for j in range(256)]
        fes.append(fe)
        chans.append(ids)
    fes, chans = fes[:K], chans[:K]
    def produced_total():
        if shape == "grid1600fm":                         # (every subscribed bin gets one sample per frame of its bank)
            return sum(fes[i].pfb_produced() * len(chans[i]) for i in range(K))
        return sum(fes[i].chan_produced(c) for i in range(K) for c in chans[i])
    produced0 = produced_total()
    n_ch = len(chans[0]) if chans else 0
    out_rate = FS / NB / 3 if shape == "pfb256" else 25000.0
    out_ring = 1 << max(10, int(np.ceil(np.log2(4 * out_rate * period))))   # four blocks of output per channel
    NP = max(1, min(n_pumps, K))
    groups, pumps = [], []
    cg0 = None
    # every front-end replays its OWN two blocks of the pinned source (K x 1.6 MB: no cache between the host's DRAM and
    # the GPU holds that); staggered: front-end i's blocks complete (i / K) of a period after front-end 0's -- independent
    # SDRs are not synchronised, and the GPU then sees a steady flow; burst: all at the same instant
    src_arr = src["array"]
    assert len(src_arr) >= 2 * blk * 2 * K
    t_classes = [native.T_PFB, native.T_FIR_DERIVED, native.T_TAPS, native.T_DISC]
    # The pump threads FLOAT over the process's mask (the GPU's NUMA node).  Measured (profiles/r06_paced_leg_why_late.json):
    # a pump pinned to one CPU -- even the idlest of the node -- wakes up to 12 ms late a few times per second, 99 % of that
    # lateness run-queue delay (someone else's thread on that CPU; no SCHED_FIFO for this container), and misses deadlines
    # at K = 384-768; floating pumps: four K = 768 runs without one wake-up > 2 ms late.  RCF_BENCH_RT_PIN=idle pins them.
    busy = cpu_busy_sample()                               # (everyone's load on the allowed CPUs before the run: reported)
    pump_cpus = idlest_cpus(NP, busy) if busy and os.environ.get("RCF_BENCH_RT_PIN", "none") == "idle" else []
    try:
        for j in range(NP):
            mine = list(range(j, K, NP))
            grp = native.Group([fes[i] for i in mine])
            groups.append(grp)
        fes[0].timing_enable(True, classes=t_classes)     # (the grouped launches of group 0 are timed on its first member)
        fes[0].timing_stride(4)
        for c_ in t_classes:
            fes[0].timing_read(c_)
        setup_s = time.perf_counter() - t_setup
        for j in range(NP):
            mine = list(range(j, K, NP))
            rings = [src_arr[2 * blk * 2 * i: 2 * blk * 2 * (i + 1)] for i in mine]
            n_sub = int(os.environ.get("RCF_BENCH_RT_SUBS", "-1"))       # diagnosis: subscribe only the first n channels of each front-end
            subs = [(m, c) for m, i in enumerate(mine) for c in (chans[i] if n_sub < 0 else chans[i][:n_sub])]
            pumps.append(native.Pump(groups[j], rings, blk, FS, subs, fmt=native.FMT_U8, scale=1.0 / 32, offset=127.4,
                                     what="fm", gain=1.0, phase_s=[(i / K) * period if stagger else 0.0 for i in mine],
                                     out_ring_samples=out_ring, n_blocks=n_blocks, warm_blocks=warm, start_delay_s=0.25,
                                     batch_window_s=window_ms * 1e-3, rt_priority=int(os.environ.get("RCF_BENCH_RT_PRIORITY", "10")),
                                     spin_us=int(os.environ.get("RCF_BENCH_RT_SPIN_US", "0")),
                                     cpu=pump_cpus[j] if j < len(pump_cpus) else -1))   # (spinning the idle waits: measured WORSE -- 40 ms device stalls in both 10 s runs, none with sleeps)
        t_end = time.perf_counter() + n_blocks * period + 0.25 + 10.0
        stats = []
        cg0 = cgroup_cpu_stat()
        while time.perf_counter() < t_end:
            stats = [p_.stats() for p_ in pumps]
            if not any(s_["running"] for s_ in stats):
                break
            time.sleep(0.05)
        stats = [p_.stats() for p_ in pumps]
        cg1 = cgroup_cpu_stat()
        hung = any(s_["running"] for s_ in stats)
    finally:
        for p_ in pumps:
            p_.stop()
    per_batch_ms = 0.0
    for c_ in t_classes:
        ms_, n_ = fes[0].timing_read(c_)
        per_batch_ms += ms_ / n_ if n_ else 0.0
    fes[0].timing_enable(False)
    for g_ in groups:
        g_.close()
    errors = [s_.get("error_text", "error %d" % s_["error"]) for s_ in stats if s_["error"]] + (["pump still running at the deadline"] if hung else [])
    produced = produced_total() - produced0
    read = sum(s_["samples_out"] for s_ in stats)
    wall = max(s_["elapsed_s"] for s_ in stats)
    miss = sum(s_["late"] for s_ in stats)
    over = sum(s_["overruns"] for s_ in stats)
    judged = sum(s_["blocks_judged"] for s_ in stats)
    batches = sum(s_["group_blocks"] for s_ in stats)
    return {
        "front_ends": K, "staggered": bool(stagger), "seconds": wall, "blocks_per_front_end": n_blocks - warm,
        "warmup_blocks_not_judged": warm, "block_ms": period * 1e3, "pump_threads": NP, "batch_window_ms": window_ms,
        "blocks_judged": judged, "deadline_misses": miss, "ring_overruns": over,
        "output_samples_lost": int(produced - read), "errors": errors,
        "latency_ms_p50": float(np.median([s_["latency_ms_p50"] for s_ in stats])),
        "latency_ms_p99": max(s_["latency_ms_p99"] for s_ in stats),
        "latency_ms_max": max(s_["latency_ms_max"] for s_ in stats),
        "latency_note": "p50: median over the pump threads; p99 / max: the worst pump thread's",
        "group_blocks": batches, "front_ends_per_group_block_mean": (judged + K * warm) / max(1, batches),
        "front_ends_per_group_block_max": max(s_["max_batch"] for s_ in stats),
        "host_plan_fraction_busiest_pump": max(s_["host_plan_ms"] for s_ in stats) * 1e-3 / wall,
        "host_longest_plan_ms": max(s_["max_plan_ms"] for s_ in stats), "host_longest_device_wait_ms": max(s_["max_wait_ms"] for s_ in stats),
        "host_longest_sleep_overshoot_ms": max(s_["max_sleep_overshoot_ms"] for s_ in stats),
        "pump_threads_sched_fifo": sum(s_["rt_priority_granted"] for s_ in stats),
        "pump_cpus": [s_.get("cpu", -1) for s_ in stats],
        "host_cpus": {"allowed": len(busy) or len(os.sched_getaffinity(0)),
                      "mean_busy_before_the_run": (sum(busy.values()) / len(busy)) if busy else None,
                      "over_50pct_busy_before_the_run": sum(1 for b_ in busy.values() if b_ > 0.5) if busy else None},
        # why late (rcf_pump_stats_t): of the summed lateness of the slow device waits (> 5 ms) / late wake-ups (> 2 ms), how
        # much the pump thread spent RUNNABLE WITHOUT A CPU (schedstat run_delay) -- that share is the host scheduler's
        "why_late": {"slow_device_waits_ms": sum(s_.get("slow_wait_ms_total", 0.0) for s_ in stats),
                     "of_it_on_a_run_queue_ms": sum(max(0.0, s_.get("runq_ms_in_slow_waits", 0.0)) for s_ in stats),
                     "late_wakeups_ms": sum(s_.get("slow_sleep_ms_total", 0.0) for s_ in stats),
                     "of_them_on_a_run_queue_ms": sum(max(0.0, s_.get("runq_ms_in_slow_sleeps", 0.0)) for s_ in stats),
                     "run_queue_ms_total_worst_pump": max(s_.get("runq_ms_total", -1.0) for s_ in stats),
                     "involuntary_switches": sum(s_.get("involuntary_switches", 0) for s_ in stats)},
        "slow_plans_waits_sleeps": [sum(s_[k_] for s_ in stats) for k_ in ("slow_plans", "slow_waits", "slow_sleeps")],
        "host_cgroup": {"cpu_quota_cores": cg1[3],
                        "throttled_periods": (cg1[0] - cg0[0]) if cg0 and cg0[0] is not None and cg1[0] is not None else None,
                        "throttled_ms": (cg1[1] - cg0[1]) / 1e3 if cg0 and cg0[1] is not None and cg1[1] is not None else None,
                        "cpu_cores_used_mean": (cg1[2] - cg0[2]) / 1e6 / wall if cg0 and cg0[2] is not None and cg1[2] is not None else None},
        "gpu_kernel_us_per_group_block_of_group_0": per_batch_ms * 1e3,
        "gpu_busy_percent_est": 100.0 * per_batch_ms * 1e-3 * batches / wall,
        "gpu_busy_note": "filterbank + stage-2 / tap-finalize launches of pump 0's group blocks (HIP events, every 4th) x all "
                         "group blocks / elapsed; the conversion and the gather launch are not in it",
        "pcie_GBps_in": K * FS * 2 / 1e9, "pcie_GBps_out": K * n_ch * out_rate * 4 / 1e9,
        "input_Msps_sustained": K * FS / 1e6, "setup_s": setup_s,
        "ok": not errors and miss == 0 and over == 0 and (produced == read or "RCF_BENCH_RT_SUBS" in os.environ) and judged == K * (n_blocks - warm),
    }


def realtime_leg(native, tile, carriers, device, seconds=10.0, block_ms=20.0, k_first=512, k_cap=1280,
                 shapes=("pfb256", "grid1600"), stagger=True, n_pumps=0, window_ms=1.0):
    """`sustained`, as the metric means it: how many 20 Msps front-ends one MI355X keeps up with in real time.  Short
    points (4 s judged) from k_first upwards in steps of k_first / 2 until one misses a deadline (or k_cap), downwards if
    the first one already misses; the K found is then CONFIRMED over `seconds`.  EVERY point is one attempt: a point that
    misses is a miss (`K_max_first_attempt` = the largest K whose FIRST run was clean, which is what `K_max` is too unless
    the confirmation run disagrees).  Per shape: pfb256 = BASELINE configs[1] per front-end (256-bin bank + 32 FM
    channels); grid1600 = the 1600-bin reference-grid bank (every bin one of channel.py's 25 kS/s channels) with 256
    bins tapped and demodulated; grid1600fm = the same bank with the discriminator of EVERY bin fused into its launch
    (all 1600 channels demodulated on the device, 256 of them delivered to host rings)."""
    blk = int(round(FS * block_ms * 1e-3))
    raw = native.PinnedArray(2 * blk * 2 * k_cap, np.uint8)   # two blocks of its own per front-end
    t8 = np.clip(np.round(tile.view(np.float32) * 32 + 127.4), 0, 255).astype(np.uint8)
    for b in range(2 * k_cap):                           # the tile read from a different start for every block
        at = 2 * ((b * 40961) % (len(tile) - blk))
        raw.array[2 * blk * b: 2 * blk * (b + 1)] = t8[at: at + 2 * blk]
    src = {"array": raw.array}
    if not n_pumps:
        try:
            n_pumps = max(1, min(4, (os.cpu_count() or 8) // 8))
        except Exception:
            n_pumps = 4
    out = {"what": "K independent 20 Msps u8 front-ends on one GPU, paced at wall-clock rate in %.0f ms blocks; %d native "
                   "pump threads (rcf_pump_*), each driving one group of front-ends (rcf_group_*): the blocks that are "
                   "complete go out as ONE conversion / filterbank / stage-2 or tap-finalize / gather launch, every channel's "
                   "discriminator output lands in its pinned host ring; %s" % (
                       block_ms, n_pumps,
                       "the front-ends' block boundaries are spread evenly over the block period (independent SDRs are not "
                       "synchronised)" if stagger else "every front-end's block completes at the same instant (worst case)"),
           "pump_threads": n_pumps, "seconds_of_the_confirmation_run_at_K_max": seconds,
           "seconds_per_search_point": min(4.0, seconds), "staggered": bool(stagger),
           "attempts_per_point": 1, "batch_window_ms": window_ms,
           "batch_window_note": "a complete block waits up to this long for the blocks that complete meanwhile: they share its launches"}
    for shape in shapes:
        if shape in ("grid1600", "grid1600fm"):
            k_cap = min(k_cap, 1024)                     # (256 tapped bins per front-end: a point above this does not pay for its setup time)
        pts, good, bad = [], None, None
        pool = {"fes": [], "chans": []}
        search_s = min(4.0, seconds)                     # the search runs short points; K_max is then CONFIRMED over `seconds`
        step = max(16, k_first // 2)

        def point(K, secs=None):
            secs = search_s if secs is None else secs
            try:
                p = realtime_point(native, pool, K, shape, src, carriers, device, secs, block_ms, n_pumps, stagger, window_ms)
            except Exception as e:                       # (out of memory opening front-end K, ...): a failed point, not a failed leg
                p = {"front_ends": K, "ok": False, "errors": ["%s: %s" % (type(e).__name__, e)], "seconds": 0.0}
            pts.append(p)
            return p

        K = min(k_first, k_cap)
        while K <= k_cap:
            if point(K)["ok"]:
                good = K
                if K == k_cap:
                    break
                K = min(K + step, k_cap)
            else:
                bad = K
                break
        while good is None and bad is not None and bad > 16:           # the starting point itself failed: search downwards
            K = bad - step if bad > step else bad // 2
            if point(K)["ok"]:
                good = K
            else:
                bad = K
        # (the starting point failed and a point a whole step below it held: one more point half-way, so that one hiccup at
        # the starting point does not cost the shape half its K)
        if good is not None and bad is not None and bad - good >= step and good + step // 2 < bad and len(pts) >= 2 and not pts[0].get("ok"):
            mid = good + step // 2
            if point(mid)["ok"]:
                good = mid
            else:
                bad = mid
        first_attempt = good or 0
        # confirmation: the K the short points found, over the full `seconds`; if it does not hold, one step less
        best = None
        for _ in range(3):
            if not good or seconds <= search_s:
                break
            p = point(good, seconds)
            p["confirmation_run"] = True
            if p["ok"]:
                best = p
                break
            bad, good = good, max(step // 2, good - step // 2)
        if best is None:
            best = next((p for p in reversed(pts) if p["front_ends"] == good and p["ok"]), None)
            if best is None:
                good = 0
        for fe in pool["fes"]:
            fe.close()
        bins, demod = (NB, len(carriers)) if shape == "pfb256" else ((1600, 256) if shape == "grid1600" else (1600, 1600))
        keys = ("front_ends", "ok", "deadline_misses", "ring_overruns", "latency_ms_p50", "latency_ms_p99", "latency_ms_max",
                "gpu_busy_percent_est", "front_ends_per_group_block_mean", "host_plan_fraction_busiest_pump", "host_longest_plan_ms",
                "host_longest_device_wait_ms", "host_longest_sleep_overshoot_ms", "slow_plans_waits_sleeps", "host_cgroup", "pump_threads_sched_fifo", "pump_cpus", "host_cpus", "why_late", "errors",
                "seconds", "confirmation_run")
        out[shape] = {
            "K_max": good or 0, "K_max_first_attempt": first_attempt, "first_K_that_missed": bad,
            # the largest K whose run(s) all held every deadline AND kept the block latency's p99 under 5 ms (near the host
            # link's ceiling the queueing delay grows long before a deadline is missed)
            "K_max_p99_under_5ms": max([K_ for K_ in {p["front_ends"] for p in pts}
                                        if all(p.get("ok") and p.get("latency_ms_p99", 1e9) < 5.0
                                               for p in pts if p["front_ends"] == K_)] or [0]),
            "bins_per_front_end": bins, "demodulated_per_front_end": demod, "delivered_to_host_per_front_end": len(carriers) if shape == "pfb256" else 256,
            "channels_sustained": (good or 0) * bins, "fm_channels_sustained": (good or 0) * demod,
            "input_Msps_sustained": (good or 0) * FS / 1e6,
            "at_K_max": best, "points": [{k: p[k] for k in keys if k in p} for p in pts],
        }
    raw.free()
    return out
