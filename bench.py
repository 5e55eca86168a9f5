#!/usr/bin/env python3
"""bench.py -- headline benchmark: BASELINE.json configs[1]

  256-channel polyphase filterbank over one 20 Msps synthetic IQ stream per MI355X, stage-2 xlating
  FIR (/3) + FM discriminator on 32 active bins.

One "step" = one commit of a BLOCK-sample batch that is already resident in HBM: PFB kernel (all 256 bins
written), 32 stage-2 FIRs with their discriminators fused in, history carry-over.  value = input IQ Msamples/s
over all ranks.  N > 1: one independent 20 Msps front-end per GPU (config_denver_massive_p25-style, BASELINE
configs[3]); weak scaling; no data-path collective -- the only RCCL traffic is the all-gather of detected-peak
lists (ncclAllGather through librcf's C ABI) after the timed region, reported as peaks_allgather_us.

No PyTorch: device work goes through librcf's C ABI (ctypes), ranks meet over rcf.multigpu.HostGroup (TCP next
to MASTER_PORT), the barrier on both sides of the timed region is `rcf_allreduce_max` (stream sync + one
ncclAllReduce), which also yields the max-over-ranks time.

Outside the timed region, rank 0 at N = 1 also reports:
  channels.direct_bank   the reference-shaped bank (one 2909-tap xlating FIR /800 + discriminator per channel)
                         actually opened and run at 256 .. 196608 (--sweep-max) channels, kernel ms per block
                         and TFLOP/s at each point
  channels.reference_grid_filterbank   the 1600-bin bank whose bins are the reference's channels (20 Msps, D = 800)
  scan                   BASELINE configs[2]: 1M-point FFT x 1000 frames / 100-frame average + peak pick
  end_to_end             PCIe-inclusive ingest (pinned cf32 rcf_push_iq, u8 rcf_push_raw), copy overlapped
  realtime               the paced leg: K independent 20 Msps u8 front-ends on this GPU fed at WALL-CLOCK rate in
                         20 ms blocks by native pump threads (rcf_pump_*), each driving a GROUP of front-ends
                         (rcf_group_*: one conversion / filterbank / stage-2 / gather launch for all blocks that are
                         complete), outputs in pinned host rings; K raised until a block misses its deadline -> K_max
                         (one attempt per point), channels sustained, per-block latency p50 / p99 / max, overruns; for the
                         256-bin + 32 FM shape and for the 1600-bin reference-grid bank with 256 bins demodulated
  control_plane          100 x create / release through the frontend_connector protocol
  cpu_baseline           the oracle's C port of the reference path on the host cores: one channel on one core, every
                         physical core busy (pinned, private first-touched streams, 2 s of signal per channel), SURVEY's
                         cores x single-core formula, and a time-tiled "best CPU" form (+ the parity check of the timed
                         configuration's FM outputs against the oracle)

--config cfg5 = BASELINE configs[4]'s per-GPU shape instead (25 Msps slice, 512-bin bank, N = 2^20 / 1000 / 100 scan on
the slice, <= 1024 peaks per rank into the all-gather); the default (cfg4) is configs[1] / configs[3].

`sustained`: the metric says "sustained" -- after the timed region the same commit loop runs for >= 2 s and reports the
filterbank launch time per window of 100 launches (first, last, slowest) with the shader clock read from sysfs.

Launch: python bench.py [--gpus N --steps K --warmup W].  For N > 1 either a launcher starts the ranks
        (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...: RANK / LOCAL_RANK /
        WORLD_SIZE / MASTER_* come from the environment), or -- when WORLD_SIZE is not set -- bench.py starts its own
        N rank processes (one per GPU, the way the reference starts one channelizer process per source:
        rc_frontend/receiver.py:67-70, systemd/radiocapture-channelizer@.service:11), waits for them and relays rank
        0's line.  Fewer than N visible devices is an error (rc 2), not a silent downgrade; RCF_BENCH_DEVICE=<d> puts
        every rank on device d (the N > 1 code on a 1-GPU box: host transport, RCCL cannot span one device twice).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "radiocapture-rf_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

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


# ------------------------------------------------------------------------------------------- CPU baseline leg
def cpu_baseline(tile, carriers, fm_check=None, signal_seconds=2.0, reps=3, FS=FS, chans_per_thread=2):
    """The reference's structure on the host cores: one 2909-tap xlating FIR (D=800) + discriminator per channel over
    the whole 20 Msps stream (rc_frontend/channel.py:31-38), oracle C port (oracle/rcf_oracle.c: ro_bank_bench, the same
    arithmetic as the oracle's channel bank -- tests/test_oracle_kat.py holds the two together bit for bit).  This leg
    is the only place bench.py touches oracle/: as the timed CPU baseline and as the checker of the GPU's FM outputs.

    SURVEY 8(d) asks for (i) one channel on one core and (ii) all cores busy; `signal_seconds` of signal per channel
    (a 0.25 s periodic tile walked 8 times), one thread pinned per PHYSICAL core, every thread on its own first-touched
    copy of the stream (the reference hands every channel flowgraph its own copy too: zeromq.pub_sink -> sub_source,
    channel.py:29), only the filtering timed.  Three all-core figures are reported and the GPU is compared with the
    LARGEST: the measured reference structure, SURVEY's formula cores x single-core rate, and a time-tiled form
    ("best CPU": blocks outer, the thread's channels inner, stream read once) that GNU Radio does not run."""
    from oracle import cbind as OC
    from oracle import grspec as G
    D, taps = G.channel_params(FS, 12500)
    n_tile = int(FS * 0.25) // D * D
    x = np.tile(tile, (n_tile + len(tile) - 1) // len(tile))[:n_tile]
    passes = max(1, int(round(signal_seconds * FS / n_tile)))
    signal_s = passes * n_tile / FS
    cores = OC.physical_cores()
    n_thr = max(1, len(cores)) if cores else max(1, OC.max_threads())
    cpu_ids = cores if cores else None
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
    # (i) a single channel on one core == one of the reference's per-channel GNU Radio flowgraphs
    t1 = med([OC.bank_bench(x, passes, D, ct[:1], inc[:1], gains[:1], 1, 1, cpu_ids)[0] for _ in range(reps)])
    # (ii) every physical core busy with `cpt` channels, the reference's structure (each channel walks the whole stream)
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
    rt_single = signal_s / t1
    rt_ref = n_ch * signal_s / t_ref
    rt_tiled = n_thr * cpt_t * (p_t * n_tile / FS) / t_tiled
    rt_formula = n_thr * rt_single
    traffic_ref = n_ch * passes * n_tile * 8.0 / t_ref
    largest = max(rt_ref, rt_tiled, rt_formula)
    out = {
        "value": signal_s * FS / t_ref / 1e6,
        "unit": "Msamples/s",
        "cores": n_thr,
        "cores_are": "physical cores, one pinned thread each (%d hardware threads on the box)" % OC.max_threads(),
        "kind": "port",
        "sample": "%.2f s of the same %g Msps synthetic stream per channel (a %.2f s periodic tile x %d), %d concurrent "
                  "12.5 kHz channels = %d per physical core (%d-tap xlating FIR /%d + discriminator each), median of %d; "
                  "CPU restatement of the reference's GNU Radio path (GNU Radio itself unavailable)"
                  % (signal_s, FS / 1e6, n_tile / FS, passes, n_ch, cpt, len(taps), D, reps),
        "channels": n_ch,
        "realtime_channels_at_20Msps": rt_ref,
        "single_channel_one_core": {"seconds_per_second_of_signal": t1 / signal_s, "Msamples_per_s": signal_s * FS / t1 / 1e6,
                                    "realtime_channels_per_core_at_20Msps": rt_single},
        "all_cores": {
            "reference_structure_measured": {"realtime_channels": rt_ref, "seconds": t_ref,
                                             "stream_read_GBps": traffic_ref / 1e9,
                                             "what": "channel outer, whole stream per channel: how GNU Radio runs it"},
            "survey_formula_cores_x_single_core": {"realtime_channels": rt_formula,
                                                   "what": "SURVEY 8(d): cores x 1 / per-channel real-time fraction"},
            "best_cpu_time_tiled_measured": {"realtime_channels": rt_tiled, "seconds": t_tiled,
                                             "channels": n_thr * cpt_t, "signal_seconds_per_channel": p_t * n_tile / FS,
                                             "what": "NOT the reference's structure: 64-output time blocks outer, the "
                                                     "thread's %d channels inner, stream read from DRAM once per thread" % cpt_t},
            "measured_over_formula": rt_ref / rt_formula,
            "host_read_bandwidth_GBps": bw / 1e9,
            "measured_vs_formula": (
                "measured all-core figure within %.0f %% of cores x single-core" % (100 * abs(1 - rt_ref / rt_formula))
                if rt_ref / rt_formula > 0.8 else
                "below the formula because every channel streams the whole wideband buffer (160 MB/s x its speed-up): %d "
                "channels at once read %.0f GB/s, %.0f %% of the %.0f GB/s the same pinned threads reach summing private "
                "buffers, while the single-core run has the memory system to itself; the time-tiled form (stream read "
                "once per thread) shows what is left when that is taken away"
                % (n_ch, traffic_ref / 1e9, 100 * traffic_ref / bw, bw / 1e9)),
        },
        "largest_cpu_realtime_channels": largest,
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


# ------------------------------------------------------------------------------------------- GPU legs (untimed)
def direct_bank_sweep(native, tile, device, counts, block=1 << 22):
    """The reference-shaped bank on this GPU: C channels of rcf_chan_open(12500, f) == channel.py:31-38 each
    (D = 800, T = 2909, GR-faithful float32 phases), opened for real, twelve blocks timed per point."""
    D, T = native.channel_params(FS, 12500)
    fd = native.Frontend(FS, 0.0, device=device, block_capacity=block, hist_capacity=1 << 16, out_capacity=1 << 13)
    for at in range(0, block, len(tile)):
        fd.ingest_write(tile[: min(len(tile), block - at)], at)
    fd.commit(block)
    ids, points = [], []
    block_s = block / FS
    for C_ in counts:
        t0 = time.perf_counter()
        while len(ids) < C_:
            k = len(ids)
            # 6.25 kHz raster across +-9.9 MHz, wrapped: distinct NCO phases, all inside the band
            f = ((k * 6250.0 + 9.9e6) % 19.8e6) - 9.9e6
            ids.append(fd.chan_open(12500, f))
        open_s = time.perf_counter() - t0
        fd.commit(block)                            # first block after opening: zero-history launch + bank pack
        fd.commit(block)
        fd.sync()
        fd.timing_enable(True, classes=[native.T_FIR, native.T_FIR_MFMA, native.T_DISC])
        for w in (native.T_FIR, native.T_FIR_MFMA, native.T_DISC):
            fd.timing_read(w)
        # wall clock per block in steady state: rcf_commit builds the block's launch records on the host (~0.5 us
        # per channel) and queues the kernels; the next commit's host work runs while they execute.  Twelve blocks,
        # one sync: (host + 12 x max(host, GPU)) / 12 -- the first block's host work is not hidden, so this is an
        # upper bound on the steady-state period
        n_timed = 12
        t0 = time.perf_counter()
        for _ in range(n_timed):
            fd.commit(block)
        fd.sync()
        wall = (time.perf_counter() - t0) / n_timed
        fms, fn = fd.timing_read(native.T_FIR)
        mms, mn = fd.timing_read(native.T_FIR_MFMA)
        dms, dn = fd.timing_read(native.T_DISC)
        fd.timing_enable(False)
        fir_ms = (fms + mms) / max(fn, mn, 1)
        per_block_s = (fir_ms + dms / max(dn, 1)) * 1e-3
        n_out = block // D
        tf = 8.0 * T * n_out * C_ / (fir_ms * 1e-3) / 1e12
        points.append({
            "channels": C_, "kernel_ms_per_block": per_block_s * 1e3, "fir_ms": fir_ms,
            "wall_ms_per_block": wall * 1e3, "block_ms_of_signal": block_s * 1e3,
            "real_time": bool(wall < block_s and per_block_s < block_s),
            "kernel": "fir_mfma_kernel (fp32 matrix cores)" if mn else "fir_bank_kernel (vector)",
            "tflops_fp32": tf, "frac_of_fp32_matrix_peak": tf / FP32_MATRIX_PEAK_TF,
            "realtime_channels_at_20Msps_extrapolated": C_ * block_s / per_block_s,
            "open_ms_per_channel": open_s * 1e3 / max(1, C_ - (points[-1]["channels"] if points else 0)),
        })
    fd.close()
    rt = [p["channels"] for p in points if p["real_time"]]
    return {"block_samples": block, "points": points,
            "channels_run_in_real_time": max(rt) if rt else 0,
            "note": "every count was opened and run (no extrapolation); flop = 8 T per output per channel; "
                    "peak 157.3 TF (datasheet) -- a bare v_mfma_f32_16x16x4_f32 loop with non-zero operands "
                    "sustains ~140 TF on this chip (tools/mfma_peak_probe.hip)"}


def reference_grid_leg(native, tile, device, B=1 << 25, n_taps=256):
    """The filterbank whose bins ARE the reference's channels: 1600 bins on the 12.5 kHz grid of one
    20 Msps front-end, built from channel.py's own filter (D = 800, T = 2909), every bin a 25 kS/s channel.
    Timed twice: the bank alone (that is what `roofline` is about), then with 256 bins tapped as channels with
    the discriminator (what frontend_mode = 'pfb' serves requests from).  Block 2^25 like the timed configuration:
    the two resident input buffers (2 x 268 MB) do not fit the 256 MB Infinity Cache -- at 2^24 they half do and
    the same kernel measures 20 % faster."""
    D, T = native.channel_params(FS, 12500)
    taps = native.design_low_pass_2(1.0, FS, 6250.0, 6250.0, 20.0)
    fe = native.Frontend(FS, 0.0, device=device, block_capacity=B, hist_capacity=1 << 16, out_capacity=1 << 17)
    fe.pfb_open(1600, D, taps)
    for _ in range(2):
        for at in range(0, B, len(tile)):
            fe.ingest_write(tile[: min(len(tile), B - at)], at)
        fe.commit(B)

    def timed(n=10):
        fe.commit(B)
        fe.sync()
        fe.timing_enable(True, classes=[native.T_PFB, native.T_TAPS])
        fe.timing_read(native.T_PFB)
        fe.timing_read(native.T_TAPS)
        t0 = time.perf_counter()
        for _ in range(n):
            fe.commit(B)
        fe.sync()
        wall = (time.perf_counter() - t0) / n
        pfb_ms, pn = fe.timing_read(native.T_PFB)
        disc_ms, dn = fe.timing_read(native.T_TAPS)
        fe.timing_enable(False)
        return pfb_ms / max(pn, 1), disc_ms / max(dn, 1), wall * 1e3

    for _ in range(300):                               # ~50 ms of work: the launch time settles after ~15 ms (see `sustained`)
        fe.commit(B)
    bank_ms, _, bank_wall = timed(100)
    sustained = sustained_leg(fe, native, B, 24.0 * B)
    tap_points = []
    ids = []
    for n_t, fm_only in ((n_taps, False), (1600, False), (1600, True)):
        for i in ids:
            fe.chan_close(i)
        # 256 scattered bins (all through the tap matrix), then every bin once (all read from the bank's ring), then every bin
        # as a channel that is only demodulated (rcf_chan_set_fm_only: the discriminator ring alone is written)
        ids = [fe.pfb_tap_open((7 + 6 * i) % 1600 if n_t < 1600 else i, gr_phase=True) for i in range(n_t)]
        if fm_only:
            if not hasattr(fe, "chan_set_fm_only"):
                continue
            for i in ids:
                fe.chan_set_fm_only(i, True)
        for _ in range(60):                            # steady state again (opening 1600 taps idled the queue)
            fe.commit(B)
        tap_ms, fin_ms, tap_wall = timed(50)
        assert fe.chan_produced(ids[0]) > 0
        tap_points.append({"bins_tapped": n_t, "discriminator_only": fm_only, "pfb_ms_per_block": tap_ms, "tap_finalize_ms_per_block": fin_ms,
                           "wall_ms_per_block": tap_wall, "realtime_factor_at_20Msps": B / FS / (tap_wall * 1e-3),
                           "pfb_over_untapped": tap_ms / bank_ms})
    fe.close()
    # the 6.25 kHz grid: 3200 bins, rings of 3.4 GB -- (a) the reference's own 6.25 kHz channel,
    # channel.py:31-35 at cr = 6250: D = 1600, T = 5819; (b) its 12.5 kHz channel filter on the finer raster, D = 800
    fine = []
    for cr, label in ((6250, "channel.py rule at cr = 6250: every bin == one 6.25 kHz reference channel at 12.5 kS/s"),
                      (12500, "the 12.5 kHz channel filter on the 6.25 kHz raster (oversampled x4), 25 kS/s per bin")):
        D2, T2 = native.channel_params(FS, cr)
        taps2 = native.design_low_pass_2(1.0, FS, cr / 2.0, cr / 2.0, 20.0)
        fe = native.Frontend(FS, 0.0, device=device, block_capacity=B, hist_capacity=1 << 16,
                             out_capacity=1 << (16 if D2 == 1600 else 17))
        fe.pfb_open(3200, D2, taps2)
        for _ in range(2):
            for at in range(0, B, len(tile)):
                fe.ingest_write(tile[: min(len(tile), B - at)], at)
            fe.commit(B)
        for _ in range(150):                           # ~40 ms of work before the timed hundred, as for the 1600-bin bank
            fe.commit(B)
        ms, _, wall = timed(100)
        fe.close()
        alg2 = (8.0 + 8.0 * 3200 / D2) * B
        fine.append({"bins": 3200, "decim": D2, "taps": T2, "what": label, "block_samples": B, "pfb_ms_per_block": ms,
                     "wall_ms_per_block": wall, "algorithmic_bytes_per_launch": alg2,
                     "frac_of_hbm_peak": alg2 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
    alg = 24.0 * B                                    # 8 B read + 8 * 1600 / 800 B written per input sample
    return {
        "workload": "1600-bin filterbank, decim 800, 2909-tap channel.py prototype (every bin == one reference "
                    "channel at 25 kS/s), 20 Msps cf32, block %d" % B,
        "kernel": "pfb5_kernel<20,4,2,2>", "pfb_ms_per_block": bank_ms, "wall_ms_per_block": bank_wall,
        "input_Msamples_per_s_kernel": B / (bank_ms * 1e-3) / 1e6,
        "realtime_factor_at_20Msps": B / FS / (bank_wall * 1e-3),
        "reference_channels_per_frontend": 1600,
        "roofline": {"bound": "hbm", "algorithmic_bytes_per_launch": alg, "achieved": alg / (bank_ms * 1e-3) / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (bank_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        "sustained": sustained,
        "grid_6k25": fine,
        "with_taps": {"note": "tapped bins leave the bank's kernel as a compact frame-major matrix (whole rows) -- except "
                              "complete aligned runs of 16 bins, which are read from the bank's own ring; "
                              "tap_finalize_kernel transposes either into the channels' rings with GNU Radio's rotator per "
                              "tap and the discriminator fused in (pfb_ms = the bank incl. the matrix, tap_finalize_ms = "
                              "that kernel).  Points: 256 scattered bins (matrix), all 1600 bins (ring)",
                      "points": tap_points},
    }


def scan_leg(native, synth, device):
    """BASELINE configs[2]: 1M-point FFT, 1000 frames, 100-frame average (fft_vector.py:31-60) at 100 Msps from a
    16-frame periodic resident buffer, then the device peak pick (fft_peak_detection.py:38-73)."""
    N, F, L, fs = 1 << 20, 1000, 100, 100e6
    rng = np.random.default_rng(3003)                # 12 carriers, bin centres >= 5000 bins apart
    centres = [40000 + 80000 * i + int(rng.integers(-3000, 3000)) for i in range(12)]
    carriers = [(c, float(rng.uniform(4000, 9000)), 45.0) for c in centres]
    x = synth.scan_stream(fs, N, 16, carriers, seed=3003)
    B = 16 * N
    fe = native.Frontend(fs, 0.0, device=device, block_capacity=B, hist_capacity=N, out_capacity=1 << 10)
    for _ in range(2):
        fe.ingest_write(x, 0)
        fe.commit(B)
    fe.sync()
    res = {}
    for rep in range(4):                            # the last pass counts: buffers warm, launch times settled (~15 ms of work)
        fe.timing_enable(True, classes=[native.T_SCAN_FFT, native.T_SCAN_MOVSUM])
        fe.timing_read(native.T_SCAN_FFT)
        fe.timing_read(native.T_SCAN_MOVSUM)
        fe.scan_start(N, F, L)
        t0 = time.perf_counter()
        while fe.scan_frames_done() < F:
            fe.commit(B)
        fe.sync()
        wall = time.perf_counter() - t0
        fft_ms, _ = fe.timing_read(native.T_SCAN_FFT)
        mov_ms, _ = fe.timing_read(native.T_SCAN_MOVSUM)
        fe.timing_enable(False)
        t0 = time.perf_counter()
        idx, mean, _ = fe.scan_find_peaks(cap=1024)
        pick_ms = (time.perf_counter() - t0) * 1e3
        samples = float(N) * F
        res = {
            "workload": "BASELINE configs[2]: N=2^20, 1000 frames, 100-frame average, 100 Msps, 12 carriers",
            "fft_logmag_ms": fft_ms, "moving_sum_ms": mov_ms, "peak_pick_ms_incl_readback": pick_ms,
            "wall_ms": wall * 1e3, "peaks_found": int(len(idx)),
            "input_Msamples_per_s": samples / ((fft_ms + mov_ms) * 1e-3) / 1e6,
            "realtime_factor_at_100Msps": samples / fs / ((fft_ms + mov_ms) * 1e-3),
            "roofline": {"bound": "hbm", "algorithmic_bytes": 12.0 * samples,
                         "achieved": 12.0 * samples / ((fft_ms + mov_ms) * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s",
                         "frac": 12.0 * samples / ((fft_ms + mov_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "ceiling_note": SCAN_CEILING_NOTE},
        }
    fe.close()
    return res


def end_to_end_leg(native, tile, device, B=1 << 24):
    """PCIe-inclusive ingest with the filterbank and the FM channels running: pinned host buffers handed to
    rcf_push_iq (cf32, 8 B/sample) and rcf_push_raw (u8, 2 B/sample); block n+1 is copied while block n runs."""
    fe = native.Frontend(FS, 0.0, device=device, block_capacity=B, hist_capacity=1 << 16, out_capacity=1 << 18)
    fe.pfb_open(NB, NB, proto_taps(native))
    out = {"block_samples": B}
    pin = native.PinnedArray(B, np.complex64)
    pin.array[:] = np.tile(tile, B // len(tile))
    for _ in range(2):
        fe.push(pin.array)
    fe.sync()
    t0 = time.perf_counter()
    for _ in range(8):
        fe.push(pin.array)
    fe.sync()
    out["pinned_cf32_push_iq_Msps"] = 8 * B / (time.perf_counter() - t0) / 1e6
    pin.free()
    raw = native.PinnedArray(2 * B, np.uint8)
    raw.array[:] = np.tile((np.clip(np.round(tile.view(np.float32) * 32 + 127.4), 0, 255)).astype(np.uint8),
                           B // len(tile))
    for _ in range(2):
        fe.push_raw(raw.array, native.FMT_U8, 1.0 / 128, 127.4)
    fe.sync()
    t0 = time.perf_counter()
    for _ in range(8):
        fe.push_raw(raw.array, native.FMT_U8, 1.0 / 128, 127.4)
    fe.sync()
    out["pinned_u8_push_raw_Msps"] = 8 * B / (time.perf_counter() - t0) / 1e6
    raw.free()
    fe.close()
    out["note"] = "host -> HBM over PCIe Gen5 x16 (63 GB/s spec) + 256-bin PFB per block; never `value`"
    return out


def control_plane_leg(device):
    """100 x create / release through the reference's client (frontend_connector.py:242-251 times exactly this)."""
    import types
    from rcf import frontend_connector as FC, protocol, receiver

    class OneChannelizer:
        def get_channelizer_for_frequency(self, f):
            return ("127.0.0.1", 0)

    cfg = types.SimpleNamespace(sources={0: dict(type="synthetic", center_freq=855000000, samp_rate=20000000)},
                                frontend_mode="xlat")
    tb = receiver.receiver(cfg, device=device)
    srv = protocol.FrontendServer(tb)
    fc = FC.frontend_connector("bench", OneChannelizer(), heartbeat=False,
                               transport_factory=lambda h, p: protocol.LoopbackTransport(srv))
    t_create, t_release = [], []
    for i in range(100):
        t0 = time.perf_counter()
        cid, port = fc.create_channel(12500, int(855e6 + 12500 * (i - 50)))
        t1 = time.perf_counter()
        assert cid, "create_channel failed"
        fc.release_channel()
        t2 = time.perf_counter()
        t_create.append(t1 - t0)
        t_release.append(t2 - t1)
    # and 100 distinct channels held at once (no idle reuse): what a busy trunked system asks for
    t0 = time.perf_counter()
    held = [tb.connect_channel(12500, int(855e6 + 12500 * (i - 50)))[0] for i in range(100)]
    t_hold = (time.perf_counter() - t0) / 100
    for b in held:
        tb.release_channel(b)
    tb.sweep_idle_channels(now=time.time() + 60)
    tb.close()
    return {"n": 100, "create_ms_median": sorted(t_create)[50] * 1e3, "create_ms_max": max(t_create) * 1e3,
            "release_ms_median": sorted(t_release)[50] * 1e3,
            "connect_channel_new_ms_mean": t_hold * 1e3,
            "note": "create = connect_channel (reuses an idle channel after the first, as receiver.py:311-319 does) "
                    "+ protocol; channel buffers come from the handle's slab pool"}


# ------------------------------------------------------------------------------------------- live PMC traffic
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
    child = [sys.executable, os.path.abspath(__file__), "--steps", "5", "--warmup", "1", "--block", str(block), "--no-extras",
             "--no-cpu-baseline", "--no-sustained", "--no-live-traffic", "--prewarm-seconds", "0.2"]
    if cfg5:
        child += ["--config", "cfg5"]
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="rcf_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
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


# ------------------------------------------------------------------------------------------- grouped launches, resident
def group_capacity_leg(native, tile, carriers, device, G=80, blk=409600, seconds=1.5):
    """What ONE grouped launch per stage is worth when the GPU is kept busy: G front-ends (each the BASELINE configs[1]
    shape: 256-bin bank + 32 FM channels) with a real-time-sized block resident in HBM, committed back to back as one
    group block (rcf_group_commit: one records + history launch, ONE filterbank launch over all G x 100 chunks, ONE stage-2
    launch over all G x 32 channels), against the same G front-ends committed one after the other.  The paced real-time
    leg runs the same launches at a duty cycle of a few per cent (the chip idles between group blocks and clocks down:
    its launches are slower than these)."""
    fes, ids = [], []
    for i in range(G):
        fe = native.Frontend(FS, 0.0, device=device, block_capacity=blk, hist_capacity=1 << 14, out_capacity=1 << 12)
        fe.pfb_open(NB, NB, proto_taps(native))
        ids.append([fe.pfb_chan_open(c["bin"] % NB, 12500, c["delta"]) for c in carriers])
        fes.append(fe)
    x = np.tile(tile, blk // len(tile) + 1)[:blk]
    out = {"front_ends": G, "block_samples": blk, "samples_per_group_block": G * blk,
           "algorithmic_bytes_per_filterbank_launch": 16.0 * G * blk}
    for mode in ("one_by_one", "grouped"):
        grp = native.Group(fes) if mode == "grouped" else None
        for _ in range(2):                               # both ping-pong buffers of every member hold data
            if grp is not None:
                grp.push([np.roll(x, 977 * i) for i in range(G)], native.FMT_CF32)
            else:
                for i, fe in enumerate(fes):
                    fe.push(np.roll(x, 977 * i))
        step = (lambda: grp.commit([blk] * G)) if grp is not None else (lambda: [fe.commit(blk) for fe in fes])
        sync = grp.sync if grp is not None else (lambda: [fe.sync() for fe in fes])
        for _ in range(30):
            step()
        sync()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < seconds / 2:
            step()
            n += 1
        sync()
        wall = (time.perf_counter() - t0) / n
        e = {"wall_ms_per_group_block": wall * 1e3, "input_Msps": G * blk / wall / 1e6, "group_blocks": n}
        # the filterbank launch itself, HIP events on the launch stream (the grouped launches are timed on member 0)
        fes[0].timing_enable(True, classes=[native.T_PFB, native.T_FIR_DERIVED])
        fes[0].timing_read(native.T_PFB)
        fes[0].timing_read(native.T_FIR_DERIVED)
        for _ in range(100):
            step()
        sync()
        ms, k = fes[0].timing_read(native.T_PFB)
        ms2, k2 = fes[0].timing_read(native.T_FIR_DERIVED)
        fes[0].timing_enable(False)
        if k:
            per = (16.0 * G * blk) if grp is not None else 16.0 * blk
            e["filterbank_launch_us"] = ms / k * 1e3
            e["filterbank_launches_timed"] = k
            e["filterbank_frac_of_hbm_peak"] = per / (ms / k * 1e-3) / 1e9 / HBM_PEAK_GBS
        if k2:
            e["stage2_launch_us"] = ms2 / k2 * 1e3
        out[mode] = e
        if grp is not None:
            # the grouped outputs are the one-by-one outputs (bit for bit: tests/test_gpu_group.py); here only that every
            # channel produced the same count either way
            e["outputs_per_channel"] = fes[G // 2].chan_produced(ids[G // 2][0])
            grp.close()
    out["grouped_over_one_by_one"] = out["one_by_one"]["wall_ms_per_group_block"] / out["grouped"]["wall_ms_per_group_block"]
    for fe in fes:
        fe.close()
    return out


# ------------------------------------------------------------------------------------------- paced real-time leg
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
        else:                                            # the bank whose bins ARE the reference's channels + 256 of them tapped
            D, T = native.channel_params(FS, 12500)
            fe = native.Frontend(FS, 0.0, device=device, block_capacity=blk, hist_capacity=1 << 14, out_capacity=1 << 11)
            fe.pfb_open(1600, D, native.design_low_pass_2(1.0, FS, 6250.0, 6250.0, 20.0))
            ids = [fe.pfb_tap_open((7 + 6 * j) % 1600, gr_phase=True) for j in range(256)]
        fes.append(fe)
        chans.append(ids)
    fes, chans = fes[:K], chans[:K]
    produced0 = sum(fes[i].chan_produced(c) for i in range(K) for c in chans[i])
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
                                     spin_us=int(os.environ.get("RCF_BENCH_RT_SPIN_US", "0"))))   # (spinning the idle waits: measured WORSE -- 40 ms device stalls in both 10 s runs, none with sleeps)
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
    produced = sum(fes[i].chan_produced(c) for i in range(K) for c in chans[i]) - produced0
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
    bins tapped and demodulated."""
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
        if shape == "grid1600":
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
        bins, demod = (NB, len(carriers)) if shape == "pfb256" else (1600, 256)
        keys = ("front_ends", "ok", "deadline_misses", "ring_overruns", "latency_ms_p50", "latency_ms_p99", "latency_ms_max",
                "gpu_busy_percent_est", "front_ends_per_group_block_mean", "host_plan_fraction_busiest_pump", "host_longest_plan_ms",
                "host_longest_device_wait_ms", "host_longest_sleep_overshoot_ms", "slow_plans_waits_sleeps", "host_cgroup", "pump_threads_sched_fifo", "errors",
                "seconds", "confirmation_run")
        out[shape] = {
            "K_max": good or 0, "K_max_first_attempt": first_attempt, "first_K_that_missed": bad,
            # the largest K whose run(s) all held every deadline AND kept the block latency's p99 under 5 ms (near the host
            # link's ceiling the queueing delay grows long before a deadline is missed)
            "K_max_p99_under_5ms": max([K_ for K_ in {p["front_ends"] for p in pts}
                                        if all(p.get("ok") and p.get("latency_ms_p99", 1e9) < 5.0
                                               for p in pts if p["front_ends"] == K_)] or [0]),
            "bins_per_front_end": bins, "demodulated_per_front_end": demod,
            "channels_sustained": (good or 0) * bins, "fm_channels_sustained": (good or 0) * demod,
            "input_Msps_sustained": (good or 0) * FS / 1e6,
            "at_K_max": best, "points": [{k: p[k] for k in keys if k in p} for p in pts],
        }
    raw.free()
    return out


# ------------------------------------------------------------------------------------------------ rank launcher
def pin_to_gpu_numa(local_rank):
    """N > 1: keep this rank's threads on the CPUs of its GPU's NUMA node (the pump threads of a per-GPU real-time leg and
    the pinned buffers they touch then sit next to the GPU's PCIe root).  The AMD render nodes in PCI order are HIP's
    device order; node -1 (no NUMA information) or any failure: no pinning.  -> {numa_node, cpus} for the line."""
    try:
        import glob
        nodes = []
        for d in sorted(glob.glob("/sys/class/drm/renderD*/device")):
            try:
                if open(os.path.join(d, "vendor")).read().strip() != "0x1002":
                    continue
                nodes.append((os.path.basename(os.path.realpath(d)), int(open(os.path.join(d, "numa_node")).read())))
            except Exception:
                continue
        nodes.sort()
        dev = int(os.environ.get("RCF_BENCH_DEVICE", local_rank))
        if dev >= len(nodes) or nodes[dev][1] < 0:
            return {"numa_node": None, "cpus": len(os.sched_getaffinity(0)), "pinned": False}
        node = nodes[dev][1]
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return {"numa_node": node, "cpus": len(os.sched_getaffinity(0)), "pinned": False}
        os.sched_setaffinity(0, cpus)
        return {"numa_node": node, "cpus": len(cpus), "pinned": True, "pci": nodes[dev][0]}
    except Exception as e:
        return {"numa_node": None, "cpus": None, "pinned": False, "error": "%s: %s" % (type(e).__name__, e)}


def load_native():
    """librcf's ctypes layer.  RCF_BENCH_NATIVE=<module> swaps in another module with the same surface: the CPU test of
    the launcher (tests/test_bench_launcher.py) runs the whole N-rank protocol over a stub Frontend that way."""
    name = os.environ.get("RCF_BENCH_NATIVE")
    if name:
        import importlib
        return importlib.import_module(name)
    from rcf import native
    return native


def spawn_ranks(n, argv, timeout_s=3600.0):
    """--gpus n > 1 and no launcher around us: start n rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set, HIP_VISIBLE_DEVICES untouched, device = local rank), relay rank 0's stdout, return
    the first non-zero exit code (the other ranks are then terminated by pid)."""
    import socket
    import subprocess
    native = load_native()
    have = native.device_count()
    if have < n and "RCF_BENCH_DEVICE" not in os.environ:
        print("bench.py: --gpus %d but %d HIP device(s) visible: refusing to measure fewer GPUs than asked for "
              "(RCF_BENCH_DEVICE=<d> runs every rank on device d)" % (n, have), file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RCF_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    deadline = time.time() + timeout_s
    rc, out0 = 0, b""
    pending = set(range(n))
    try:
        while pending and rc == 0:
            for r in sorted(pending):
                if r == 0:
                    try:                               # drain rank 0's pipe while waiting (its line can be > 64 KB)
                        o, _ = procs[0].communicate(timeout=0.2)
                        out0 += o or b""
                    except subprocess.TimeoutExpired:
                        continue
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0:
                    rc = code
                    print("bench.py: rank %d exited with %d" % (r, code), file=sys.stderr)
                    break
            if time.time() > deadline:
                rc = 124
                print("bench.py: ranks still running after %.0f s" % timeout_s, file=sys.stderr)
            time.sleep(0.05)
    finally:
        for r in pending:
            if procs[r].poll() is None:
                procs[r].terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    if rc == 0 and not out0.strip():
        print("bench.py: rank 0 printed no line", file=sys.stderr)
        rc = 1
    return rc


# ------------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--block", type=int, default=1 << 25, help="samples per step (resident batch)")
    ap.add_argument("--config", choices=["cfg4", "cfg5"], default="cfg4",
                    help="cfg4: BASELINE configs[1] per GPU (x N = configs[3]); cfg5: BASELINE configs[4]'s per-GPU shape")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed legs (sweep, scan, end-to-end, ...)")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 2 s sustained leg")
    ap.add_argument("--sustained-seconds", type=float, default=2.0,
                    help="length of the sustained leg of the timed configuration (profiles/rNN_sustained_60s.json: 60)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not run the two rocprofv3 --pmc child passes that measure roofline.traffic in this run")
    ap.add_argument("--alone-warm", type=int, default=300,
                    help="untimed commits before the filterbank-alone pass (steady state: the chip's ramp after an idle queue is ~15 ms)")
    ap.add_argument("--alone-launches", type=int, default=100, help="timed launches of the filterbank-alone pass")
    ap.add_argument("--time-every", type=int, default=4,
                    help="HIP events around every n-th filterbank launch of the timed region (1 = all of them)")
    ap.add_argument("--prewarm-seconds", type=float, default=0.5,
                    help="untimed commits before the warm-up steps: the metric is SUSTAINED throughput, and the chip "
                         "needs ~100 launches (15 ms) after idling before the filterbank launch settles (the first "
                         "window of the sustained leg shows it); 0 = none")
    ap.add_argument("--sweep-max", type=int, default=196608, help="largest direct-bank channel count to open")
    ap.add_argument("--rt-seconds", type=float, default=10.0, help="seconds per point of the paced real-time leg (0 = skip it)")
    ap.add_argument("--rt-block-ms", type=float, default=20.0, help="block length of the paced real-time leg")
    ap.add_argument("--rt-k-first", type=int, default=512, help="front-end count the real-time search starts at")
    ap.add_argument("--rt-k-cap", type=int, default=1280, help="largest front-end count the real-time search tries")
    ap.add_argument("--rt-k-per-gpu", type=int, default=512, help="N > 1: front-ends of the one paced real-time point every rank runs")
    ap.add_argument("--rt-pumps", type=int, default=0, help="native pump threads (groups) of the real-time leg (0: four)")
    ap.add_argument("--rt-window-ms", type=float, default=1.0, help="batching window of the pumps: a complete block waits this long for company")
    ap.add_argument("--rt-shapes", default="pfb256,grid1600", help="shapes of the real-time leg")
    ap.add_argument("--rt-burst", action="store_true",
                    help="real-time leg: every front-end's block completes at the same instant (default: spread over the period)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "RCF_BENCH_DEVICE" in os.environ:             # two ranks on ONE GPU: exercises the N > 1 code on a 1-GPU box
        local_rank = int(os.environ["RCF_BENCH_DEVICE"])
    n_gpus = world if world > 1 else 1
    if args.gpus != n_gpus and rank == 0:
        print("note: --gpus %d but WORLD_SIZE=%d; using %d" % (args.gpus, world, n_gpus), file=sys.stderr)

    # N > 1: this rank's threads stay on the CPUs of its GPU's NUMA node (before the HIP runtime starts its own threads)
    numa = pin_to_gpu_numa(local_rank) if world > 1 and os.environ.get("RCF_BENCH_NO_PIN", "0") in ("", "0") else None
    # (the host driver only supports dmabuf IPC: without this RCCL's peer mappings fail -- set before the HIP runtime loads)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from rcf import multigpu, synth
    native = load_native()
    if native.device_count() < 1:
        raise RuntimeError("bench.py needs an MI355X (no HIP device visible)")
    if native.device_count() <= local_rank:
        raise RuntimeError("rank %d: device %d asked for, %d visible" % (rank, local_rank, native.device_count()))

    cfg5 = args.config == "cfg5"
    # cfg5 (BASELINE configs[4]): 8 spectrum slices of 25 Msps, a 512-bin bank each = 4096 channels at 200 Msps
    # aggregate; every rank scans its slice (N = 2^20, 1000 frames, 100-frame average: fft_vector.py:31-60) and
    # contributes <= 1024 peaks (fft_peak_detection.py:38-73) to the all-gather
    fs, nb, n_active = (25e6, 512, 0) if cfg5 else (FS, NB, N_ACTIVE)
    SCAN_N, SCAN_F, SCAN_L = 1 << 20, 1000, 100
    B = args.block
    assert B % nb == 0 and (not cfg5 or B % SCAN_N == 0)
    frames = B // nb
    out_cap = 1
    while out_cap < 2 * frames + 64:                  # two blocks of frames + the stage-2 channels' reach: the stage-2 launch of
                                                      # block n rides in block n + 1's filterbank launch (rcf_set_stage2_lag)
        out_cap <<= 1
    fe = native.Frontend(fs, 0.0, device=local_rank, block_capacity=B, hist_capacity=SCAN_N if cfg5 else 1 << 16,
                         out_capacity=out_cap)
    group = None
    use_rccl = os.environ.get("RCF_BENCH_TRANSPORT", "rccl") == "rccl"
    rccl_ranks, rccl_proof = 0, None
    if world > 1:
        group = multigpu.HostGroup(rank, world, os.environ.get("MASTER_ADDR", "127.0.0.1"),
                                   int(os.environ.get("MASTER_PORT", "29500")) + 101)
        # a communicator that never comes up (a rank missing, a fabric problem) must not hang the run for an hour: if the
        # join + proof below are not through in RCF_BENCH_RCCL_TIMEOUT seconds (default 300) this rank says so and exits
        import threading
        rccl_done = threading.Event()

        def _watchdog(limit=float(os.environ.get("RCF_BENCH_RCCL_TIMEOUT", "300"))):
            if not rccl_done.wait(limit):
                print("bench.py: rank %d: communicator set-up / proof not finished after %.0f s -- giving up "
                      "(RCF_BENCH_TRANSPORT=host runs without RCCL)" % (rank, limit), file=sys.stderr, flush=True)
                os._exit(3)
        threading.Thread(target=_watchdog, daemon=True).start()
        if use_rccl:
            # ncclCommInitRank on this rank's GPU.  If any rank cannot (no librccl, two ranks told to share one
            # GPU, ...) every rank falls back to the host rendezvous for the barrier and the gather -- the data
            # path has no collective, so the measurement itself does not depend on it.
            uid = None
            if group.rank == 0 and "RCF_BENCH_DEVICE" not in os.environ:
                try:
                    uid = native.comm_unique_id()
                except Exception as e:
                    print("note: RCCL unavailable on rank 0 (%s): host transport" % e, file=sys.stderr)
            uid = group.broadcast(uid if uid is not None else b"")
            ok = len(uid) == 128
            if ok:
                try:
                    fe.comm_init(group.rank, group.world, uid)
                except Exception as e:
                    print("note: rank %d could not join the RCCL communicator (%s)" % (rank, e), file=sys.stderr)
                    ok = False
            use_rccl = all(p == b"1" for p in group.all_gather(b"1" if ok else b"0"))
            if not use_rccl:
                fe.comm_destroy()
        if use_rccl:
            # prove the communicator BEFORE anything is timed: one ncclAllGather of the rank numbers and one
            # ncclAllReduce(max) over all `world` GPUs; a run whose RCCL ring does not work fails here, loudly
            rccl_ranks = fe.comm_size()
            parts = fe.allgather_peaks(np.array([rank], dtype=np.int64), multigpu.PEAK_CAP)   # the capacity the real gather uses
            seen = [int(p[0]) if len(p) else -1 for p in parts]
            top = fe.allreduce_max(float(rank))
            if rccl_ranks != world or seen != list(range(world)) or top != float(world - 1):
                raise RuntimeError("RCCL proof failed on rank %d: comm size %d of %d, all-gather %s, all-reduce max %s"
                                   % (rank, rccl_ranks, world, seen, top))
            rccl_proof = {"allgather_of_rank_numbers": seen, "allreduce_max_of_rank_numbers": top,
                          "when": "before the warm-up steps"}
        rccl_done.set()
    taps = proto_taps(native, fs, nb)
    fe.pfb_open(nb, nb, taps)
    if cfg5:
        # the slice's stream: a 16-frame periodic tile with 12 scan-shaped carriers (SURVEY 8(d) cfg3's recipe at
        # 25 Msps: occupied widths 4-9 kHz = 170-380 bins of 23.8 Hz, inside find_peaks' [126, 1258] window)
        rng = np.random.default_rng(5000 + rank)
        centres = [40000 + 80000 * i + int(rng.integers(-3000, 3000)) for i in range(12)]
        scan_carriers = [(c, float(rng.uniform(4000, 9000)), 45.0) for c in centres]
        tile = synth.scan_stream(fs, SCAN_N, 16, scan_carriers, seed=5000 + rank)
        meta = {"carriers": [{"f_off": (k - nb // 2 + 0.5) * fs / nb * 0.9} for k in range(0, nb, nb // 32)]}
        chans = []
    else:
        tile, meta = synth.cfg2(n=1 << 20, seed=2002 if n_gpus == 1 else 4000 + rank, n_bins=nb,
                                n_active=n_active)
        chans = [fe.pfb_chan_open(c["bin"] % nb, 12500, c["delta"]) for c in meta["carriers"]]

    # make the batch resident in both ping-pong buffers (not timed: "inputs already resident in HBM")
    for _ in range(2):
        for at in range(0, B, len(tile)):
            fe.ingest_write(tile[: min(len(tile), B - at)], at)
        fe.commit(B)
    fe.sync()

    def barrier_max(v=0.0):
        """barrier + device sync on every rank, max of v over ranks"""
        fe.sync()
        if group is None:
            return v
        return fe.allreduce_max(v) if use_rccl else group.max(v)

    tp = time.perf_counter()
    n_prewarm = 0
    while time.perf_counter() - tp < args.prewarm_seconds:
        for _ in range(32):
            fe.commit(B)
        n_prewarm += 32
    # HIP events only on the kernel the roofline reports, and only on every 4th launch of it.  The two events are ATTACHED
    # to the filterbank's dispatch (hipExtLaunchKernelGGL), not recorded around it: one barrier packet less inside the
    # measured interval (bracket 102.9 us, attached 101.1-102.2 on one box; RCF_TIMING_BRACKET=1 keeps the bracket;
    # rocprofv3's kernel trace reads another 2.5-5 us less for the same launches).  A timed launch costs the step ~4 us
    # (all 20 timed: 0.1232 instead of 0.1193 ms), hence every 4th.  Switched on BEFORE the warm-up steps, so that
    # nothing but the barrier and one counter reset lies between them and the timed region.
    fe.timing_enable(True, classes=[native.T_PFB])
    time_every = args.time_every if args.steps >= 2 * args.time_every else 1     # a short run times every launch
    fe.timing_stride(time_every)
    for _ in range(args.warmup):
        fe.commit(B)
    barrier_max()
    fe.timing_read(native.T_PFB, reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fe.commit(B)
    fe.sync()
    t1 = time.perf_counter()
    elapsed = barrier_max(t1 - t0)
    import struct
    per_rank_ms = [(t1 - t0) / args.steps * 1e3] if group is None else \
        [struct.unpack("<d", p)[0] / args.steps * 1e3 for p in group.all_gather(struct.pack("<d", t1 - t0))]

    pfb_ms, pfb_n = fe.timing_read(native.T_PFB)
    fe.timing_stride(1)
    # ... and a second pass, NOT timed by the wall clock, in which EVERY launch of the same number of steps (at least 20)
    # carries its two events: the large-sample launch time beside the every-4th one of the timed region
    n_all = max(args.steps, 20)
    for _ in range(n_all):
        fe.commit(B)
    fe.sync()
    pfb_all_ms, pfb_all_n = fe.timing_read(native.T_PFB)
    # the FM channels' newest outputs, for the parity check against the oracle (done in the cpu_baseline leg)
    fm_check = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline and chans:
        fm_check = {"taps": taps, "total_in": fe.samples_in, "carriers": meta["carriers"],
                    "fm": [fe.chan_read_fm(c, 1.0, max_samples=out_cap) for c in chans]}
    # per-kernel breakdown of the other launches: a few extra, untimed steps with every class instrumented
    fe.timing_enable(True)
    n_extra = min(args.steps, 5)
    for _ in range(n_extra):
        fe.commit(B)
    fe.sync()
    fe.timing_read(native.T_PFB)
    fir2_ms, _ = fe.timing_read(native.T_FIR_DERIVED)
    disc_ms, _ = fe.timing_read(native.T_DISC)
    hist_ms, _ = fe.timing_read(native.T_HISTORY)
    fe.timing_enable(False)
    if chans:
        assert fe.chan_produced(chans[0]) > 0        # the FM channels really produced output

    alg_bytes_pfb = 16.0 * B                          # 8 B read + 8 B written per input sample (critically sampled)
    # the stage-2 work of the previous block rides in the filterbank's launch (rcf_set_stage2_lag: 256-bin kernel only):
    # the launch then also moves that work's algorithmic bytes -- SURVEY 8(d): per active bin 8 B read per frame of its
    # stream, 8 B (IQ) + 4 B (fused discriminator) written per output at a third of the frame rate
    s2_rides = bool(chans) and nb == 256 and os.environ.get("RCF_S2_LAG", "1") != "0" and hasattr(fe, "set_stage2_lag")
    alg_bytes_s2 = len(chans) * (8.0 * (B // nb) + 12.0 * ((B // nb) // 3)) if s2_rides else 0.0
    alg_bytes = alg_bytes_pfb + alg_bytes_s2
    # ... and the filterbank kernel ALONE (the lag switched off for a pass of its own: every launch timed) -- the figure
    # that compares with the rounds before the rider existed
    pfb_alone_ms = pfb_alone_n = None
    if s2_rides:
        fe.set_stage2_lag(False)
        # (the read-backs and timing reads above idled the queue: the chip needs ~15 ms of work before its launch time
        # settles -- `sustained`: first window of 100 launches 125 us, the rest 113-115 -- and up to round 5's first
        # profiles this pass sat on that ramp: 107-108 us where tools/pfb_probe.py, 500 launches in, measures 97-100)
        for _ in range(args.alone_warm):
            fe.commit(B)
        fe.timing_enable(True, classes=[native.T_PFB])
        fe.timing_read(native.T_PFB)
        for _ in range(max(args.steps, args.alone_launches)):
            fe.commit(B)
        fe.sync()
        pfb_alone_ms, pfb_alone_n = fe.timing_read(native.T_PFB)
        fe.timing_enable(False)
        fe.set_stage2_lag(True)
    sustained = None
    if not args.no_sustained:                         # every rank runs it (the ranks stay in step); rank 0 reports
        sustained = sustained_leg(fe, native, B, alg_bytes, seconds=args.sustained_seconds)
        if group is not None:
            sustained["kernel_us_last_window_max_over_ranks"] = barrier_max(sustained["kernel_us_last_window"])

    # ---- scan of the rank's slice (cfg5) and the peak-list all-gather, outside the timed region
    allgather_us, gathered_n, scan_out = None, None, None
    freqs = []
    if cfg5:
        fe.timing_enable(True, classes=[native.T_SCAN_FFT, native.T_SCAN_MOVSUM])
        fe.timing_read(native.T_SCAN_FFT)
        fe.timing_read(native.T_SCAN_MOVSUM)
        fe.scan_start(SCAN_N, SCAN_F, SCAN_L)
        ts = time.perf_counter()
        while fe.scan_frames_done() < SCAN_F:
            fe.commit(B)                              # the bank keeps running: scan and channelizer share the stream
        fe.sync()
        scan_wall = time.perf_counter() - ts
        fft_ms, _ = fe.timing_read(native.T_SCAN_FFT)
        mov_ms, _ = fe.timing_read(native.T_SCAN_MOVSUM)
        fe.timing_enable(False)
        tp = time.perf_counter()
        idx, _, _ = fe.scan_find_peaks(cap=1024)
        pick_ms = (time.perf_counter() - tp) * 1e3
        centre = 851e6 + fs * rank                    # slice g is centred fs * g above the first
        freqs = [native.peak_frequency(int(i), fs, SCAN_N, centre) for i in idx]
        scan_out = {"workload": "N=2^20, 1000 frames, 100-frame average over the rank's 25 Msps slice, 12 carriers",
                    "fft_logmag_ms": fft_ms, "moving_sum_ms": mov_ms, "peak_pick_ms_incl_readback": pick_ms,
                    "wall_ms_with_the_bank_running": scan_wall * 1e3, "peaks_found_rank0": int(len(idx)),
                    "scan_ms_max_over_ranks": barrier_max(fft_ms + mov_ms),
                    "realtime_factor_at_25Msps": float(SCAN_N) * SCAN_F / fs / ((fft_ms + mov_ms) * 1e-3)}
    if group is not None:
        if not cfg5:
            fe.scan_start(16384, 8, 4)
            fe.commit(B)
            idx, _, _ = fe.scan_find_peaks(cap=1024)
            freqs = [native.peak_frequency(int(i), fs, 16384, 851e6 + 25e6 * rank) for i in idx]
            if not freqs:                            # the filterbank tile has no scan-shaped carriers: exchange its
                freqs = [int(851e6 + 25e6 * rank + c["f_off"]) for c in meta["carriers"]]   # 32 known ones instead
        gather = (lambda: multigpu.allgather_peaks(fe, freqs)) if use_rccl else \
                 (lambda: multigpu.allgather_peaks_host(group, freqs))
        gather()                                     # warm-up (RCCL ring setup)
        barrier_max()
        ta = time.perf_counter()
        everyone = gather()
        allgather_us = barrier_max((time.perf_counter() - ta) * 1e6)
        gathered_n = len(everyone)
    pfb_avg_ms_max = barrier_max(pfb_ms / max(pfb_n, 1)) if group is not None else pfb_ms / max(pfb_n, 1)
    by_rank = None
    if group is not None:
        # what the line says about EVERY rank, not only the slowest: launch time / roofline fraction, the sustained leg's
        # last window, the rank's peak count (their sum must be what the gather returned), its NUMA pinning -- and, with
        # --rt-seconds > 0, one paced real-time point per GPU (K = --rt-k-per-gpu front-ends on every rank at the same time)
        mine = {"rank": rank, "avg_launch_ms": pfb_ms / max(pfb_n, 1),
                "frac": alg_bytes / (pfb_ms / max(pfb_n, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS if pfb_ms > 0 else None,
                "sustained_frac_last_window": sustained["frac_last_window"] if sustained else None,
                "peaks": len(freqs), "numa": numa}
        if args.rt_seconds > 0 and not args.no_extras:
            try:
                barrier_max()
                blk = int(round(FS * args.rt_block_ms * 1e-3))
                raw = native.PinnedArray(2 * blk * 2 * args.rt_k_per_gpu, np.uint8)
                t8 = np.clip(np.round(tile.view(np.float32) * 32 + 127.4), 0, 255).astype(np.uint8)
                for b_ in range(2 * args.rt_k_per_gpu):
                    at = 2 * ((b_ * 40961) % (len(tile) - blk))
                    raw.array[2 * blk * b_: 2 * blk * (b_ + 1)] = t8[at: at + 2 * blk]
                pool = {"fes": [], "chans": []}
                p = realtime_point(native, pool, args.rt_k_per_gpu, "pfb256", {"array": raw.array},
                                   meta["carriers"] if not cfg5 else synth.cfg2(n=1 << 16, seed=2002)[1]["carriers"], local_rank,
                                   min(4.0, args.rt_seconds), args.rt_block_ms, args.rt_pumps or 4, not args.rt_burst, args.rt_window_ms)
                for f_ in pool["fes"]:
                    f_.close()
                raw.free()
                mine["realtime"] = {k_: p[k_] for k_ in ("front_ends", "ok", "deadline_misses", "ring_overruns", "latency_ms_p50",
                                                         "latency_ms_p99", "latency_ms_max", "output_samples_lost", "errors")}
            except Exception as e:
                mine["realtime"] = {"front_ends": args.rt_k_per_gpu, "ok": False, "errors": ["%s: %s" % (type(e).__name__, e)]}
        by_rank = sorted((json.loads(b.decode("utf-8")) for b in group.all_gather(json.dumps(mine).encode("utf-8"))),
                         key=lambda r_: r_["rank"])

    if rank == 0:
        total_samples = float(B) * args.steps * n_gpus
        value = total_samples / elapsed / 1e6
        avg_pfb_s = (pfb_ms / max(pfb_n, 1)) * 1e-3
        achieved = alg_bytes / avg_pfb_s / 1e9 if avg_pfb_s > 0 else 0.0
        traffic, traffic_src = None, None
        live = None
        if n_gpus == 1 and not args.no_live_traffic and B == 1 << 25:
            fe.sync()
            live = measure_traffic_live(cfg5, B, "pfb_kernel_2b<256" if cfg5 else "pfb_kernel_os<256")
        tname = "pfb512_traffic.json" if cfg5 else "pfb_traffic.json"
        tpath = os.path.join(ROOT, "profiles", tname)
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get("block") == B:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_src = "profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of %s (not measured " \
                                  "in this run)" % (tname, tj.get("measured", "an earlier run of this configuration"))
            except Exception:
                traffic = None
        traffic_file, traffic_file_src = traffic, traffic_src
        if live is not None:
            traffic = live["hbm_bytes_per_launch"]
            traffic_src = ("measured in this run on this box: two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE; "
                           "--kernel-trace only) over `bench.py --steps 5 --no-extras --no-cpu-baseline --no-sustained`, "
                           "%d dispatches of the kernel averaged; KiB x 1024, FETCH x 2 (gfx950, MI355X_MICROARCH.md)"
                           % live["dispatches_averaged"])
        if cfg5:
            workload = ("BASELINE configs[4], per-GPU shape: 512-bin critically-sampled PFB (6981-tap prototype) over "
                        "one 25 Msps cf32 spectrum slice per GPU (x8 = 4096 channels at 200 Msps), N=2^20 scan of the "
                        "slice + <=1024 peaks per rank into the all-gather outside the timed region")
            kname = "pfb_kernel_2b<256, 14, 3, false> (512 bins: 256-thread workgroups, two branches per thread)"
        else:
            workload = ("BASELINE configs[1]: 256-bin critically-sampled PFB (3491-tap prototype) over one "
                        "20 Msps cf32 stream per GPU, stage-2 xlating FIR /3 + FM discriminator on 32 active bins")
            kname = "pfb_kernel_os<256,1,14,4,false>"
        out = {
            "metric": "input IQ Msamples/s + concurrent 12.5 kHz FM channels sustained",
            "value": value,
            "unit": "Msamples/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "prewarm": {"seconds": args.prewarm_seconds, "untimed_steps": n_prewarm,
                        "why": "steady state before the W warm-up steps (metric: sustained); see `sustained`"},
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_by_rank": per_rank_ms,
            "by_rank": by_rank,
            "ranks_started_by": ("bench.py itself (one process per GPU)" if os.environ.get("RCF_BENCH_SPAWNED")
                                 else "the launcher's environment (RANK / WORLD_SIZE)") if world > 1 else "single process",
            "rccl_ranks": rccl_ranks if world > 1 else 1,
            "transport": ("rccl" if use_rccl else "host-tcp") if world > 1 else "none (one rank)",
            "rccl_proof": rccl_proof,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "samp_rate": fs, "pfb_bins": nb, "fm_channels_per_gpu": n_active,
                "block_samples": B, "parallelism": "1 front-end per GPU x%d" % n_gpus,
            },
            "channels": {"pfb_bins_total": nb * n_gpus, "fm_demod_total": n_active * n_gpus,
                         "realtime_factor_at_%dMsps" % int(fs / 1e6): value / n_gpus / (fs / 1e6)},
            "roofline": {
                "bound": "hbm", "kernel": kname,
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_src,
                "traffic_fetch_x2_bytes": live["fetch_bytes_corrected_x2"] if live else None,
                "traffic_write_bytes": live["write_bytes"] if live else None,
                "traffic_from_tracked_file": traffic_file if live is not None else None,
                "algorithmic_bytes_per_launch": alg_bytes,
                "algorithmic_bytes_filterbank": alg_bytes_pfb, "algorithmic_bytes_stage2_rider": alg_bytes_s2,
                "stage2_rides_in_this_launch": s2_rides,
                "filterbank_alone": ({"avg_launch_ms": pfb_alone_ms / max(pfb_alone_n, 1), "launches": pfb_alone_n,
                                      "frac": alg_bytes_pfb / (pfb_alone_ms / max(pfb_alone_n, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                      "what": "the same kernel without the rider (rcf_set_stage2_lag off for a pass of its own, every "
                                              "launch timed): 16 B x block / launch time, the figure of the rounds before the rider"}
                                     if pfb_alone_n else None),
                "avg_launch_ms": avg_pfb_s * 1e3, "launches": pfb_n, "timed_every": time_every,
                "avg_launch_ms_every_launch_pass": pfb_all_ms / max(pfb_all_n, 1), "launches_every_launch_pass": pfb_all_n,
                "frac_every_launch_pass": alg_bytes / (pfb_all_ms / max(pfb_all_n, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS if pfb_all_ms > 0 else None,
                "every_launch_pass_note": "a second, untimed pass of max(steps, 20) commits right behind the timed region with "
                                          "events on EVERY filterbank launch (each costs the step ~4 us, which is why the timed "
                                          "region itself times every 4th)",
                "timed_how": "bracket of two hipEventRecord (RCF_TIMING_BRACKET)" if os.environ.get("RCF_TIMING_BRACKET", "0") not in ("", "0")
                else "HIP events attached to the kernel's dispatch (hipExtLaunchKernelGGL start / stop events), on the launch stream",
                "avg_launch_ms_slowest_rank": pfb_avg_ms_max,
                "frac_slowest_rank": alg_bytes / (pfb_avg_ms_max * 1e-3) / 1e9 / HBM_PEAK_GBS if pfb_avg_ms_max > 0 else 0.0,
            },
            "kernel_ms_per_step": {
                "pfb": pfb_ms / max(pfb_n, 1),
                "stage2_fir_with_fused_discriminator": fir2_ms / max(n_extra, 1),
                "stage2_note": ("rides in the NEXT block's filterbank launch (its first workgroups): no launch of its own in steady "
                                "state -- the figure above is the one flush the timing read forced, spread over the steps") if s2_rides else None,
                "separate_discriminator_launches": disc_ms / max(n_extra, 1),
                "launch_records_and_history_copy": hist_ms / max(n_extra, 1),
                "launch_records_and_history_copy_note": "0 = no launch of its own: the filterbank kernel's first workgroups "
                                                        "do both copies on the way in (PfbLaunch::rider_*)",
            },
        }
        if by_rank is not None:
            out["roofline"]["frac_by_rank"] = [r_["frac"] for r_ in by_rank]
            out["numa_by_rank"] = [r_["numa"] for r_ in by_rank]
            if any("realtime" in r_ for r_ in by_rank):
                rts = [r_.get("realtime", {"ok": False}) for r_ in by_rank]
                out["realtime_per_gpu"] = {
                    "what": "one paced point per GPU, all GPUs at the same time: K 20 Msps u8 front-ends per GPU (256-bin bank + 32 "
                            "FM channels each) in %.0f ms blocks through native pumps; no search at N > 1" % args.rt_block_ms,
                    "front_ends_per_gpu": args.rt_k_per_gpu, "ok_by_rank": [bool(r_.get("ok")) for r_ in rts],
                    "latency_ms_p99_by_rank": [r_.get("latency_ms_p99") for r_ in rts],
                    "deadline_misses_by_rank": [r_.get("deadline_misses") for r_ in rts],
                    "front_ends_sustained_total": sum(args.rt_k_per_gpu for r_ in rts if r_.get("ok")),
                    "fm_channels_sustained_total": sum(args.rt_k_per_gpu * 32 for r_ in rts if r_.get("ok")),
                    "input_Msps_sustained_total": sum(args.rt_k_per_gpu * FS / 1e6 for r_ in rts if r_.get("ok")),
                    "errors": [e_ for r_ in rts for e_ in (r_.get("errors") or [])]}
        if sustained is not None:
            if by_rank is not None:
                sustained["frac_last_window_by_rank"] = [r_["sustained_frac_last_window"] for r_ in by_rank]
            out["sustained"] = sustained
        if scan_out is not None:
            out["scan"] = scan_out
        if allgather_us is not None:
            out["peaks_allgather_us"] = allgather_us
            out["peaks_allgather"] = {"transport": "ncclAllGather via rcf_allgather_peaks" if use_rccl else "host TCP",
                                      "values_gathered": gathered_n, "ranks": world,
                                      "peaks_by_rank": [r_["peaks"] for r_ in by_rank] if by_rank else None,
                                      "values_expected": sum(r_["peaks"] for r_ in by_rank) if by_rank else None,
                                      "ok": (gathered_n == sum(r_["peaks"] for r_ in by_rank)) if by_rank else None,
                                      "peaks_from": "N=2^20 scan of each rank's slice" if cfg5 else
                                                    "16384-point scan / the tile's known carriers"}
    fe.close()
    if group is not None:
        group.close()
    if rank != 0:
        return

    extras = n_gpus == 1 and not args.no_extras and not cfg5
    if extras:
        # the bandwidth-bound legs first: seconds of matrix-core work at full power (the sweep) leave the chip at
        # lower clocks for whatever runs next (measured: the same filterbank launch 0.112 ms before it, 0.139 after)
        def leg(fn, *a_, **k_):                             # a leg outside the timed region must not cost the run its line
            try:
                return fn(*a_, **k_)
            except Exception as e:
                return {"error": "%s: %s" % (type(e).__name__, e)}
        out["channels"]["reference_grid_filterbank"] = leg(reference_grid_leg, native, tile, local_rank)
        out["scan"] = leg(scan_leg, native, synth, local_rank)
        out["end_to_end"] = leg(end_to_end_leg, native, tile, local_rank)
        try:
            out["group_capacity"] = group_capacity_leg(native, tile, meta["carriers"], local_rank)
        except Exception as e:
            out["group_capacity"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if args.rt_seconds > 0:
            try:
                out["realtime"] = realtime_leg(native, tile, meta["carriers"], local_rank, seconds=args.rt_seconds,
                                               block_ms=args.rt_block_ms, k_first=args.rt_k_first, k_cap=args.rt_k_cap,
                                               stagger=not args.rt_burst, n_pumps=args.rt_pumps, window_ms=args.rt_window_ms,
                                               shapes=tuple(x for x in args.rt_shapes.split(",") if x))
            except Exception as e:                       # a leg outside the timed region must not cost the run its line
                out["realtime"] = {"error": "%s: %s" % (type(e).__name__, e)}
        out["control_plane"] = leg(control_plane_leg, local_rank)
        counts = [c for c in (256, 1024, 4096, 16384, 65536, 131072, 196608) if c <= args.sweep_max]
        out["channels"]["direct_bank"] = leg(direct_bank_sweep, native, tile, local_rank, counts)
    if n_gpus == 1 and not args.no_cpu_baseline:
        if cfg5:
            # same structure on the slice's stream: the reference would run one 3637-tap xlating FIR /1000 per channel
            out["cpu_baseline"] = cpu_baseline(tile[: 1 << 20], meta["carriers"], None, FS=fs)
        else:
            out["cpu_baseline"] = cpu_baseline(tile, meta["carriers"], fm_check)
        db = out["channels"].get("direct_bank")
        if db and "channels_run_in_real_time" in db:
            # against the LARGEST of the CPU figures (measured reference structure, SURVEY's formula, time-tiled best CPU)
            cb = out["cpu_baseline"]
            cb["gpu_channels_run_in_real_time_over_cpu_realtime_channels"] = (
                db["channels_run_in_real_time"] / cb["largest_cpu_realtime_channels"])
            cb["gpu_over_cpu_note"] = ("%d reference-shaped channels opened and run in real time on one MI355X (direct "
                                       "bank) / %.0f, the largest CPU figure above" % (db["channels_run_in_real_time"],
                                                                                      cb["largest_cpu_realtime_channels"]))
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))


if __name__ == "__main__":
    main()
