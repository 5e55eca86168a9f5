"""Untimed GPU legs: the reference-shaped direct bank sweep and the reference-grid filterbanks."""
import time

import numpy as np

from .common import FS, HBM_PEAK_GBS, FP32_MATRIX_PEAK_TF
from .sustained import sustained_leg

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
    # every bin demodulated INSIDE the bank's kernel (rcf_pfb_fm_enable): no tap matrix, no IQ round trip
    # through HBM, no tap_finalize pass -- mode 2 writes the frame-major discriminator ring instead of the bins ring
    # (8 + 8 bytes per input sample), mode 1 beside it (8 + 16 + 8)
    fused = []
    for i in ids:
        fe.chan_close(i)
    ids = []
    if hasattr(fe, "pfb_fm_enable"):
        for mode, alg_b in ((2, 16.0), (1, 32.0)):
            try:
                fe.pfb_fm_enable(mode, gr_phase=True)
                for _ in range(60):
                    fe.commit(B)
                f_ms, fin_ms, f_wall = timed(50)
                fused.append({"mode": mode, "what": "discriminator ring only" if mode == 2 else "bins ring + discriminator ring",
                              "pfb_ms_per_block": f_ms, "tap_finalize_ms_per_block": fin_ms, "wall_ms_per_block": f_wall,
                              "algorithmic_bytes_per_launch": alg_b * B,
                              "frac_of_hbm_peak": alg_b * B / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "over_untapped_bank": f_ms / bank_ms,
                              "realtime_factor_at_20Msps": B / FS / (f_wall * 1e-3)})
            except Exception as e:  # noqa: BLE001 -- an untimed leg reports its own failure
                fused.append({"mode": mode, "error": "%s: %s" % (type(e).__name__, e)})
        fe.pfb_fm_enable(0)
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
        "fused_discriminator": {"kernel": "pfb5_fmlb_kernel<20,4,2,2>", "points": fused,
                                "two_kernel_path_ms_per_block": next((p["pfb_ms_per_block"] + p["tap_finalize_ms_per_block"]
                                                                      for p in tap_points if p.get("discriminator_only")), None),
                                "note": "every one of the 1600 bins demodulated in the bank's own launch (one chunk per "
                                        "workgroup, the frame before a chunk handed over wave to wave through L2): 8 + 8 bytes "
                                        "per input sample in mode 2 against the 8 + 16 of the bank plus the 16 + 8 of a "
                                        "tap_finalize pass behind it (two_kernel_path_ms_per_block)"},
    }
