"""How far can what the oracle CANNOT pin move the outputs?  (TEST INFRASTRUCTURE ONLY -- oracle/grspec.py header.)

The GR-block arithmetic of rows a4 / a6 / a8 is third-party (GNU Radio 3.8, VOLK, FFTW: absent from /root/reference
and from this image), so oracle/grspec.py restates it "parity unpinned".  Four details of that restatement depend on
how GNU Radio / VOLK were BUILT or on literals that are not on disk.  Each function here perturbs one of them by its
worst case and returns the movement of the judged quantity; tests/test_oracle_unpinned_bounds.py asserts the bars
still hold, tools/unpinned_bounds.py prints the table DESIGN.md 2 quotes.

  1. volk_32f_log2_32f: the SIMD kernels evaluate a polynomial (LOG_POLY_DEGREE 6 in VOLK's source: Fonseca's minimax
     fit, |error| ~3e-6 in log2 units; a degree-3 build would be ~1.5e-4), the generic kernel calls log2f.  Bound used:
     LOG2_ABS_ERR = 2e-4 per value, any sign pattern  ->  the 100-frame float32 running sum moves by <= 100 x that.
  2. gr::fast_atan2f's 257 table literals: reconstructed as atan(i/255) printed "%.6e"; each could differ in the last
     printed digit  ->  +-1 unit of the 7th significant digit on every entry (a superset of +-1 float32 ulp).
  3. the order in which the FIR's dot product is summed: VOLK picks a kernel at run time (generic: sequential; SSE /
     AVX: 4 / 8 lanes; AVX-512: 16), the GPU sums 16 x 4 MFMA partial products.
  4. gr::blocks::rotator's complex multiply: rounded product by product (x86-64 baseline build) or contracted into
     fused multiply-adds (-march=native / aarch64 builds).
"""
from __future__ import annotations

import math

import numpy as np

from . import cbind as OC
from . import grspec as G
from . import peaks as P

f32 = np.float32
LOG2_ABS_ERR = 2e-4


# ------------------------------------------------------------------------------------------- 1. log2 polynomial
def scan_logframes(x, N):
    """the U = len(x) / N distinct log-magnitude frames of a periodic scan stream (fft_vector.py:37-52), and the
    log2 values they were made from"""
    win = G.blackman_harris(N)
    frames = x[: len(x) // N * N].reshape(-1, N)
    X = G.fft_vcc_shift(frames, win)
    mag2 = (X.real.astype(f32) ** 2 + X.imag.astype(f32) ** 2).astype(f32)
    with np.errstate(divide="ignore"):
        l2 = np.log2(mag2).astype(f32)
    return np.where(np.isinf(l2), np.copysign(f32(127.0), l2), l2).astype(f32)


def spectrum_from_log2(l2, n_frames=1000, avg_len=100):
    scale = f32(1.0 / math.log2(10.0))
    v = ((l2 * scale).astype(f32) + f32(1.0)).astype(f32)            # nlog10_ff(1, N, 1)
    return G.scan_chain_periodic(list(v), n_frames, avg_len)


def peaks_under_log2_error(x, N, fs, fc, err=LOG2_ABS_ERR, seeds=(1, 2, 3)):
    """peak indices of the unperturbed chain and of the chain with every log2 value moved by up to +-err: random signs,
    all up, all down, alternating by bin, and the sign pattern that sharpens / flattens every local maximum"""
    l2 = scan_logframes(x, N)
    base = spectrum_from_log2(l2)
    ref, _ = P.peak_detect_scipy(base, fs, fc)
    k = np.arange(N)
    shape = np.sign(np.gradient(np.gradient(base.astype(np.float64))))      # curvature sign: pushes maxima up or down
    patterns = {"all_up": np.ones(N), "all_down": -np.ones(N), "alternating": np.where(k & 1, 1.0, -1.0),
                "sharpen": -shape, "flatten": shape}
    for s in seeds:
        patterns["random_%d" % s] = np.random.default_rng(s).uniform(-1, 1, size=l2.shape)
    out = {}
    worst_shift = 0.0
    for name, pat in patterns.items():
        pert = (l2 + (err * pat).astype(f32)).astype(f32)
        spec = spectrum_from_log2(pert)
        got, _ = P.peak_detect_scipy(spec, fs, fc)
        out[name] = list(map(int, got))
        worst_shift = max(worst_shift, float(np.max(np.abs(spec.astype(np.float64) - base))))
    return list(map(int, ref)), out, worst_shift


def log2_error_that_moves_a_peak(x, N, fs, fc, start=LOG2_ABS_ERR, stop=1.0):
    """smallest per-value log2 error (doubling from `start`) at which ANY of the patterns changes the index list"""
    e = start
    while e <= stop:
        ref, got, _ = peaks_under_log2_error(x, N, fs, fc, err=e, seeds=(1,))
        if any(v != ref for v in got.values()):
            return e
        e *= 2
    return None


# ------------------------------------------------------------------------------------------- 2. atan table literals
def fm_under_table_perturbation(y, gain, seeds=(1, 2, 3)):
    """max |delta fm| over the stream when every table entry moves by +-1 unit of its 7th significant digit"""
    base_tab = G.FAST_ATAN_TABLE.copy()
    ref = G.quadrature_demod_cf(y, gain)
    unit = np.array([10.0 ** (math.floor(math.log10(v)) - 6) if v > 0 else 1e-9 for v in base_tab.astype(np.float64)])
    worst = 0.0
    try:
        pats = [np.ones(257), -np.ones(257), np.where(np.arange(257) & 1, 1.0, -1.0)]
        pats += [np.random.default_rng(s).choice([-1.0, 1.0], size=257) for s in seeds]
        for pat in pats:
            G.FAST_ATAN_TABLE[:] = (base_tab.astype(np.float64) + pat * unit).astype(f32)
            got = G.quadrature_demod_cf(y, gain)
            worst = max(worst, float(np.max(np.abs(got.astype(np.float64) - ref))))
    finally:
        G.FAST_ATAN_TABLE[:] = base_tab
    return worst


# ------------------------------------------------------------------------------------------- 3. summation order
def iq_under_summation_orders(x, D, ctaps):
    """relative rms distance of every float32 summation order from the float64 sum, and the largest distance between
    two float32 orders"""
    v = {m: OC.fir_summation(x, D, ctaps, m) for m in (OC.SUM_8_LANES, OC.SUM_SEQUENTIAL, OC.SUM_PAIRWISE,
                                                       OC.SUM_16_LANES, OC.SUM_FLOAT64)}
    ref = v[OC.SUM_FLOAT64].astype(np.complex128)
    p = float(np.mean(np.abs(ref) ** 2))
    rel = lambda a, b: float(np.sqrt(np.mean(np.abs(a.astype(np.complex128) - b) ** 2) / p))
    names = {OC.SUM_8_LANES: "8_lanes", OC.SUM_SEQUENTIAL: "sequential", OC.SUM_PAIRWISE: "pairwise", OC.SUM_16_LANES: "16_lanes"}
    vs64 = {names[m]: rel(v[m], ref) for m in names}
    between = max(rel(v[a], v[b].astype(np.complex128)) for a in names for b in names if a < b)
    return vs64, between, v


# ------------------------------------------------------------------------------------------- 4. rotator contraction
def rotator_fma_drift(incr, n=1_000_000):
    """phases with and without FMA contraction over n outputs: max |delta phase| (the IQ stream's drift), and the max
    difference of the per-sample phase STEP (what the discriminator sees)"""
    a = OC.rotator_phases(incr, n, fma=False).astype(np.complex128)
    b = OC.rotator_phases(incr, n, fma=True).astype(np.complex128)
    drift = np.abs(a - b)
    step = lambda p: np.angle(p[1:] * np.conj(p[:-1]))
    dstep = np.abs(step(a) - step(b))
    return {"max_phase_difference": float(drift.max()), "phase_difference_at_end": float(drift[-1]),
            "max_step_difference_rad": float(dstep.max()),
            "magnitude_excursion_unfused": float(np.max(np.abs(np.abs(a) - 1))),
            "magnitude_excursion_fused": float(np.max(np.abs(np.abs(b) - 1)))}


# =========================================================================================== round 5: f-2 and the routing budget
# The analog voice chain (logging_receiver.py:211-222) adds three details the restatement cannot pin:
#   5. analog.fm_deemph's iir_filter_ffd: the ORDER in which one output's three products are accumulated (GNU Radio's
#      iir_filter.h: feed-forward taps first, then feedback; double accumulator) and whether a build keeps the accumulator
#      in double at all (iir_filter_ffd does; the restatement follows it) -- moved here to: feedback first, transposed
#      direct form II, and a float32 accumulator (the worst a differently-typed build could do);
#   6. gr-filter's pm_remez grid density (optfir passes 16) and its convergence: scipy's exchange at density 16 / 32 / 64;
#   7. the rational resampler's Kaiser taps: window and sinc evaluated in float (GNU Radio) or double, +-1 float32 ulp per tap.
# and the routing of frontend_mode = 'pfb':
#   8. rcf_pfb_tap_leakage predicts the discriminator error of a bank bin against GNU Radio's float32-phase channel from
#      |g|_2; the prediction must stay an upper bound (within its margin) when GNU Radio's phase arithmetic is perturbed the
#      way a different build could compute it (fwT0 kept in double, only the product rounded to float).
def _rms(v):
    return float(np.sqrt(np.mean(np.asarray(v, dtype=np.float64) ** 2)))


def deemph_forms(fm, rate, tau=75e-6):
    """the de-emphasised stream under four evaluation orders of the same first-order section; -> {form: float32 stream}"""
    from . import audio as A
    b, a = A.fm_deemph_taps(rate, tau)
    b0, b1, fb1 = float(b[0]), float(b[1]), -float(a[1])
    x = np.asarray(fm, dtype=f32)
    out = {"gnuradio_order_double": A.iir_filter_ffd(x, b, a)}
    y = np.empty(len(x), dtype=f32)
    px = py = 0.0
    for i, v in enumerate(x):                          # feedback product first
        acc = fb1 * py
        acc += b1 * px
        acc += b0 * float(v)
        py, px = acc, float(v)
        y[i] = f32(acc)
    out["feedback_first_double"] = y
    y = np.empty(len(x), dtype=f32)
    s = 0.0
    for i, v in enumerate(x):                          # transposed direct form II: y = b0 x + s; s = b1 x + fb1 y
        acc = b0 * float(v) + s
        s = b1 * float(v) + fb1 * acc
        y[i] = f32(acc)
    out["transposed_df2_double"] = y
    y = np.empty(len(x), dtype=f32)
    pxf = pyf = f32(0.0)
    fb0, fb1_, ffb = f32(b0), f32(b1), f32(fb1)
    for i, v in enumerate(x):                          # everything in float32 (NOT iir_filter_ffd: the worst case of a build)
        acc = f32(fb0 * v)
        acc = f32(acc + f32(fb1_ * pxf))
        acc = f32(acc + f32(ffb * pyf))
        pyf, pxf = acc, v
        y[i] = acc
    out["float32_accumulator"] = y
    return out


def audio_after_deemph(de, rate):
    """the rest of the voice chain behind the de-emphasis (audio low-pass, 300 Hz high-pass, 8 kHz resampler)"""
    from . import audio as A
    lpf = A.optfir_low_pass(8, rate, rate * 0.25, rate * 0.25 + 2000, 0.1, 60)
    hpf = A.high_pass(1, rate, 300, 30, A.WIN_HAMMING, 6.76)
    return A.rational_resampler_fff(A.fir_filter_fff(A.fir_filter_fff(de, lpf), hpf), 8000, int(rate))


def audio_under_deemph_forms(fm, rate):
    forms = deemph_forms(fm, rate)
    ref = audio_after_deemph(forms["gnuradio_order_double"], rate)
    scale = _rms(ref)
    return {k: _rms(audio_after_deemph(v, rate).astype(np.float64) - ref) for k, v in forms.items() if k != "gnuradio_order_double"}, scale


def remez_density_taps(rate, densities=(16, 32, 64)):
    """optfir.low_pass(8, rate, 0.25 rate, 0.25 rate + 2000, 0.1, 60) with the exchange run at several grid densities"""
    from scipy.signal import remez
    from . import audio as A
    r = 10.0 ** (0.1 / 20.0)
    n, fo, ao, w = A.remezord_lowpass(rate * 0.25, rate * 0.25 + 2000, (8, 0), [(r - 1.0) / (r + 1.0), 10.0 ** (-60 / 20.0)], rate)
    return {d: np.asarray(remez(n + 3, [f / 2.0 for f in fo], [ao[0], ao[2]], weight=w, type="bandpass", grid_density=d, fs=1.0)).astype(f32)
            for d in densities}


def audio_under_remez_density(de, rate):
    from . import audio as A
    taps = remez_density_taps(rate)
    hpf = A.high_pass(1, rate, 300, 30, A.WIN_HAMMING, 6.76)
    outs = {d: A.rational_resampler_fff(A.fir_filter_fff(A.fir_filter_fff(de, t), hpf), 8000, int(rate)) for d, t in taps.items()}
    ref = outs[16].astype(np.float64)
    return ({d: _rms(o.astype(np.float64) - ref) for d, o in outs.items() if d != 16},
            {d: float(np.max(np.abs(t.astype(np.float64) - taps[16]))) for d, t in taps.items() if d != 16}, _rms(ref))


def audio_under_resampler_tap_rounding(hp, rate, seeds=(1, 2, 3)):
    """the 8 kHz audio when every resampler tap moves by +-1 float32 ulp (random signs, all up, all down)"""
    from . import audio as A
    d = math.gcd(8000, int(rate))
    I, Dm = 8000 // d, int(rate) // d
    taps = A.design_resampler_taps(I, Dm)
    ref = A.rational_resampler_fff(hp, I, Dm, taps).astype(np.float64)
    ulp = np.spacing(np.abs(taps)).astype(f32)
    worst = 0.0
    pats = [np.ones(len(taps)), -np.ones(len(taps))] + [np.random.default_rng(s).choice([-1.0, 1.0], size=len(taps)) for s in seeds]
    for pat in pats:
        t = (taps + (pat * ulp).astype(f32)).astype(f32)
        worst = max(worst, _rms(A.rational_resampler_fff(hp, I, Dm, t).astype(np.float64) - ref))
    return worst, _rms(ref)


def bin_fm_error_vs_gr(x, fs, n_bins, taps, D, k, gain, phase_mode="float_product"):
    """rms discriminator difference between bank bin k (exact phases; the constant rotation a tap's rotator carries does
    not reach a discriminator) and GNU Radio's channel at the same offset -- whose tap phases are float32(i * fwT0)
    ('float_product': fwT0 = float(2 pi f0 / fs), what freq_xlating_fir_filter_ccc 3.8 computes) or float32(i * fwT0) with fwT0
    kept in double ('double_fwT0': what a build that declares it double would compute; the product of two floats rounded once
    in double IS the float product, so that variant is not a perturbation)"""
    f0 = (k if k < n_bins // 2 else k - n_bins) * fs / n_bins
    exact = G.xlating_fir_exact(x, D, taps, f0, fs).astype(np.complex64)
    fwT0 = f32(2.0 * math.pi * f0 / fs)
    i = np.arange(len(taps))
    if phase_mode == "float_product":
        ph = (i.astype(f32) * fwT0).astype(f32)
    elif phase_mode == "double_fwT0":                     # a build that keeps fwT0 in double and rounds only the product
        ph = (i.astype(np.float64) * (2.0 * math.pi * f0 / fs)).astype(f32)
    else:
        raise ValueError(phase_mode)
    ct = (np.asarray(taps, dtype=f32) * np.exp(1j * ph.astype(np.float64))).astype(np.complex64)
    a = f32(-fwT0 * f32(D))
    incr = np.complex64(complex(math.cos(float(a)), math.sin(float(a))))
    gr, _ = OC.channel_bank(x, D, ct[None, :], np.array([incr]), gains=[gain])
    n = min(len(exact), gr.shape[1])
    fe = G.quadrature_demod_cf(exact[:n], f32(gain))
    fg = G.quadrature_demod_cf(gr[0][:n], f32(gain))
    # GNU Radio's rotator turns by float32(-fwT0 D) per output instead of the exact angle: a constant in the discriminator
    # (carried by the tap's own rotator, rcf_pfb_tap_open(gr_phase)); what is compared is what is left
    d = fe[8:].astype(np.float64) - fg[8:]
    return _rms(d - np.mean(d))
