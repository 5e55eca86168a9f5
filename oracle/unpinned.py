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
