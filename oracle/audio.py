"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's analog NBFM voice chain.

/root/reference/logging_receiver.py:211-222 (and file_to_wav.py:45-51,109-122) builds, per analog call,

    pwr_squelch_cc(-100, 0.01, 0, True)
      -> fm_demod_cf(channel_rate=rate, audio_decim=1, deviation=15000, audio_pass=0.25 rate,
                     audio_stop=0.25 rate + 2000, gain=8, tau=75e-6)
      -> fir_filter_fff(1, firdes.high_pass(1, rate, 300, 30, WIN_HAMMING, 6.76))
      -> rational_resampler_fff(interpolation=8000, decimation=rate)

Every block is GNU Radio 3.8 (absent here, PARITY UNPINNED): the functions below restate the published block
semantics ("[GR-spec]"), each naming the GR source it follows.  The one live third-party piece is the
Parks-McClellan exchange itself: GR's optfir.low_pass calls pm_remez(order, bands, ampl, weights, "bandpass",
grid_density=16); scipy.signal.remez is the same algorithm (both descend from the McClellan-Parks-Rabiner
program) and stands in for it -- the optimum is unique, implementations differ at the convergence tolerance.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math

import numpy as np

from . import grspec as G

f32 = np.float32

WIN_HAMMING, WIN_HANN, WIN_BLACKMAN, WIN_RECTANGULAR, WIN_KAISER, WIN_BLACKMAN_HARRIS = 0, 1, 2, 3, 4, 5


# ------------------------------------------------------------------ gr-fft window.cc
def izero(x: float) -> float:
    """window.cc Izero(): power series of the modified Bessel function, IzeroEPSILON = 1e-21."""
    s = u = 1.0
    n = 1
    halfx = x / 2.0
    while True:
        temp = halfx / float(n)
        n += 1
        temp *= temp
        u *= temp
        s += u
        if not (u >= 1e-21 * s):
            return s


def kaiser(ntaps: int, beta: float) -> np.ndarray:
    ibeta = 1.0 / izero(beta)
    inm1 = 1.0 / float(ntaps - 1)
    w = np.empty(ntaps, dtype=f32)
    for i in range(ntaps):
        temp = 2 * i * inm1 - 1
        w[i] = f32(izero(beta * math.sqrt(1.0 - temp * temp)) * ibeta)
    return w


def max_attenuation(wintype: int, beta: float = 6.76) -> float:
    return {WIN_HAMMING: 53.0, WIN_HANN: 44.0, WIN_BLACKMAN: 74.0, WIN_RECTANGULAR: 21.0,
            WIN_KAISER: beta / 0.1102 + 8.7, WIN_BLACKMAN_HARRIS: 92.0}[wintype]


def window(wintype: int, ntaps: int, beta: float = 6.76) -> np.ndarray:
    if wintype == WIN_KAISER:
        return kaiser(ntaps, beta)
    return G.window(wintype, ntaps)


def compute_ntaps(fs: float, tw: float, wintype: int, beta: float = 6.76) -> int:
    """firdes.cc compute_ntaps()"""
    n = int(max_attenuation(wintype, beta) * fs / (22.0 * tw))
    return n if n & 1 else n + 1


# ------------------------------------------------------------------ gr-filter firdes.cc
def low_pass(gain, fs, fc, tw, wintype=WIN_HAMMING, beta=6.76) -> np.ndarray:
    ntaps = compute_ntaps(fs, tw, wintype, beta)
    w = window(wintype, ntaps, beta)
    M = (ntaps - 1) // 2
    fwT0 = 2.0 * math.pi * fc / fs
    taps = np.empty(ntaps, dtype=f32)
    for n in range(-M, M + 1):
        if n == 0:
            taps[n + M] = f32(fwT0 / math.pi * float(w[n + M]))
        else:
            taps[n + M] = f32(math.sin(n * fwT0) / (n * math.pi) * float(w[n + M]))
    fmax = float(taps[M])
    for n in range(1, M + 1):
        fmax += 2.0 * float(taps[n + M])
    g = gain / fmax
    return (taps.astype(np.float64) * g).astype(f32)


def high_pass(gain, fs, fc, tw, wintype=WIN_HAMMING, beta=6.76) -> np.ndarray:
    ntaps = compute_ntaps(fs, tw, wintype, beta)
    w = window(wintype, ntaps, beta)
    M = (ntaps - 1) // 2
    fwT0 = 2.0 * math.pi * fc / fs
    taps = np.empty(ntaps, dtype=f32)
    for n in range(-M, M + 1):
        if n == 0:
            taps[n + M] = f32((1.0 - (fwT0 / math.pi)) * float(w[n + M]))
        else:
            taps[n + M] = f32(-math.sin(n * fwT0) / (n * math.pi) * float(w[n + M]))
    fmax = float(taps[M])
    for n in range(1, M + 1):
        fmax += 2.0 * float(taps[n + M]) * math.cos(n * math.pi)
    g = gain / fmax
    return (taps.astype(np.float64) * g).astype(f32)


# ------------------------------------------------------------------ gr-filter python optfir.py
def lporder(freq1, freq2, delta_p, delta_s) -> float:
    df = abs(freq2 - freq1)
    ddp = math.log10(delta_p)
    dds = math.log10(delta_s)
    a1, a2, a3, a4, a5, a6 = 5.309e-3, 7.114e-2, -4.761e-1, -2.66e-3, -5.941e-1, -4.278e-1
    b1, b2 = 11.01217, 0.5124401
    t1 = a1 * ddp * ddp
    t2 = a2 * ddp
    t3 = a4 * ddp * ddp
    t4 = a5 * ddp
    dinf = ((t1 + t2 + a3) * dds) + (t3 + t4 + a6)
    ff = b1 + b2 * (ddp - dds)
    return dinf / df - ff * df + 1


def remezord_lowpass(f1, f2, mags, devs, fsamp):
    """optfir.remezord() for the two-band (low-pass) case -> (order, band edges in [0,1], ampls, weights)"""
    fcuts = [float(f1) / fsamp, float(f2) / fsamp]
    devs = list(devs)
    for i in range(len(mags)):
        if mags[i] != 0:
            devs[i] = devs[i] / mags[i]
    n = int(math.ceil(lporder(fcuts[0], fcuts[1], devs[0], devs[1]))) - 1
    ff = [0.0, 2 * fcuts[0], 2 * fcuts[1], 1.0]
    aa = [mags[0], mags[0], mags[1], mags[1]]
    max_dev = max(devs)
    wts = [max_dev / d for d in devs]
    return n, ff, aa, wts


def optfir_low_pass(gain, fs, freq1, freq2, passband_ripple_db, stopband_atten_db, nextra_taps=2) -> np.ndarray:
    from scipy.signal import remez
    r = 10.0 ** (passband_ripple_db / 20.0)
    passband_dev = (r - 1.0) / (r + 1.0)
    stopband_dev = 10.0 ** (-stopband_atten_db / 20.0)
    n, fo, ao, w = remezord_lowpass(freq1, freq2, (gain, 0), [passband_dev, stopband_dev], fs)
    taps = remez(n + nextra_taps + 1, [f / 2.0 for f in fo], [ao[0], ao[2]], weight=w, type="bandpass",
                 grid_density=16, fs=1.0)
    return np.asarray(taps, dtype=np.float64).astype(f32)       # pm_remez returns doubles, fir_filter_fff takes floats


# ------------------------------------------------------------------ gr-analog python fm_emph.py
def fm_deemph_taps(fs, tau=75e-6):
    w_c = 1.0 / tau
    w_ca = 2.0 * fs * math.tan(w_c / (2.0 * fs))
    k = -w_ca / (2.0 * fs)
    z1 = -1.0
    p1 = (1.0 + k) / (1.0 - k)
    b0 = -k / (1.0 - k)
    return [b0 * 1.0, b0 * -z1], [1.0, -p1]


def iir_filter_ffd(x: np.ndarray, btaps, ataps) -> np.ndarray:
    """iir_filter<float,float,double,double>, oldstyle=False: y[n] = b0 x[n] + b1 x[n-1] - a1 y[n-1], double
    accumulator and double output history, result cast to float."""
    b0, b1 = float(btaps[0]), float(btaps[1])
    fb1 = -float(ataps[1])
    y = np.empty(len(x), dtype=f32)
    px, py = 0.0, 0.0
    for i, v in enumerate(np.asarray(x, dtype=f32)):
        acc = b0 * float(v)
        acc += b1 * px
        acc += fb1 * py
        py = acc
        px = float(v)
        y[i] = f32(acc)
    return y


# ------------------------------------------------------------------ gr-analog pwr_squelch_cc / squelch_base_cc
def pwr_squelch_cc(x: np.ndarray, db=-100.0, alpha=0.01, gate=True) -> np.ndarray:
    """ramp = 0: state flips MUTED <-> UNMUTED immediately; gate=True drops muted samples."""
    thr = 10.0 ** (db / 10.0)
    x = np.asarray(x, dtype=np.complex64)
    pwr = 0.0
    muted = True
    keep = np.zeros(len(x), dtype=bool)
    re, im = x.real.astype(f32), x.imag.astype(f32)
    p = (re * re + im * im).astype(f32)                 # float expression, then widened (single_pole_iir<double,...>)
    for i in range(len(x)):
        pwr = alpha * float(p[i]) + (1.0 - alpha) * pwr
        mute = pwr < thr
        if muted:
            if not mute:
                muted = False
        else:
            if mute:
                muted = True
        keep[i] = not muted
    if gate:
        return x[keep]
    out = x.copy()
    out[~keep] = 0
    return out


def pwr_squelch_margin(x: np.ndarray, db=-100.0, alpha=0.01) -> float:
    """min over the stream of |pwr / threshold - 1| for pwr_squelch_cc's power estimate: how well the gate's decisions are
    conditioned.  Consecutive estimates differ by ~alpha while a carrier fades in or out, so a crossing lands within 1e-5 of
    the threshold about once in a thousand crossings -- and then two channel streams that agree to 1e-7 (the device's and the
    oracle's) may gate one sample apart (tests/test_gpu_fuzz.py, seed 78068)."""
    thr = 10.0 ** (db / 10.0)
    x = np.asarray(x, dtype=np.complex64)
    re, im = x.real.astype(f32), x.imag.astype(f32)
    p = (re * re + im * im).astype(f32)
    pwr, best = 0.0, float("inf")
    for i in range(len(x)):
        pwr = alpha * float(p[i]) + (1.0 - alpha) * pwr
        best = min(best, abs(pwr / thr - 1.0))
    return best


# ------------------------------------------------------------------ gr-filter fir_filter_fff / rational_resampler_base_fff
def fir_filter_fff(x: np.ndarray, taps: np.ndarray) -> np.ndarray:
    """decimation 1, zero history; float64 accumulation rounded once (VOLK's summation order is unspecified)."""
    x64 = np.asarray(x, dtype=np.float64)
    t64 = np.asarray(taps, dtype=np.float64)
    return np.convolve(x64, t64)[: len(x64)].astype(f32)


def design_resampler_taps(interpolation: int, decimation: int, fractional_bw=0.4) -> np.ndarray:
    """rational_resampler.py design_filter()"""
    beta = 7.0
    halfband = 0.5
    rate = float(interpolation) / float(decimation)
    if rate >= 1.0:
        trans_width = halfband - fractional_bw
        mid = halfband - trans_width / 2.0
    else:
        trans_width = rate * (halfband - fractional_bw)
        mid = rate * halfband - trans_width / 2.0
    return low_pass(interpolation, interpolation, mid, trans_width, WIN_KAISER, beta)


def rational_resampler_fff(x: np.ndarray, interpolation: int, decimation: int, taps=None) -> np.ndarray:
    d = math.gcd(int(interpolation), int(decimation))
    if taps is None:
        interpolation, decimation = int(interpolation) // d, int(decimation) // d
        taps = design_resampler_taps(interpolation, decimation)
    taps = np.asarray(taps, dtype=f32)
    pad = (-len(taps)) % interpolation
    tp = np.concatenate([taps, np.zeros(pad, dtype=f32)]).astype(np.float64)
    nt = len(tp) // interpolation
    x64 = np.concatenate([np.zeros(nt - 1), np.asarray(x, dtype=np.float64)])
    n_out = (len(x) * interpolation + decimation - 1) // decimation
    out = np.empty(n_out, dtype=f32)
    for m in range(n_out):
        p = (m * decimation) // interpolation
        ctr = (m * decimation) % interpolation
        sub = tp[ctr::interpolation]                          # xtaps[ctr][k] = taps[ctr + I k]
        seg = x64[p: p + nt][::-1]                            # x[p - k]
        out[m] = f32(np.dot(sub, seg))
    return out


# ------------------------------------------------------------------ the whole chain
def analog_chain(iq: np.ndarray, rate: float, stages=False):
    """logging_receiver.py:211-222 on one channel's complex stream at `rate` samples/s -> 8 kHz float audio"""
    g = pwr_squelch_cc(iq, -100.0, 0.01, True)
    k = rate / (2 * math.pi * 15000)
    fm = G.quadrature_demod_cf(g, f32(k))
    b, a = fm_deemph_taps(rate, 75e-6)
    de = iir_filter_ffd(fm, b, a)
    lpf = optfir_low_pass(8, rate, rate * 0.25, rate * 0.25 + 2000, 0.1, 60)
    au = fir_filter_fff(de, lpf)
    hpf = high_pass(1, rate, 300, 30, WIN_HAMMING, 6.76)
    hp = fir_filter_fff(au, hpf)
    out = rational_resampler_fff(hp, 8000, int(rate))
    if stages:
        return dict(gated=g, fm=fm, deemph=de, lpf=au, hpf=hp, audio=out, lpf_taps=lpf, hpf_taps=hpf)
    return out
