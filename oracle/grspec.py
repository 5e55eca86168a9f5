"""CPU restatement (numpy) of the GNU Radio 3.8 blocks on the radiocapture-rf hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it, and only
as the checker.  The product path (``radiocapture-rf_amd/``) never imports this module.

PARITY UNPINNED.  GNU Radio 3.8 / VOLK / FFTW are third-party dependencies of the reference that
are neither vendored under /root/reference nor installable here, and the reference ships no tests,
golden vectors or fixtures for this path.  Every function below is a
restatement of the published GR 3.8 block semantics ("[GR-spec]" in SURVEY.md section 8(c)) and is
validated by analytic known-answer tests (tests/test_oracle_kat.py), by agreement with the
independent C restatement in oracle/rcf_oracle.c, and -- for every formula that has a public twin --
against third-party implementations of the same mathematics (scipy.signal.firwin / windows / bilinear /
lfilter, numpy FFT: tests/test_oracle_thirdparty.py).  GNU Radio's float32 operation ORDER (tap-phase
rounding, rotator iteration, VOLK summation) has no such twin and stays unpinned.  The live third-party
oracles are ``scipy.signal.find_peaks`` (oracle/peaks.py) and ``scipy.signal.remez`` (oracle/audio.py).
What the reference does in its own Python IS pinned, by running it in the build container (tests/golden/*.json|npz and
the make_*_goldens.py beside them): the channel parameter rule and the arguments handed to GNU Radio's blocks
(channel.py, p25_control_demod.py, logging_receiver.py, fft_vector.py), the peak detection, the control plane.

Reference call sites each function follows (paths relative to /root/reference):
  * low_pass_2 / windows      rc_frontend/channel.py:33, p25_control_demod.py:106-108
  * channel_params            rc_frontend/channel.py:31-35
  * xlating_fir_ccc           rc_frontend/channel.py:35,61-63
  * quadrature_demod_cf       p25_control_demod.py:120-121, moto_control_demod.py:105,
                              edacs_control_demod.py:84, logging_receiver.py:233-234,335-336,345-346
  * scan_chain                fft_vector.py:37-60
"""
from __future__ import annotations

import math
import numpy as np

WIN_HAMMING = 0
WIN_BLACKMAN = 2
WIN_BLACKMAN_HARRIS = 5

f32 = np.float32


# --------------------------------------------------------------------------------------------
# windows (gr-fft window.cc) -- float32 results, M = ntaps - 1
# --------------------------------------------------------------------------------------------
def _coswindow(ntaps: int, coeffs) -> np.ndarray:
    """GR `window::coswindow`: coefficients are C floats, the cosines are evaluated in double."""
    M = float(ntaps - 1)
    n = np.arange(ntaps, dtype=np.float64)
    c = [float(f32(v)) for v in coeffs]
    w = np.full(ntaps, c[0], dtype=np.float64)
    sign = -1.0
    for k in range(1, len(c)):
        w += sign * c[k] * np.cos((2.0 * k * math.pi * n) / M)
        sign = -sign
    return w.astype(f32)


def hamming(ntaps: int) -> np.ndarray:
    M = float(ntaps - 1)
    n = np.arange(ntaps, dtype=np.float64)
    return (0.54 - 0.46 * np.cos((2.0 * math.pi * n) / M)).astype(f32)


def blackman(ntaps: int) -> np.ndarray:
    return _coswindow(ntaps, (0.42, 0.5, 0.08))


def blackman_harris(ntaps: int) -> np.ndarray:
    """4-term, 92 dB Blackman-Harris (GR default for `window.blackmanharris(n)`)."""
    return _coswindow(ntaps, (0.35875, 0.48829, 0.14128, 0.01168))


def window(wintype: int, ntaps: int) -> np.ndarray:
    if wintype == WIN_HAMMING:
        return hamming(ntaps)
    if wintype == WIN_BLACKMAN:
        return blackman(ntaps)
    if wintype == WIN_BLACKMAN_HARRIS:
        return blackman_harris(ntaps)
    raise ValueError("unsupported window type %r" % (wintype,))


# --------------------------------------------------------------------------------------------
# firdes.low_pass_2 (gr-filter firdes.cc)
# --------------------------------------------------------------------------------------------
def ntaps_windes(fs: float, tw: float, att_db: float) -> int:
    n = int(att_db * fs / (22.0 * tw))
    if (n & 1) == 0:
        n += 1
    return n


def low_pass_2(gain: float, fs: float, fc: float, tw: float, att_db: float,
               wintype: int = WIN_HAMMING) -> np.ndarray:
    ntaps = ntaps_windes(fs, tw, att_db)
    w = window(wintype, ntaps)
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


# --------------------------------------------------------------------------------------------
# rc_frontend/channel.py:31-35 parameter derivation
# --------------------------------------------------------------------------------------------
def channel_params(samp_rate: float, channel_rate: float, py2_floor: bool = False):
    """(D, taps) exactly as rc_frontend/channel.py:31-35 derives them.

    The reference computes ``int(samp_rate/channel_rate)/2`` (a python float).  Only integral
    values are meaningful; non-integral cases are rejected (documented divergence) --
    unless ``py2_floor``: under Python 2, which the line was written for, int / int floors
    (configs/config_denver_massive_p25.py:20: 10 666 666 sps -> 853 / 2 = 426).
    """
    q = int(samp_rate / channel_rate)
    if (q % 2 != 0 and not py2_floor) or q < 2:
        raise ValueError("decimation int(fs/cr)/2 is not a positive integer for fs=%r cr=%r"
                         % (samp_rate, channel_rate))
    D = q // 2
    taps = low_pass_2(1.0, float(samp_rate), channel_rate / 2, channel_rate / 2, 20.0, WIN_HAMMING)
    return D, taps


# --------------------------------------------------------------------------------------------
# filter.freq_xlating_fir_filter_ccc
# --------------------------------------------------------------------------------------------
def _libm_sincosf(theta):
    """(cosf, sinf) of float32 angles by the C library's float routines -- what GNU Radio's exp(gr_complex(0, x)) /
    gr_expj(x) end in (cexpf -> sincosf).  numpy's own float32 cos / sin and a correctly rounded double result both
    differ from libm's in the last bit on a quarter of the arguments: harmless per tap (6e-8), but one ulp in the ROTATOR
    INCREMENT is a phase drift of 6e-8 rad per output -- 1e-5 of IQ error after a few hundred outputs, the size of the
    parity bar.  The C restatement (rcf_oracle.c) and the product (rcf_design.cpp) call the same routines."""
    import ctypes
    import ctypes.util
    th = np.ascontiguousarray(theta, dtype=f32).ravel()
    try:
        libm = _libm_sincosf.lib
    except AttributeError:
        libm = _libm_sincosf.lib = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
        libm.cosf.restype = libm.sinf.restype = ctypes.c_float
        libm.cosf.argtypes = libm.sinf.argtypes = [ctypes.c_float]
    c = np.fromiter((libm.cosf(float(v)) for v in th), dtype=f32, count=len(th))
    s = np.fromiter((libm.sinf(float(v)) for v in th), dtype=f32, count=len(th))
    return c, s


def xlating_composite(taps: np.ndarray, D: int, f0: float, fs: float):
    """GR `build_composite_fir`: float32 phase arithmetic is part of the answer.

    returns (ctaps complex64[T], incr complex64) where
      fwT0 = float32(2 pi f0 / fs); ctaps[i] = taps[i] * exp(j * float32(i * fwT0));
      incr = exp(j * float32(-fwT0 * D)).
    """
    fwT0 = f32(2.0 * math.pi * f0 / fs)
    i = np.arange(len(taps), dtype=np.uint32).astype(f32)
    theta = (i * fwT0).astype(f32)
    c, s = _libm_sincosf(theta)
    ctaps = np.empty(len(taps), dtype=np.complex64)
    ctaps.real = taps.astype(f32) * c
    ctaps.imag = taps.astype(f32) * s
    a = f32(f32(-fwT0) * f32(D))
    ci, si = _libm_sincosf(np.array([a], dtype=f32))
    incr = np.complex64(complex(ci[0], si[0]))
    return ctaps, incr


def rotator_phases(incr: np.complex64, n: int, phase0=np.complex64(1.0), counter0: int = 0):
    """GR `blocks::rotator`: z = in*phase; phase *= incr; every 512th call phase /= |phase|.

    float32 complex arithmetic without FMA contraction (x86-64 baseline build of GR).
    returns (phases complex64[n] applied to outputs 0..n-1, phase_after, counter_after).
    """
    out = np.empty(n, dtype=np.complex64)
    pr, pi = f32(phase0.real), f32(phase0.imag)
    ir, ii = f32(incr.real), f32(incr.imag)
    cnt = counter0
    for k in range(n):
        out[k] = complex(pr, pi)
        cnt += 1
        nr = f32(f32(pr * ir) - f32(pi * ii))
        ni = f32(f32(pr * ii) + f32(pi * ir))
        pr, pi = nr, ni
        if cnt % 512 == 0:
            mag = f32(math.hypot(float(pr), float(pi)))
            pr, pi = f32(pr / mag), f32(pi / mag)
    return out, np.complex64(complex(pr, pi)), cnt


def fir_decim_cc(x: np.ndarray, ctaps: np.ndarray, D: int, n_out: int | None = None,
                 chunk: int = 2048) -> np.ndarray:
    """v[n] = sum_i ctaps[i] * x[n*D - i], x[t<0] = 0 (GR history of T-1 zeros).

    The dot product is accumulated in float64 and rounded once to complex64: VOLK's and the
    GPU's float32 summation orders both approximate this value to ~1e-6 relative.
    Output n exists as soon as sample n*D exists: n_out = floor((len(x)-1)/D) + 1.
    """
    x = np.asarray(x, dtype=np.complex64)
    T = len(ctaps)
    if n_out is None:
        n_out = 0 if len(x) == 0 else (len(x) - 1) // D + 1
    xp = np.concatenate([np.zeros(T - 1, dtype=np.complex128), x.astype(np.complex128)])
    crev = ctaps[::-1].astype(np.complex128)        # crev[j] = ctaps[T-1-j]
    out = np.empty(n_out, dtype=np.complex64)
    for n0 in range(0, n_out, chunk):
        n1 = min(n_out, n0 + chunk)
        idx = (np.arange(n0, n1) * D)[:, None] + np.arange(T)[None, :]   # xp[nD + j], j=0..T-1
        out[n0:n1] = (xp[idx] @ crev).astype(np.complex64)
    return out


def xlating_fir_ccc(x: np.ndarray, D: int, taps: np.ndarray, f0: float, fs: float,
                    n_out: int | None = None) -> np.ndarray:
    """y[n] = rot[n] * sum_i (h[i] e^{j i fwT0}) x[nD - i]   (GR-faithful float32 phases)."""
    ctaps, incr = xlating_composite(taps, D, f0, fs)
    v = fir_decim_cc(x, ctaps, D, n_out)
    ph, _, _ = rotator_phases(incr, len(v))
    # complex64 multiply, unfused
    vr, vi = v.real.astype(f32), v.imag.astype(f32)
    pr, pi = ph.real.astype(f32), ph.imag.astype(f32)
    y = np.empty(len(v), dtype=np.complex64)
    y.real = vr * pr - vi * pi
    y.imag = vr * pi + vi * pr
    return y


def xlating_fir_exact(x: np.ndarray, D: int, taps: np.ndarray, f0: float, fs: float,
                      n_out: int | None = None) -> np.ndarray:
    """Same operator with mathematically exact (float64) phases -- what a polyphase filterbank
    computes for an on-grid bin.  complex128 result."""
    x = np.asarray(x, dtype=np.complex64)
    T = len(taps)
    w0 = 2.0 * math.pi * f0 / fs
    i = np.arange(T, dtype=np.float64)
    ct = taps.astype(np.float64) * np.exp(1j * w0 * i)
    if n_out is None:
        n_out = 0 if len(x) == 0 else (len(x) - 1) // D + 1
    xp = np.concatenate([np.zeros(T - 1, dtype=np.complex128), x.astype(np.complex128)])
    crev = ct[::-1]
    out = np.empty(n_out, dtype=np.complex128)
    for n0 in range(0, n_out, 2048):
        n1 = min(n_out, n0 + 2048)
        idx = (np.arange(n0, n1) * D)[:, None] + np.arange(T)[None, :]
        out[n0:n1] = xp[idx] @ crev
    n = np.arange(n_out, dtype=np.float64)
    return out * np.exp(-1j * w0 * D * n)


# --------------------------------------------------------------------------------------------
# analog.quadrature_demod_cf  (gr::fast_atan2f: 256-interval table + linear interpolation)
# --------------------------------------------------------------------------------------------
TAN_MAP_RES = f32(0.003921569)
TAN_MAP_SIZE = 255
# GNU Radio ships this table as 257 decimal literals of 7 significant digits (0.000000e+00, 3.921549e-03, ... ,
# 7.853982e-01, 7.853982e-01): atan(i / 255) PRINTED with %.6e, then read as float -- which is not always the float
# nearest to atan(i / 255).  Restated the same way (the literals themselves are not available here: parity unpinned).
FAST_ATAN_TABLE = np.array([float("%.6e" % math.atan(i / 255.0)) for i in range(256)] +
                           [float("%.6e" % (math.pi / 4.0))], dtype=np.float64).astype(f32)   # last one duplicated


def fast_atan2f(y: np.ndarray, x: np.ndarray) -> np.ndarray:
    y = np.asarray(y, dtype=f32)
    x = np.asarray(x, dtype=f32)
    ya, xa = np.abs(y), np.abs(x)
    zero = ~((ya > 0) | (xa > 0))
    with np.errstate(divide="ignore", invalid="ignore"):
        z = np.where(ya < xa, ya / xa, xa / ya).astype(f32)
    z = np.where(zero, f32(0), z).astype(f32)
    alpha = (z * f32(TAN_MAP_SIZE)).astype(f32)
    index = alpha.astype(np.int32) & 0xFF
    frac = (alpha - index.astype(f32)).astype(f32)
    t0 = FAST_ATAN_TABLE[index]
    t1 = FAST_ATAN_TABLE[index + 1]
    base = (t0 + ((t1 - t0).astype(f32) * frac).astype(f32)).astype(f32)
    base = np.where(z < TAN_MAP_RES, z, base).astype(f32)
    PI, PI2 = f32(math.pi), f32(math.pi / 2)
    # octant fix-up
    ang_x = np.where(x >= 0, np.where(y >= 0, base, -base),
                     np.where(y >= 0, PI - base, base - PI))
    ang_y = np.where(y >= 0, np.where(x >= 0, PI2 - base, PI2 + base),
                     np.where(x >= 0, -PI2 + base, -PI2 - base))
    ang = np.where(xa > ya, ang_x, ang_y).astype(f32)
    return np.where(zero, f32(0), ang).astype(f32)


def quadrature_demod_cf(x: np.ndarray, gain: float, prev=np.complex64(0)) -> np.ndarray:
    """out[n] = gain * fast_atan2f(imag(x[n] conj(x[n-1])), real(.)), history 1 sample (zero)."""
    x = np.asarray(x, dtype=np.complex64)
    xm1 = np.concatenate([[np.complex64(prev)], x[:-1]])
    ar, ai = x.real.astype(f32), x.imag.astype(f32)
    br, bi = xm1.real.astype(f32), xm1.imag.astype(f32)
    tr = (ar * br + ai * bi).astype(f32)          # a * conj(b)
    ti = (ai * br - ar * bi).astype(f32)
    return (f32(gain) * fast_atan2f(ti, tr)).astype(f32)


def p25_fm_gain(rate: float) -> float:
    """p25_control_demod.py:106,120: fm_demod_gain = channel_rate / (2 pi symbol_deviation)."""
    return rate / (2.0 * math.pi * 600.0)


# --------------------------------------------------------------------------------------------
# scan chain: fft_vector.py:37-60
# --------------------------------------------------------------------------------------------
def fft_vcc_shift(frames: np.ndarray, win: np.ndarray) -> np.ndarray:
    """fft_vcc(N, forward, window, shift=True): X = fftshift(FFT(x * w)), complex64."""
    N = frames.shape[-1]
    buf = (frames.astype(np.complex64) * win.astype(f32)[None, :]).astype(np.complex64)
    X = np.fft.fft(buf.astype(np.complex128), axis=-1)
    half = (N + 1) // 2
    X = np.concatenate([X[..., half:], X[..., :half]], axis=-1)
    return X.astype(np.complex64)


def nlog10_ff(p: np.ndarray, n: float = 1.0, k: float = 1.0) -> np.ndarray:
    """nlog10_ff(n, vlen, k): VOLK log2 (generic: log2f, -inf -> -127) * (n / log2(10)) + k."""
    with np.errstate(divide="ignore"):
        l2 = np.log2(p.astype(f32)).astype(f32)
    l2 = np.where(np.isinf(l2), np.copysign(f32(127.0), l2), l2).astype(f32)
    scale = f32(n / math.log2(10.0))
    return ((l2 * scale).astype(f32) + f32(k)).astype(f32)


def moving_sum_ff(v: np.ndarray, length: int = 100, scale: float = 1.0) -> np.ndarray:
    """moving_average_ff(length, scale, max_iter, vlen): float32 running sum, add-then-subtract."""
    F, N = v.shape
    s = np.zeros(N, dtype=f32)
    out = np.empty_like(v, dtype=f32)
    for i in range(F):
        s = (s + v[i]).astype(f32)
        out[i] = (s * f32(scale)).astype(f32)
        if i - (length - 1) >= 0:
            s = (s - v[i - (length - 1)]).astype(f32)
    return out


def scan_chain(x: np.ndarray, N: int = 16384, n_frames: int = 1000, avg_len: int = 100):
    """fft_vector.py flowgraph: the single float32[N] vector written to /tmp/fft_source_<i>.

    stream_to_vector(N) -> fft_vcc(N, True, blackmanharris(N), True) -> complex_to_mag_squared
    -> nlog10_ff(1,N,1) -> moving_average_ff(avg_len,1,..,N) -> head(n_frames)
    -> skiphead(n_frames-1).
    """
    x = np.asarray(x, dtype=np.complex64)
    if len(x) < N * n_frames:
        raise ValueError("need %d samples, have %d" % (N * n_frames, len(x)))
    win = blackman_harris(N)
    s = np.zeros(N, dtype=f32)
    ring = []
    out = None
    for i in range(n_frames):
        X = fft_vcc_shift(x[i * N:(i + 1) * N][None, :], win)[0]
        mag2 = (X.real.astype(f32) * X.real.astype(f32) + X.imag.astype(f32) * X.imag.astype(f32)).astype(f32)
        v = nlog10_ff(mag2, 1.0, 1.0)
        ring.append(v)
        s = (s + v).astype(f32)
        out = s.copy()
        if i - (avg_len - 1) >= 0:
            s = (s - ring[i - (avg_len - 1)]).astype(f32)
            ring[i - (avg_len - 1)] = None
    return out


def scan_chain_periodic(frame_logspec, n_frames: int = 1000, avg_len: int = 100):
    """fft_vector.py's moving_average_ff -> head(n_frames) -> skiphead(n_frames - 1) over a stream whose frames
    repeat with period U = len(frame_logspec): the float32 running sum in GNU Radio's operation order (add the
    newest frame, emit, subtract the frame avg_len - 1 back), frame f being frame_logspec[f % U].  Lets a test run
    the reference's own lengths (1000 frames, 100-frame average) at N = 2^20 without 8 GB of input: the caller
    computes the U distinct log-magnitude frames (cbind.scan_chain(x_u, N, 1, 1)) once."""
    v = [np.asarray(a, dtype=f32) for a in frame_logspec]
    U = len(v)
    s = np.zeros_like(v[0], dtype=f32)
    out = None
    for f in range(n_frames):
        s = (s + v[f % U]).astype(f32)
        if f == n_frames - 1:
            out = s.copy()
        if f - (avg_len - 1) >= 0:
            s = (s - v[(f - (avg_len - 1)) % U]).astype(f32)
    return out
