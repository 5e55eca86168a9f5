"""Synthetic wideband IQ generators: seeded, numpy only.

The reference has no recorded IQ; its `file_to_wav.py` / `logging_receiver.log_dat` replay format is
raw interleaved cf32, which is what these return (complex64).  Used by tests and bench.py as the
'synthetic' source type.
"""
from __future__ import annotations

import math
import numpy as np


def awgn(rng: np.random.Generator, n: int) -> np.ndarray:
    """Unit-variance complex noise: re, im ~ N(0, 1/2)."""
    s = math.sqrt(0.5)
    out = np.empty(n, dtype=np.complex64)
    out.real = rng.standard_normal(n, dtype=np.float32) * np.float32(s)
    out.imag = rng.standard_normal(n, dtype=np.float32) * np.float32(s)
    return out


def nbfm_carrier(n: int, fs: float, f_off: float, f_mod: float, dev: float, amp: float,
                 phase0: float = 0.0, t0: int = 0) -> np.ndarray:
    """amp * exp(j(2 pi f_off t + (dev/f_mod) sin(2 pi f_mod t) + phase0)), complex128."""
    t = (np.arange(n, dtype=np.float64) + t0) / fs
    ph = 2.0 * math.pi * f_off * t + (dev / f_mod) * np.sin(2.0 * math.pi * f_mod * t) + phase0
    return amp * np.exp(1j * ph)


def snr_amp(snr_db: float, bw: float, fs: float) -> float:
    """Carrier amplitude for `snr_db` over unit-variance noise measured in `bw` Hz."""
    return math.sqrt((bw / fs) * 10.0 ** (snr_db / 10.0))


def cfg1(seconds: float = 1.0, seed: int = 1001):
    """Single 12.5 kHz NBFM channel in 2.4 Msps IQ (BASELINE config 1).

    Carrier offset -62 500 Hz = 854 987 500 - 855 050 000
    (/root/reference/configs/config_denver_dev_den817.py:32,127), 1 kHz tone, +-2.5 kHz deviation,
    +30 dB over the noise in 12.5 kHz.
    """
    fs = 2.4e6
    n = int(round(fs * seconds))
    rng = np.random.Generator(np.random.PCG64(seed))
    x = awgn(rng, n).astype(np.complex128)
    x += nbfm_carrier(n, fs, -62500.0, 1000.0, 2500.0, snr_amp(30.0, 12500.0, fs))
    meta = dict(fs=fs, center_freq=855050000, channel_rate=12500, freq=854987500,
                offset=-62500.0, f_mod=1000.0, dev=2500.0)
    return x.astype(np.complex64), meta


def cfg2(n: int = 10_000_000, seed: int = 2002, n_bins: int = 256, n_active: int = 32):
    """32 NBFM carriers in 32 distinct bins of a 256-bin PFB over 20 Msps (BASELINE config 2)."""
    fs = 20e6
    rng = np.random.Generator(np.random.PCG64(seed))
    bin_w = fs / n_bins
    bins = rng.permutation(np.arange(-120, 121))[:n_active]
    deltas = rng.choice(np.array([-25000.0, -12500.0, 0.0, 12500.0, 25000.0]), size=n_active)
    fmods = rng.uniform(300.0, 3000.0, size=n_active)
    snrs = rng.uniform(25.0, 35.0, size=n_active)
    phases = rng.uniform(0, 2 * math.pi, size=n_active)
    x = awgn(rng, n).astype(np.complex128)
    carriers = []
    for k, d, fm, snr, ph in zip(bins, deltas, fmods, snrs, phases):
        f = k * bin_w + d
        x += nbfm_carrier(n, fs, f, fm, 2500.0, snr_amp(snr, 12500.0, fs), ph)
        carriers.append(dict(bin=int(k), delta=float(d), f_off=float(f), f_mod=float(fm),
                             snr_db=float(snr)))
    meta = dict(fs=fs, n_bins=n_bins, carriers=carriers)
    return x.astype(np.complex64), meta


def scan_stream(fs: float, N: int, n_unique_frames: int, carriers, seed: int, scale: float = 0.003):
    """Periodic scan input: `n_unique_frames` frames of N samples, seamless when tiled.

    carriers: list of (bin_centre, fwhm_hz, peak_db) -- bin_centre is the fft-shifted index on the
    N-point grid (0 = -fs/2); each carrier is band-limited Gaussian noise whose PSD is a Gaussian bump
    of full width `fwhm_hz` at half (linear) maximum peaking `peak_db` above the unit noise floor,
    i.e. a smooth single-humped log-spectrum like a busy trunked control channel's.
    The whole tile is synthesised in the frequency domain (one IFFT), hence exactly periodic.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    n = N * n_unique_frames
    fgrid = np.fft.fftfreq(n, d=1.0 / fs)
    psd = np.ones(n, dtype=np.float64)
    for (b, fwhm, peak_db) in carriers:
        f0 = (b - N // 2) * fs / N
        sig = fwhm / 2.3548200450309493
        psd += 10.0 ** (peak_db / 10.0) * np.exp(-0.5 * ((fgrid - f0) / sig) ** 2)
    spec = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * np.sqrt(psd * 0.5)
    x = np.fft.ifft(spec) * math.sqrt(n)
    # RTL-SDR-like level: the noise floor's log10|X|^2 + 1 is negative, so the reference's
    # `data += abs(min(data))` (fft_peak_detection.py:58-59) lifts the floor to ~0 as its author intended
    return (x * scale).astype(np.complex64)
