"""CPU: the oracle's restatements of GNU Radio's design and DSP blocks against INDEPENDENT third-party
implementations of the same mathematics (scipy.signal / numpy).  GNU Radio itself cannot be run here (parity
unpinned); these checks pin every formula that has a public twin: windowed-sinc designs (firwin), windows, the
bilinear de-emphasis section, the recursive filters (lfilter), the discriminator (angle of the lag product), the
scan chain (numpy FFT) -- next to scipy.signal.find_peaks and scipy.signal.remez, which the oracle uses live."""
import math

import numpy as np
import pytest
from scipy import signal

from oracle import audio as A
from oracle import grspec as G


@pytest.mark.parametrize("fs,fc,tw,att,win,name", [
    (2.4e6, 6250.0, 6250.0, 20.0, G.WIN_HAMMING, "hamming"),            # channel.py:33 at 2.4 Msps
    (20e6, 6250.0, 6250.0, 20.0, G.WIN_HAMMING, "hamming"),             # ... at 20 Msps (2909 taps)
    (25000.0, 6250.0, 500.0, 30.0, G.WIN_BLACKMAN, "blackman"),         # p25_control_demod.py:106-108 (69 taps)
    (20e6, 31250.0, 15625.0, 60.0, G.WIN_BLACKMAN_HARRIS, "blackmanharris"),   # the bench's PFB prototype
])
def test_low_pass_2_is_scipy_firwin(fs, fc, tw, att, win, name):
    """firdes.low_pass_2 = truncated ideal low-pass x window, scaled to unit DC gain == scipy.signal.firwin"""
    mine = G.low_pass_2(1.0, fs, fc, tw, att, win)
    ref = signal.firwin(len(mine), fc, window=name, fs=fs, scale=True)
    assert np.abs(mine - ref).max() < 2e-7 * np.abs(ref).max() + 1e-9


def test_high_pass_is_scipy_firwin_pass_zero_false():
    mine = A.high_pass(1.0, 25000.0, 300.0, 30.0)                       # logging_receiver.py:215
    ref = signal.firwin(len(mine), 300.0, window="hamming", fs=25000.0, pass_zero=False, scale=True)
    assert len(mine) == 2007 and np.abs(mine - ref).max() < 2e-7


@pytest.mark.parametrize("n", [16, 349, 16384])
def test_windows_are_scipy_windows(n):
    assert np.abs(G.window(G.WIN_HAMMING, n) - signal.windows.hamming(n, sym=True)).max() < 1e-7
    assert np.abs(G.window(G.WIN_BLACKMAN, n) - signal.windows.blackman(n, sym=True)).max() < 1e-7
    assert np.abs(G.window(G.WIN_BLACKMAN_HARRIS, n) - signal.windows.blackmanharris(n, sym=True)).max() < 1e-7
    assert np.abs(A.kaiser(n, 7.0) - signal.windows.kaiser(n, 7.0, sym=True)).max() < 2e-7


def test_deemphasis_is_the_bilinear_transform_of_a_one_pole_lowpass():
    """fm_emph.py: H(s) = w_ca / (s + w_ca) with the prewarped corner, through the bilinear transform"""
    fs, tau = 25000.0, 75e-6
    b, a = A.fm_deemph_taps(fs, tau)
    w_ca = 2.0 * fs * math.tan(1.0 / (2.0 * fs * tau))
    bz, az = signal.bilinear([w_ca], [1.0, w_ca], fs)
    assert np.allclose(b, bz, rtol=1e-12, atol=0) and np.allclose(a, az, rtol=1e-12, atol=0)


def test_recursive_blocks_are_lfilter():
    rng = np.random.default_rng(3)
    x = rng.standard_normal(4000).astype(np.float32)
    b, a = A.fm_deemph_taps(25000.0)
    mine = A.iir_filter_ffd(x, b, a)
    ref = signal.lfilter(b, a, x.astype(np.float64)).astype(np.float32)
    assert np.abs(mine - ref).max() < 1e-6
    # pwr_squelch_cc's power estimate: single-pole IIR of |x|^2; the gate opens where it reaches the threshold
    z = (0.01 * (rng.standard_normal(3000) + 1j * rng.standard_normal(3000))).astype(np.complex64)
    z[:500] = 0
    p = signal.lfilter([0.01], [1.0, -0.99], (np.abs(z.astype(np.complex128)) ** 2))
    want = int(np.count_nonzero(p >= 1e-10))
    assert len(A.pwr_squelch_cc(z, -100.0, 0.01, True)) == want


def test_discriminator_is_the_angle_of_the_lag_product():
    """quadrature_demod_cf: gain * atan2 of x[n] conj(x[n-1]); gr::fast_atan2f is good to ~1e-5 rad"""
    rng = np.random.default_rng(4)
    x = (rng.standard_normal(5000) + 1j * rng.standard_normal(5000)).astype(np.complex64)
    mine = G.quadrature_demod_cf(x, np.float32(6.6315))
    prod = x.astype(np.complex128) * np.conj(np.concatenate([[0], x[:-1]]).astype(np.complex128))
    ref = 6.6315 * np.angle(prod)
    ref[0] = 0.0
    assert np.abs(mine - ref).max() < 6.6315 * 2e-5


def test_scan_chain_is_numpy_fft_log_power_running_sum():
    """fft_vector.py:37-60 against numpy: window, FFT, shift, |X|^2, 10 log10 scaled as nlog10_ff(1, N, 1), 100-frame sum"""
    rng = np.random.default_rng(5)
    N, F, L = 1024, 30, 10
    x = (rng.standard_normal(N * F) + 1j * rng.standard_normal(N * F)).astype(np.complex64)
    mine = G.scan_chain(x, N, F, L)
    w = signal.windows.blackmanharris(N, sym=True)
    X = np.fft.fftshift(np.fft.fft(x.reshape(F, N).astype(np.complex128) * w, axis=1), axes=1)
    v = np.log10(np.abs(X) ** 2) + 1.0
    ref = v[F - L:F].sum(axis=0)
    assert np.abs(mine - ref).max() < 2e-3 and np.abs(mine - ref).mean() < 2e-4


def test_rational_resampler_matches_scipy_upfirdn():
    """rational_resampler_base_fff == zero-stuff by I, FIR, keep every D-th (scipy.signal.upfirdn) with the same taps"""
    rng = np.random.default_rng(6)
    x = rng.standard_normal(3000).astype(np.float32)
    taps = A.design_resampler_taps(8, 25)
    mine = A.rational_resampler_fff(x, 8000, 25000)
    ref = signal.upfirdn(taps.astype(np.float64), x.astype(np.float64), up=8, down=25)[: len(mine)]
    assert np.abs(mine - ref).max() < 1e-5
