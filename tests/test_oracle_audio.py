"""CPU: known-answer tests that pin oracle/audio.py (the restatement of the reference's analog voice chain,
logging_receiver.py:211-222) and check librcf's host-side designs against it bit for bit."""
import math

import numpy as np
import pytest

from oracle import audio as A
from rcf import audio as host_audio
from rcf import native


def gain_at(taps, f, fs):
    n = np.arange(len(taps))
    return abs(np.sum(np.asarray(taps, dtype=np.float64) * np.exp(-2j * math.pi * f / fs * n)))


def test_kaiser_window_and_izero():
    assert abs(A.izero(0.0) - 1.0) < 1e-15
    assert abs(A.izero(7.0) - np.i0(7.0)) < 1e-9 * np.i0(7.0)
    w = A.kaiser(101, 7.0)
    assert abs(w[50] - 1.0) < 1e-7 and np.allclose(w, w[::-1], atol=1e-7)
    assert np.allclose(w, np.kaiser(101, 7.0), atol=2e-7)


def test_firdes_tap_counts_and_gains():
    hp = A.high_pass(1, 25000, 300, 30)
    assert len(hp) == 2007                                   # int(53 * 25000 / (22 * 30)) = 2007 (odd)
    assert gain_at(hp, 0, 25000) < 2e-3 and abs(gain_at(hp, 12500, 25000) - 1.0) < 1e-6
    assert abs(gain_at(hp, 1000, 25000) - 1.0) < 1e-2 and gain_at(hp, 100, 25000) < 1e-2
    lp = A.low_pass(1, 2.4e6, 0.6e6, 0.3e6)
    assert len(lp) == 19 and np.array_equal(lp, __import__("oracle.grspec", fromlist=["x"]).low_pass_2(
        1, 2.4e6, 0.6e6, 0.3e6, 53.0))                        # low_pass == low_pass_2 at the window's attenuation
    rs = A.design_resampler_taps(8, 25)
    assert len(rs) == 821 and abs(float(rs.astype(np.float64).sum()) - 8.0) < 1e-5


def test_optfir_low_pass_meets_its_spec():
    t = A.optfir_low_pass(8, 25000, 6250, 8250, 0.1, 60)
    assert len(t) == 45
    for f in (0, 2000, 5000, 6250):
        assert abs(20 * math.log10(gain_at(t, f, 25000) / 8)) < 0.1
    for f in (8250, 10000, 12500):
        assert 20 * math.log10(gain_at(t, f, 25000) / 8) < -58


def test_deemphasis_section():
    b, a = A.fm_deemph_taps(25000, 75e-6)
    assert abs((b[0] + b[1]) / (1 + a[1]) - 1.0) < 1e-12     # unit gain at DC
    x = np.ones(400, dtype=np.float32)
    y = A.iir_filter_ffd(x, b, a)
    assert abs(y[-1] - 1.0) < 1e-6 and y[0] == np.float32(b[0])
    # -3 dB near 1 / (2 pi tau) = 2122 Hz
    w = 2 * math.pi * 2122.0 / 25000
    h = (b[0] + b[1] * np.exp(-1j * w)) / (1 + a[1] * np.exp(-1j * w))
    assert abs(20 * math.log10(abs(h)) + 3.0) < 0.15


def test_squelch_gates_and_passes():
    rng = np.random.default_rng(5)
    sig = (0.5 * np.exp(2j * math.pi * 0.01 * np.arange(3000))).astype(np.complex64)
    x = np.concatenate([np.zeros(100, np.complex64), sig, np.zeros(4000, np.complex64), sig])
    g = A.pwr_squelch_cc(x, -100.0, 0.01, True)
    # opens on the first non-zero sample (0.01 * 0.25 >> 1e-10); closes ln(0.25/1e-10)/0.01005 ~ 2153 samples
    # into the silence; reopens with the second burst
    n_tail = len(g) - 2 * len(sig)
    assert 2100 < n_tail < 2200
    assert np.array_equal(g[:3000], sig)
    ng = A.pwr_squelch_cc(x, -100.0, 0.01, False)
    assert len(ng) == len(x) and np.count_nonzero(ng) == 2 * len(sig)


def test_rational_resampler_rate_and_tone():
    fs = 25000
    x = np.sin(2 * math.pi * 440.0 * np.arange(5000) / fs).astype(np.float32)
    y = A.rational_resampler_fff(x, 8000, fs)
    assert len(y) == (5000 * 8 + 24) // 25
    n = np.arange(400, len(y))
    delay = (821 - 1) / 2 / 8 / fs                            # group delay of the 821-tap filter at 8 x 25 kHz
    ref = np.sin(2 * math.pi * 440.0 * (n / 8000.0 - delay))
    assert np.sqrt(np.mean((y[400:] - ref) ** 2)) < 2e-3


def test_library_designs_equal_the_oracle_bit_for_bit():
    p = host_audio.analog_chain_params(25000)
    assert np.array_equal(p["hpf_taps"], A.high_pass(1, 25000, 300, 30))
    assert np.array_equal(p["rs_taps"], A.design_resampler_taps(8, 25))
    assert np.array_equal(p["lpf_taps"], A.optfir_low_pass(8, 25000, 6250, 8250, 0.1, 60))
    b, a = A.fm_deemph_taps(25000)
    assert p["deemph_b"] == b and p["deemph_a"] == a
    assert (p["interpolation"], p["decimation"]) == (8, 25)
    assert abs(p["quad_gain"] - 25000 / (2 * math.pi * 15000)) < 1e-12
    k = native.design_firdes(native.FIR_LOW_PASS, 8, 8, 0.144, 0.032, native.WIN_KAISER, 7.0)
    assert np.array_equal(k, A.low_pass(8, 8, 0.144, 0.032, A.WIN_KAISER, 7.0))


@pytest.mark.parametrize("args", [(8, 25000, 6250, 8250, 0.1, 60), (1, 48000, 3000, 4000, 0.5, 40),
                                  (8, 12500, 3125, 5125, 0.1, 60), (2, 8000, 1000, 1300, 0.2, 50),
                                  (1, 1.0, 0.1, 0.2, 1.0, 30)])
def test_native_parks_mcclellan_equals_scipy_remez(args):
    """librcf's own exchange (rcf_design_optfir_low_pass) against the live third-party one behind the oracle"""
    mine = native.design_optfir_low_pass(*args)
    ref = A.optfir_low_pass(*args)
    assert len(mine) == len(ref)
    assert np.abs(mine - ref).max() <= 1e-6 * np.abs(ref).max()
    assert np.allclose(mine, mine[::-1], atol=1e-9)


def test_analog_chain_recovers_the_tone():
    """whole oracle chain on a clean FM carrier at the channel rate: 1 kHz tone out, right amplitude"""
    rate, dev, fm = 25000.0, 2500.0, 1000.0
    n = 20000
    t = np.arange(n) / rate
    iq = (0.5 * np.exp(1j * (dev / fm) * np.sin(2 * math.pi * fm * t))).astype(np.complex64)
    y = A.analog_chain(iq, rate)
    assert len(y) == (n * 8 + 24) // 25
    seg = y[2000:].astype(np.float64)
    m = np.arange(len(seg))
    c = 2 * np.mean(seg * np.exp(-2j * math.pi * fm * m / 8000.0))
    # k * 2 pi dev / rate = dev / 15000 per unit gain; x8 audio gain; de-emphasis at 1 kHz: -0.87 dB
    b, a = A.fm_deemph_taps(rate)
    w = 2 * math.pi * fm / rate
    hd = abs((b[0] + b[1] * np.exp(-1j * w)) / (1 + a[1] * np.exp(-1j * w)))
    assert abs(abs(c) - 8 * dev / 15000 * hd) < 0.02 * 8 * dev / 15000
    assert np.sqrt(np.mean((seg - np.real(c * np.exp(2j * math.pi * fm * m / 8000.0))) ** 2)) < 0.02


def test_chain_parameters_are_the_ones_the_reference_hands_to_gnuradio():
    """tests/golden/demod_params.json: the constructor arguments /root/reference/logging_receiver.py ('analog', 'p25') and
    p25_control_demod.py (C4FM) pass to GNU Radio's blocks, read off MagicMock stand-ins while the reference's own
    constructors ran (tests/golden/make_demod_param_goldens.py).  rcf.audio's defaults, the oracle's chain defaults and
    the constants the P25 front-half tests use are those numbers."""
    import inspect
    import json
    import math
    import os
    from oracle import grspec as G
    from rcf import audio as host_audio, native
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "demod_params.json")))
    an = gold["logging_receiver_analog"]
    rate = an["fm_demod_cf"]["kwargs"]["channel_rate"]
    assert rate == 25000 and an["rational_resampler_fff"]["kwargs"] == dict(interpolation=8000, decimation=rate, taps=None, fractional_bw=None)
    d = inspect.signature(host_audio.analog_chain_params).parameters
    sq = an["pwr_squelch_cc"]["args"]
    assert [d["squelch_db"].default, d["squelch_alpha"].default] == sq[:2] and sq[2:] == [0, True]     # ramp 0, gate on
    fm = an["fm_demod_cf"]["kwargs"]
    assert (d["deviation"].default, d["gain"].default, d["tau"].default) == (fm["deviation"], fm["gain"], fm["tau"])
    assert fm["audio_decim"] == 1 and fm["audio_pass"] == rate * 0.25 and fm["audio_stop"] == rate * 0.25 + 2000
    assert d["audio_rate"].default == an["rational_resampler_fff"]["kwargs"]["interpolation"]
    hp = an["high_pass"]["args"]
    assert hp[:4] == [1, rate, 300, 30] and hp[5] == 6.76 and an["high_pass_window_is_hamming"]
    p = host_audio.analog_chain_params(rate)
    assert p["quad_gain"] == rate / (2 * math.pi * fm["deviation"])
    np.testing.assert_array_equal(p["hpf_taps"], native.design_firdes(native.FIR_HIGH_PASS, 1.0, float(rate), 300.0, 30.0,
                                                                        native.WIN_HAMMING, 6.76))
    # the P25 front half (p25_control_demod.py:105-137; logging_receiver's 'p25' branch uses the same gain)
    c4 = gold["p25_control_demod_c4fm"]
    assert c4["low_pass_2"]["args"][:5] == [1.0, 25000, 6250.0, 500.0, 30.0] and c4["low_pass_2_window_is_blackman"]
    assert c4["freq_xlating_fir_filter_ccc"]["args"][0] == 1 and c4["freq_xlating_fir_filter_ccc"]["args"][2:] == [0, 25000]
    assert c4["quadrature_demod_cf"]["args"] == [G.p25_fm_gain(25000.0)] == gold["logging_receiver_p25"]["quadrature_demod_cf"]["args"]
    assert c4["fir_filter_fff"]["args"] == [1, [1.0 / 5] * 5]
    assert c4["moving_average_ff"]["args"][:2] == [10000, 1] and c4["multiply_const_vff"]["args"] == [[0.0001]]
