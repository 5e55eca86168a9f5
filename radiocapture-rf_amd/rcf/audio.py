"""Analog NBFM voice chain parameters, derived the way /root/reference/logging_receiver.py:211-222 derives them:

    analog.pwr_squelch_cc(-100, 0.01, 0, True)
    analog.fm_demod_cf(channel_rate=rate, audio_decim=1, deviation=15000, audio_pass=0.25 rate,
                       audio_stop=0.25 rate + 2000, gain=8, tau=75e-6)
    filter.fir_filter_fff(1, firdes.high_pass(1, rate, 300, 30, firdes.WIN_HAMMING, 6.76))
    filter.rational_resampler_fff(interpolation=8000, decimation=rate)

`fm_demod_cf` is GNU Radio's Python hier block (gr-analog fm_demod.py): quadrature_demod_cf(rate / (2 pi dev)) ->
fm_deemph(rate, tau) -> fir_filter_fff(audio_decim, optfir.low_pass(gain, rate, audio_pass, audio_stop, 0.1, 60)).
The window-method designs and the de-emphasis section come from librcf (rcf_design_firdes / _fm_deemph /
_resampler).  The equiripple audio low-pass is Parks-McClellan: GNU Radio's optfir estimates the order
(`remezord`, restated below) and calls its C++ `pm_remez`; scipy.signal.remez is the same exchange algorithm
and is used here -- scipy is already a dependency of the reference (fft_peak_detection.py:7).
"""
from __future__ import annotations

import math

import numpy as np

from . import native


def _lporder(freq1, freq2, delta_p, delta_s):
    """gr-filter optfir.py lporder(): Herrmann/Rabiner/Chan length estimate for a low-pass"""
    df = abs(freq2 - freq1)
    ddp, dds = math.log10(delta_p), math.log10(delta_s)
    a1, a2, a3, a4, a5, a6 = 5.309e-3, 7.114e-2, -4.761e-1, -2.66e-3, -5.941e-1, -4.278e-1
    b1, b2 = 11.01217, 0.5124401
    dinf = ((a1 * ddp * ddp + a2 * ddp + a3) * dds) + (a4 * ddp * ddp + a5 * ddp + a6)
    ff = b1 + b2 * (ddp - dds)
    return dinf / df - ff * df + 1


def optfir_low_pass(gain, fs, freq1, freq2, passband_ripple_db, stopband_atten_db, nextra_taps=2):
    """gr-filter optfir.low_pass()"""
    from scipy.signal import remez
    r = 10.0 ** (passband_ripple_db / 20.0)
    dev_p = ((r - 1.0) / (r + 1.0)) / gain               # remezord: relative deviation for the passband
    dev_s = 10.0 ** (-stopband_atten_db / 20.0)
    f1, f2 = float(freq1) / fs, float(freq2) / fs
    order = int(math.ceil(_lporder(f1, f2, dev_p, dev_s))) - 1
    mx = max(dev_p, dev_s)
    taps = remez(order + nextra_taps + 1, [0.0, f1, f2, 0.5], [gain, 0.0], weight=[mx / dev_p, mx / dev_s],
                 type="bandpass", grid_density=16, fs=1.0)
    return np.asarray(taps, dtype=np.float64).astype(np.float32)


def analog_chain_params(rate, deviation=15000.0, gain=8.0, tau=75e-6, squelch_db=-100.0, squelch_alpha=0.01,
                        audio_rate=8000):
    """keyword arguments of native.Frontend.chan_audio_open for a channel stream at `rate` samples/s"""
    rate = float(rate)
    b, a = native.design_fm_deemph(rate, tau) if tau else ([1.0, 0.0], [1.0, 0.0])
    interp, decim, rs = native.design_resampler(int(audio_rate), int(rate))
    return dict(
        squelch_db=squelch_db, squelch_alpha=squelch_alpha,
        quad_gain=rate / (2 * math.pi * deviation),
        deemph_b=b, deemph_a=a,
        lpf_taps=optfir_low_pass(gain, rate, rate * 0.25, rate * 0.25 + 2000, 0.1, 60),
        hpf_taps=native.design_firdes(native.FIR_HIGH_PASS, 1.0, rate, 300.0, 30.0, native.WIN_HAMMING, 6.76),
        interpolation=interp, decimation=decim, rs_taps=rs)


def open_analog_voice(frontend, chan_id, rate, **kw):
    """attach the reference's analog chain to a channel; read 8 kHz float audio with frontend.chan_read_audio"""
    frontend.chan_audio_open(chan_id, **analog_chain_params(rate, **kw))
