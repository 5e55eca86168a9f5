"""Analog NBFM voice chain parameters, derived the way /root/reference/logging_receiver.py:211-222 derives them:

    analog.pwr_squelch_cc(-100, 0.01, 0, True)
    analog.fm_demod_cf(channel_rate=rate, audio_decim=1, deviation=15000, audio_pass=0.25 rate,
                       audio_stop=0.25 rate + 2000, gain=8, tau=75e-6)
    filter.fir_filter_fff(1, firdes.high_pass(1, rate, 300, 30, firdes.WIN_HAMMING, 6.76))
    filter.rational_resampler_fff(interpolation=8000, decimation=rate)

`fm_demod_cf` is GNU Radio's Python hier block (gr-analog fm_demod.py): quadrature_demod_cf(rate / (2 pi dev)) ->
fm_deemph(rate, tau) -> fir_filter_fff(audio_decim, optfir.low_pass(gain, rate, audio_pass, audio_stop, 0.1, 60)).
Every design comes from librcf: the window-method filters, the de-emphasis section and the resampler taps
(rcf_design_firdes / _fm_deemph / _resampler) and the equiripple audio low-pass (rcf_design_optfir_low_pass:
GNU Radio's `remezord` order estimate + the Parks-McClellan exchange of its `pm_remez`; checked against
scipy.signal.remez in tests/test_oracle_audio.py).
"""
from __future__ import annotations

import math

from . import native


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
        lpf_taps=native.design_optfir_low_pass(gain, rate, rate * 0.25, rate * 0.25 + 2000, 0.1, 60),
        hpf_taps=native.design_firdes(native.FIR_HIGH_PASS, 1.0, rate, 300.0, 30.0, native.WIN_HAMMING, 6.76),
        interpolation=interp, decimation=decim, rs_taps=rs)


def dsd_feed_params(rate, fm_demod_gain, out_rate=48000):
    """logging_receiver.py:333-349 ('provoice' gain 0.6, 'dsd_p25' gain 0.4): quadrature_demod_cf(gain) ->
    rational_resampler_fff(48000, rate) -- the float stream the dsd vocoder block consumes.  Expressed with the same
    chain: squelch threshold -inf dB (power is never below 0: nothing is gated), identity de-emphasis and
    one-tap unit filters, the reference's default resampler taps."""
    interp, decim, rs = native.design_resampler(int(out_rate), int(rate))
    return dict(squelch_db=float("-inf"), squelch_alpha=0.0, quad_gain=float(fm_demod_gain),
                deemph_b=[1.0, 0.0], deemph_a=[1.0, 0.0], lpf_taps=[1.0], hpf_taps=[1.0],
                interpolation=interp, decimation=decim, rs_taps=rs)


def open_analog_voice(frontend, chan_id, rate, **kw):
    """attach the reference's analog chain to a channel; read 8 kHz float audio with frontend.chan_read_audio"""
    frontend.chan_audio_open(chan_id, **analog_chain_params(rate, **kw))
