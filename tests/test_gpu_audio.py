"""-m gpu: the analog NBFM voice chain (logging_receiver.py:211-222) through the C ABI against
oracle/audio.py.  Bar: BASELINE.json north_star "demodulated audio within 1e-4 RMS"."""
import math

import numpy as np
import pytest

from oracle import audio as A
from oracle import cbind as OC
from oracle import grspec as G
from rcf import audio as host_audio
from rcf import synth

pytestmark = pytest.mark.gpu


def oracle_channel(x, fs, cr, f0):
    D, taps = G.channel_params(fs, cr)
    ct, incr = OC.xlating_composite(taps, D, f0, fs)
    y, _ = OC.channel_bank(x, D, ct[None, :], np.array([incr]), acc_double=True)
    return y[0]


def rms(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))


def test_analog_voice_chain_equals_oracle(gpu_required):
    nat = gpu_required
    x, meta = synth.cfg1(seconds=0.6)
    fs = meta["fs"]
    cuts = [0, 100000, 100000 + 96 * 2000 + 17, 900001, len(x)]            # ragged pushes
    with nat.Frontend(fs, meta["center_freq"]) as fe:
        cid = fe.chan_open(12500, meta["offset"])
        host_audio.open_analog_voice(fe, cid, 25000)
        fe.timing_enable(True)
        for a, b in zip(cuts[:-1], cuts[1:]):
            fe.push(x[a:b])
        assert fe.timing_read(nat.T_AUDIO)[1] == 4
        n_audio, n_ungated = fe.chan_audio_produced(cid)
        audio = fe.chan_read_audio(cid)
    y = oracle_channel(x, fs, 12500, meta["offset"])
    st = A.analog_chain(y, 25000.0, stages=True)
    assert n_ungated == len(st["gated"]) == len(y)                         # noise keeps the squelch open
    assert len(audio) == n_audio == len(st["audio"]) == (len(y) * 8 + 24) // 25
    print("audio rms error vs oracle: %.3e (signal rms %.3f)" % (rms(audio, st["audio"]), rms(audio, 0 * audio)))
    assert rms(audio, st["audio"]) < 1e-4
    # and it is the 1 kHz tone (deviation 2.5 kHz -> 8 * 2500 / 15000 = 1.33 before de-emphasis)
    seg = audio[1500:].astype(np.float64)
    m = np.arange(len(seg))
    c = 2 * np.mean(seg * np.exp(-2j * math.pi * 1000.0 * m / 8000.0))
    assert 1.0 < abs(c) < 1.35


def test_squelch_gate_drops_silence_like_the_oracle(gpu_required):
    """pwr_squelch_cc(..., gate=True) removes samples: the stages behind it run on a shorter stream"""
    nat = gpu_required
    fs, cr, f0 = 2.4e6, 12500, 150000.0
    n = 96 * 9000
    burst = synth.nbfm_carrier(n, fs, f0, 700.0, 2500.0, 0.4).astype(np.complex64)
    x = burst.copy()
    x[96 * 2500: 96 * 6500] = 0                                            # 4000 channel samples of exact silence
    with nat.Frontend(fs) as fe:
        cid = fe.chan_open(cr, f0)
        host_audio.open_analog_voice(fe, cid, 25000)
        fe.push(x[: 96 * 3000 + 5])
        fe.push(x[96 * 3000 + 5:])
        n_audio, n_ungated = fe.chan_audio_produced(cid)
        audio = fe.chan_read_audio(cid)
    y = oracle_channel(x, fs, cr, f0)
    st = A.analog_chain(y, 25000.0, stages=True)
    assert len(st["gated"]) < len(y) - 1500                                # the oracle did gate
    assert n_ungated == len(st["gated"])
    assert len(audio) == n_audio == len(st["audio"])
    assert rms(audio, st["audio"]) < 1e-4


def test_audio_chain_opened_mid_stream_starts_from_zero_state(gpu_required):
    nat = gpu_required
    x, meta = synth.cfg1(seconds=0.4)
    fs = meta["fs"]
    cut = 96 * 3000 + 40
    with nat.Frontend(fs, meta["center_freq"]) as fe:
        cid = fe.chan_open(12500, meta["offset"])
        fe.push(x[:cut])
        k0 = fe.chan_produced(cid)
        host_audio.open_analog_voice(fe, cid, 25000)
        fe.push(x[cut:])
        audio = fe.chan_read_audio(cid)
        fe.chan_audio_close(cid)
        with pytest.raises(nat.RcfError):
            fe.chan_read_audio(cid)
    y = oracle_channel(x, fs, 12500, meta["offset"])
    ref = A.analog_chain(y[k0:], 25000.0)                                  # a flowgraph started at that sample
    assert len(audio) == len(ref) and rms(audio, ref) < 1e-4


def test_audio_rings_wrap_and_incremental_reads(gpu_required):
    """small rings (out_capacity 4096 < stream length): every stage's ring wraps several times, audio is read
    after each push -- the concatenation equals the one-shot oracle"""
    nat = gpu_required
    x, meta = synth.cfg1(seconds=0.8)
    fs = meta["fs"]
    got = []
    with nat.Frontend(fs, meta["center_freq"], out_capacity=4096) as fe:
        cid = fe.chan_open(12500, meta["offset"])
        host_audio.open_analog_voice(fe, cid, 25000)
        step = 96 * 1500 + 31                                              # 1500 channel samples per push
        for at in range(0, len(x), step):
            fe.push(x[at:at + step])
            got.append(fe.chan_read_audio(cid))
    audio = np.concatenate(got)
    y = oracle_channel(x, fs, 12500, meta["offset"])
    assert len(y) > 4 * 4096
    ref = A.analog_chain(y, 25000.0)
    assert len(audio) == len(ref) and rms(audio, ref) < 1e-4


def test_dsd_feed_chain_demod_plus_48k_resampler(gpu_required):
    """logging_receiver.py 'provoice' / 'dsd_p25' front half: quadrature_demod_cf(0.4) -> rational_resampler_fff(48000,
    rate), the stream dsd.block_ff consumes; same device chain with the squelch, de-emphasis and filters neutral"""
    nat = gpu_required
    x, meta = synth.cfg1(seconds=0.3)
    fs = meta["fs"]
    with nat.Frontend(fs, meta["center_freq"]) as fe:
        cid = fe.chan_open(12500, meta["offset"])
        fe.chan_audio_open(cid, **host_audio.dsd_feed_params(25000, 0.4))
        fe.push(x[:300001])
        fe.push(x[300001:])
        n_out, n_ungated = fe.chan_audio_produced(cid)
        out = fe.chan_read_audio(cid)
    y = oracle_channel(x, fs, 12500, meta["offset"])
    fm = G.quadrature_demod_cf(y, np.float32(0.4))
    ref = A.rational_resampler_fff(fm, 48000, 25000)
    assert n_ungated == len(y) and len(out) == n_out == len(ref) == (len(y) * 48 + 24) // 25
    assert rms(out, ref) < 1e-4
