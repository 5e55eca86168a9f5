"""Committed golden vectors (tests/golden/dsp_vectors.npz, made by tests/golden/make_dsp_goldens.py).

not-gpu: both CPU restatements (numpy and plain C) still reproduce them; gpu: the HIP path through the
C ABI reproduces them.  They pin regressions; the reference itself ships no vectors (parity unpinned)."""
import os

import numpy as np
import pytest

from oracle import cbind as OC
from oracle import grspec as G
from oracle import peaks as P

V = np.load(os.path.join(os.path.dirname(__file__), "golden", "dsp_vectors.npz"))
VA = np.load(os.path.join(os.path.dirname(__file__), "golden", "audio_vectors.npz"))


def rel(a, b):
    return float(np.sqrt(np.mean(np.abs(a - b) ** 2) / np.mean(np.abs(b) ** 2)))


def test_oracles_reproduce_channel_vectors():
    D, taps = G.channel_params(2.4e6, 12500)
    y = G.xlating_fir_ccc(V["cfg1_x"], D, taps, -62500.0, 2.4e6)
    np.testing.assert_array_equal(y, V["cfg1_y"])
    ct, incr = OC.xlating_composite(taps, D, -62500.0, 2.4e6)
    yc, fm = OC.channel_bank(V["cfg1_x"], D, ct[None, :], np.array([incr]), gains=[5.0])
    assert rel(yc[0], V["cfg1_y"]) < 2e-6
    assert np.sqrt(np.mean((fm[0] - V["cfg1_fm5"]) ** 2)) < 2e-5
    np.testing.assert_allclose(G.quadrature_demod_cf(V["cfg1_y"], G.p25_fm_gain(25000.0)), V["cfg1_fm_p25"],
                               rtol=0, atol=1e-6)
    Dw, tw = G.channel_params(20e6, 12500)
    ctw, incw = OC.xlating_composite(tw, Dw, 5.0125e6, 20e6)
    yw, _ = OC.channel_bank(V["wide_x"], Dw, ctw[None, :], np.array([incw]))
    assert rel(yw[0], V["wide_y"]) < 2e-6


def test_oracles_reproduce_scan_vectors():
    spec = OC.scan_chain(V["scan_x"], 512, 24, 8)
    assert np.abs(spec - V["scan_spec"]).max() < 2e-3
    lines, freqs = P.peak_detect_scipy(V["scan_spec"], 2.4e6, 855.05e6)
    np.testing.assert_array_equal(lines, V["scan_lines"])
    np.testing.assert_array_equal(np.array(freqs), V["scan_freqs"])
    l2, _ = P.peak_detect_restated(V["scan_spec"], 2.4e6, 855.05e6)
    np.testing.assert_array_equal(l2, V["scan_lines"])
    from rcf import scan
    l3, f3 = scan.peak_detect(V["scan_spec"], 2.4e6, 855.05e6)          # product host picker (librcf)
    np.testing.assert_array_equal(l3, V["scan_lines"])
    assert f3 == list(V["scan_freqs"])


def test_oracle_and_library_reproduce_audio_vectors():
    from oracle import audio as A
    from rcf import audio as host_audio
    st = A.analog_chain(VA["iq"], 25000.0, stages=True)
    assert len(st["gated"]) == int(VA["gated_len"]) < len(VA["iq"])          # the silence was gated
    np.testing.assert_array_equal(st["audio"], VA["audio"])
    p = host_audio.analog_chain_params(25000)                                # librcf's designs (no oracle, no scipy)
    np.testing.assert_array_equal(p["lpf_taps"], VA["lpf_taps"])
    np.testing.assert_array_equal(p["hpf_taps"], VA["hpf_taps"])
    np.testing.assert_array_equal(p["rs_taps"], VA["rs_taps"])
    assert list(VA["deemph_b"]) == p["deemph_b"] and list(VA["deemph_a"]) == p["deemph_a"]


@pytest.mark.gpu
def test_hip_audio_chain_reproduces_golden_vectors(gpu_required):
    """the committed channel-rate stream goes in through a pass-through channel (decimation 1, one unit tap), the
    analog voice chain behind it must give the committed 8 kHz audio, gated silence included"""
    nat = gpu_required
    from rcf import audio as host_audio
    with nat.Frontend(25000.0) as fe:
        cid = fe.chan_open_taps(-1, 1, np.ones(1, dtype=np.float32), 0.0)
        host_audio.open_analog_voice(fe, cid, 25000)
        fe.push(VA["iq"][:1700])
        fe.push(VA["iq"][1700:])
        n_audio, n_ungated = fe.chan_audio_produced(cid)
        audio = fe.chan_read_audio(cid)
    assert n_ungated == int(VA["gated_len"]) and len(audio) == n_audio == len(VA["audio"])
    assert np.sqrt(np.mean((audio - VA["audio"]) ** 2)) < 1e-4


@pytest.mark.gpu
def test_hip_path_reproduces_golden_vectors(gpu_required):
    nat = gpu_required
    with nat.Frontend(2.4e6, 855.05e6) as fe:
        cid = fe.chan_open(12500, -62500.0)
        fe.pfb_open(64, 64, V["pfb_taps"])
        fe.push(V["cfg1_x"])
        y = fe.chan_read_iq(cid)
        fm5 = fe.chan_read_fm(cid, 5.0)
    assert rel(y, V["cfg1_y"]) < 1e-5
    assert np.sqrt(np.mean((fm5 - V["cfg1_fm5"]) ** 2)) < 1e-4
    with nat.Frontend(20e6) as fe:
        cid = fe.chan_open(12500, 5.0125e6)
        fe.push(V["wide_x"])
        assert rel(fe.chan_read_iq(cid), V["wide_y"]) < 1e-5
    with nat.Frontend(2.4e6) as fe:
        fe.pfb_open(64, 64, V["pfb_taps"])
        fe.push(V["pfb_x"])
        for k in (0, 3, 40, 63):
            assert rel(fe.pfb_read_bin(k), V["pfb_bin%d" % k]) < 2e-5
    with nat.Frontend(2.4e6, 855.05e6) as fe:
        fe.scan_start(512, 24, 8)
        fe.push(V["scan_x"])
        spec = fe.scan_result()
        lines, mean, _ = fe.scan_find_peaks()
    assert np.abs(spec - V["scan_spec"]).max() < 2e-3
    np.testing.assert_array_equal(lines, V["scan_lines"])
