"""-m gpu parity tests: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Tolerances (BASELINE.json north_star): discriminator output within 1e-4 RMS of the oracle; the
channel IQ is held to 1e-5 relative RMS; peak indices bit-exact.
"""
import math

import numpy as np
import pytest

from oracle import grspec as G
from oracle import cbind as OC
from oracle import peaks as P
from rcf import synth

pytestmark = pytest.mark.gpu

# (fft-shifted bin centre, FWHM Hz, peak dB over the noise PSD): SURVEY 8(d) seed-3004 variant
SCAN_CARRIERS_3004 = [(2000, 9000.0, 25.0), (5200, 12500.0, 30.0), (8192 + 900, 7000.0, 22.0),
                      (11000, 20000.0, 28.0), (15000, 12500.0, 26.0)]


def rel_rms(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    return float(np.sqrt(np.mean(np.abs(a - b) ** 2)) / (np.sqrt(np.mean(np.abs(b) ** 2)) + 1e-30))


def rms(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) ** 2)))


def oracle_channel(x, fs, cr, f0, gains):
    D, taps = G.channel_params(fs, cr)
    ct, incr = OC.xlating_composite(taps, D, f0, fs)
    y, _ = OC.channel_bank(x, D, ct[None, :], np.array([incr]), acc_double=True)
    fms = [OC.quad_demod(y[0], g) for g in gains]
    return y[0], fms


def test_cfg1_single_channel_iq_and_fm(gpu_required):
    nat = gpu_required
    x, meta = synth.cfg1(seconds=0.25)
    with nat.Frontend(meta["fs"], meta["center_freq"]) as fe:
        cid = fe.chan_open(12500, meta["offset"])
        info = fe.chan_info(cid)
        assert info["decim"] == 96 and info["ntaps"] == 349 and info["out_rate"] == 25000.0
        fe.push(x)
        y = fe.chan_read_iq(cid)
        fm5 = fe.chan_read_fm(cid, 5.0)
    g_p25 = G.p25_fm_gain(25000.0)
    yo, (fo5, fop) = oracle_channel(x, meta["fs"], 12500, meta["offset"], [5.0, g_p25])
    assert len(y) == len(yo) == (len(x) - 1) // 96 + 1
    assert rel_rms(y, yo) < 1e-5
    assert rms(fm5, fo5) < 1e-4
    # the demodulated tone: 1 kHz, deviation 2.5 kHz -> peak 5 * 2 pi 2500 / 25000
    n = np.arange(len(fm5) - 50)
    amp = 2 * abs(np.mean(fm5[50:] * np.exp(-2j * math.pi * 1000 * n / 25000)))
    assert abs(amp - 5 * 2 * math.pi * 2500 / 25000) < 0.1


def test_chunked_push_equals_single_push(gpu_required):
    nat = gpu_required
    x, meta = synth.cfg1(seconds=0.1, seed=77)
    rng = np.random.default_rng(5)
    with nat.Frontend(meta["fs"]) as fe:
        cid = fe.chan_open(12500, 311000.0)
        fe.push(x)
        y1 = fe.chan_read_iq(cid)
        f1 = fe.chan_read_fm(cid, 1.0)
    with nat.Frontend(meta["fs"]) as fe:
        cid = fe.chan_open(12500, 311000.0)
        pos, ys, fs_ = 0, [], []
        while pos < len(x):
            n = int(rng.integers(1, 40000))
            fe.push(x[pos:pos + n])
            pos += n
            if rng.random() < 0.5:
                ys.append(fe.chan_read_iq(cid))
                fs_.append(fe.chan_read_fm(cid, 1.0))
        ys.append(fe.chan_read_iq(cid))
        fs_.append(fe.chan_read_fm(cid, 1.0))
    y2, f2 = np.concatenate(ys), np.concatenate(fs_)
    assert len(y1) == len(y2)
    assert rel_rms(y2, y1) < 2e-6
    assert rms(f2, f1) < 2e-5
    yo, _ = oracle_channel(x, meta["fs"], 12500, 311000.0, [])
    assert rel_rms(y2, yo) < 1e-5


@pytest.mark.parametrize("fs,cr,offsets", [
    (20e6, 12500, [1.0e6, 5.0125e6, -9.9875e6, 12500.0, -3.3e6]),
    (2.4e6, 6250, [-62500.0, 1.0e6]),
    (2.4e6, 25000, [400000.0, -1.1e6, 0.0]),
])
def test_multichannel_bank_gr_faithful(gpu_required, fs, cr, offsets):
    """The float32 tap-phase / rotator roundings of GR are reproduced (SURVEY 8(c) (i),(ii))."""
    nat = gpu_required
    rng = np.random.default_rng(int(fs) % 1000 + cr)
    D, taps = G.channel_params(fs, cr)
    n = D * 700
    x = synth.awgn(rng, n)
    for f in offsets:
        x = x + synth.nbfm_carrier(n, fs, f + 300.0, 700.0, 2500.0, 0.5).astype(np.complex64)
    x = x.astype(np.complex64)
    with nat.Frontend(fs) as fe:
        ids = [fe.chan_open(cr, f) for f in offsets]
        fe.push(x)
        ys = [fe.chan_read_iq(c) for c in ids]
        fms = [fe.chan_read_fm(c, G.p25_fm_gain(2 * cr)) for c in ids]
    for f, y, fm in zip(offsets, ys, fms):
        yo, (fo,) = oracle_channel(x, fs, cr, f, [G.p25_fm_gain(2 * cr)])
        assert len(y) == len(yo) == 700
        assert rel_rms(y, yo) < 1e-5, f
        assert rms(fm, fo) < 1e-4, f


@pytest.mark.parametrize("fs,cr,nch,D_override", [(2.4e6, 12500, 37, None), (20e6, 12500, 12, None),
                                                  (2.4e6, 12500, 9, 75), (25e6, 12500, 9, None)])
def test_matrix_core_bank_equals_oracle(gpu_required, fs, cr, nch, D_override):
    """>= 8 channels on one source run on the FP32 matrix cores once their history is real (fir.hip
    fir_mfma_kernel): same GR-faithful result as the oracle, including a channel count that leaves dead MFMA
    rows (37, 12, 9), the reference's 2909-tap / 800 shape, an odd decimation (no LDS skew), and the largest
    shape of the SURVEY 8(a) a3 table that fits the matrix-core tile (25 Msps: D = 1000, T = 3637, 149 KB of LDS)."""
    nat = gpu_required
    rng = np.random.default_rng(nch)
    D, taps = G.channel_params(fs, cr)
    if D_override:
        D = D_override
    T = len(taps)
    n1 = T + 5 * D + 13                      # first block: its zero-history outputs are redone by a vector launch
    n_out = 150 if T > 1000 else 400
    n = n1 + D * n_out + 7
    x = (synth.awgn(rng, n) + synth.nbfm_carrier(n, fs, 31000.0, 700.0, 2500.0, 0.5)).astype(np.complex64)
    offs = [float(np.round(o / 6250) * 6250) for o in np.linspace(-0.4, 0.4, nch) * fs]
    with nat.Frontend(fs) as fe:
        ids = [fe.chan_open_taps(-1, D, taps, f) for f in offs]
        fe.timing_enable(True)
        fe.push(x[:n1])
        assert fe.timing_read(nat.T_FIR_MFMA)[1] == 1 and fe.timing_read(nat.T_FIR)[1] == 1   # + the fix-up launch
        fe.push(x[n1:])
        assert fe.timing_read(nat.T_FIR_MFMA)[1] == 1 and fe.timing_read(nat.T_FIR)[1] == 0   # matrix cores only
        ys = [fe.chan_read_iq(c) for c in ids]
    for f, y in zip(offs, ys):
        ct, incr = OC.xlating_composite(taps, D, f, fs)
        v = G.fir_decim_cc(x, ct, D)
        ph, _, _ = G.rotator_phases(incr, len(v))
        yo = (v * ph).astype(np.complex64)
        assert len(y) == len(yo)
        assert rel_rms(y, yo) < 1e-5, f


def test_largest_channel_shape_20msps_6k25(gpu_required):
    """SURVEY 8(a) a3's biggest filter: 6.25 kHz channels at 20 Msps, D = 1600, T = 5819.  Sixteen outputs' worth of
    samples no longer fit the LDS: the matrix-core kernel runs 8-output tiles (136 KB)."""
    nat = gpu_required
    fs, cr = 20e6, 6250
    rng = np.random.default_rng(3)
    D, taps = G.channel_params(fs, cr)
    assert (D, len(taps)) == (1600, 5819)
    n = D * 260 + 11
    x = synth.awgn(rng, n)
    offs = [-3.7e6 + k * 0.9e6 for k in range(9)]
    with nat.Frontend(fs) as fe:
        ids = [fe.chan_open(cr, f) for f in offs]
        fe.timing_enable(True)
        fe.push(x[: D * 90 + 7])
        fe.push(x[D * 90 + 7:])
        assert fe.timing_read(nat.T_FIR_MFMA)[1] == 2
        ys = [fe.chan_read_iq(c) for c in ids]
    for f, y in zip(offs, ys):
        ct, incr = OC.xlating_composite(taps, D, f, fs)
        v = G.fir_decim_cc(x, ct, D)
        ph, _, _ = G.rotator_phases(incr, len(v))
        yo = (v * ph).astype(np.complex64)
        assert len(y) == len(yo) and rel_rms(y, yo) < 1e-5, f


def test_channel_opened_mid_stream_has_zero_history(gpu_required):
    nat = gpu_required
    fs, cr = 2.4e6, 12500
    rng = np.random.default_rng(9)
    x = synth.awgn(rng, 96 * 400)
    with nat.Frontend(fs) as fe:
        fe.push(x[:96 * 100 + 37])                      # channel starts off the decimation grid
        cid = fe.chan_open(cr, 250000.0)
        fe.push(x[96 * 100 + 37:])
        y = fe.chan_read_iq(cid)
    start = 96 * 100 + 37
    k0 = -(-start // 96)
    xz = x.copy()
    xz[:start] = 0                                      # GR: a new flowgraph starts with zero history
    D, taps = G.channel_params(fs, cr)
    ct, incr = OC.xlating_composite(taps, D, 250000.0, fs)
    v = G.fir_decim_cc(xz, ct, D)[k0:]
    ph, _, _ = G.rotator_phases(incr, len(v))
    yo = (v * ph).astype(np.complex64)
    assert len(y) == len(yo)
    assert rel_rms(y, yo) < 1e-5


def test_retune_keeps_phase_and_history(gpu_required):
    """channel.set_offset (rc_frontend/channel.py:61-63): new taps/increment, same rotator phase."""
    nat = gpu_required
    fs, cr = 2.4e6, 12500
    rng = np.random.default_rng(10)
    D, taps = G.channel_params(fs, cr)
    x = synth.awgn(rng, D * 1300)
    cut = D * 600
    f_a, f_b = 100000.0, -437500.0
    with nat.Frontend(fs) as fe:
        cid = fe.chan_open(cr, f_a)
        fe.push(x[:cut])
        fe.chan_set_offset(cid, f_b)
        fe.push(x[cut:])
        y = fe.chan_read_iq(cid)
    L = OC.lib()
    import ctypes as C
    st = OC.RotState(1.0, 0.0, 0)
    xp = np.concatenate([np.zeros(len(taps) - 1, np.complex64), x])
    base = xp.view(np.float32)[2 * (len(taps) - 1):]
    yo = np.empty(1300, dtype=np.complex64)
    fp = C.POINTER(C.c_float)
    for (f, k0, k1) in ((f_a, 0, 600), (f_b, 600, 1300)):
        ct, incr = OC.xlating_composite(taps, D, f, fs)
        inc = np.array([incr], dtype=np.complex64)
        seg = np.empty(k1 - k0, dtype=np.complex64)
        L.ro_xlating_fir_ccc(base.ctypes.data_as(fp), k0, k1 - k0, D, ct.view(np.float32).ctypes.data_as(fp),
                             len(taps), inc.view(np.float32).ctypes.data_as(fp), C.byref(st),
                             seg.view(np.float32).ctypes.data_as(fp), 1)
        yo[k0:k1] = seg
    assert rel_rms(y, yo) < 1e-5


def test_matrix_core_bank_ring_wrap_and_empty_push(gpu_required):
    """output rings shorter than the stream (wrap inside and between launches), a zero-length push in between"""
    nat = gpu_required
    fs, cr = 2.4e6, 12500
    rng = np.random.default_rng(5)
    D, taps = G.channel_params(fs, cr)
    n = D * 2600 + 5
    x = synth.awgn(rng, n)
    offs = [float(np.round(o / 6250) * 6250) for o in np.linspace(-0.35, 0.35, 9) * fs]
    got = {f: [] for f in offs}
    with nat.Frontend(fs, out_capacity=1024) as fe:
        ids = [fe.chan_open(cr, f) for f in offs]
        step = D * 700 + 13                                  # 700 outputs per push into rings of 1024
        for at in range(0, n, step):
            fe.push(x[at:at + step])
            fe.push(x[:0])                                   # empty block: a no-op
            for f, c in zip(offs, ids):
                got[f].append(fe.chan_read_iq(c))
    for f in offs:
        ct, incr = OC.xlating_composite(taps, D, f, fs)
        v = G.fir_decim_cc(x, ct, D)
        ph, _, _ = G.rotator_phases(incr, len(v))
        yo = (v * ph).astype(np.complex64)
        y = np.concatenate(got[f])
        assert len(y) == len(yo) and rel_rms(y, yo) < 1e-5, f


def test_lagging_reader_gets_the_newest_ring_full(gpu_required):
    """a consumer slower than the ring (ZMQ PUB at its high-water mark drops the same way): reads return the newest
    out_capacity samples, oldest first -- channel IQ, discriminator and a filterbank bin"""
    nat = gpu_required
    fs, cr, nb = 2.4e6, 12500, 64
    rng = np.random.default_rng(17)
    D, taps = G.channel_params(fs, cr)
    x = synth.awgn(rng, D * 3000 + 9)
    proto = G.low_pass_2(1.0, fs, fs / nb * 0.4, fs / nb * 0.2, 60.0, G.WIN_BLACKMAN_HARRIS)
    cap = 1024
    with nat.Frontend(fs, out_capacity=cap) as fe:
        cid = fe.chan_open(cr, 250000.0)
        fe.pfb_open(nb, nb, proto)
        for at in range(0, len(x), D * 500):                 # 500 channel outputs / 750 frames per push, never read
            fe.push(x[at:at + D * 500])
        y = fe.chan_read_iq(cid)
        fm = fe.chan_read_fm(cid, 1.0)
        b = fe.pfb_read_bin(5)
        assert len(fe.chan_read_iq(cid)) == 0               # drained
    yo, (fo,) = oracle_channel(x, fs, cr, 250000.0, [1.0])
    assert len(y) == len(fm) == cap and len(yo) > 2 * cap
    assert rel_rms(y, yo[-cap:]) < 1e-5
    assert rms(fm, fo[-cap:]) < 1e-4
    want = G.xlating_fir_exact(x, nb, proto, 5 * fs / nb, fs).astype(np.complex64)
    assert len(b) == cap and rel_rms(b, want[-cap:]) < 2e-5


def test_long_churn_retunes_opens_closes(gpu_required):
    """600 small pushes with a retune every push and an open / close every few: deferred frees, the bank-matrix cache
    and the launch arenas keep up, and a channel nobody touched still equals the oracle over the whole stream"""
    nat = gpu_required
    fs, cr = 2.4e6, 12500
    rng = np.random.default_rng(99)
    D, taps = G.channel_params(fs, cr)
    step = D * 12 + 5
    n = step * 600
    x = synth.awgn(rng, n)
    offs = [float(np.round(o / 6250) * 6250) for o in np.linspace(-0.42, 0.42, 10) * fs]
    got = []
    with nat.Frontend(fs) as fe:
        ids = [fe.chan_open(cr, f) for f in offs]
        extra = []
        for it in range(600):
            fe.chan_set_offset(ids[1 + it % 8], offs[1 + it % 8] + 6250.0 * ((it // 8) % 5))
            if it % 7 == 0:
                extra.append(fe.chan_open(cr, 100000.0 + 12500.0 * (it % 11)))
            if it % 7 == 3 and extra:
                fe.chan_close(extra.pop(0))
            fe.push(x[it * step:(it + 1) * step])
            if it % 50 == 49:
                got.append(fe.chan_read_iq(ids[0]))
        got.append(fe.chan_read_iq(ids[0]))
    y = np.concatenate(got)
    ct, incr = OC.xlating_composite(taps, D, offs[0], fs)
    v = G.fir_decim_cc(x, ct, D)
    ph, _, _ = G.rotator_phases(incr, len(v))
    yo = (v * ph).astype(np.complex64)
    assert len(y) == len(yo) and rel_rms(y, yo) < 1e-5


def test_matrix_core_bank_survives_retune_close_and_open(gpu_required):
    """The matrix-core path caches per-group tap slabs keyed by (channel ids, tap versions): a retune must repack
    its group, a closed channel must leave, a newly opened channel joins the matrix-core launch at once -- the few
    outputs that still see zero history are redone by a vector-kernel launch queued behind it."""
    nat = gpu_required
    fs, cr = 2.4e6, 12500
    rng = np.random.default_rng(77)
    D, taps = G.channel_params(fs, cr)
    T = len(taps)
    seg = [T + 3 * D + 5, 200 * D, 150 * D + 11, 8 * D + 3, 170 * D]          # warm-up, A, B, C (new channel warms), E
    cuts = np.cumsum([0] + seg)
    x = synth.awgn(rng, int(cuts[-1]))
    offs = [float(np.round(o / 6250) * 6250) for o in np.linspace(-0.4, 0.4, 11) * fs]
    f_new, f_retune = 193750.0, -506250.0
    with nat.Frontend(fs) as fe:
        ids = [fe.chan_open(cr, f) for f in offs]
        fe.timing_enable(True)
        fe.push(x[cuts[0]:cuts[1]])                                  # warm-up: matrix cores + fix-up launch
        fe.push(x[cuts[1]:cuts[2]])                                  # A: matrix-core
        assert fe.timing_read(nat.T_FIR_MFMA, reset=False)[1] == 2
        assert fe.timing_read(nat.T_FIR, reset=False)[1] == 1
        fe.chan_set_offset(ids[3], f_retune)
        fe.chan_close(ids[5])
        fe.push(x[cuts[2]:cuts[3]])                                  # B: matrix-core, repacked (10 channels)
        assert fe.timing_read(nat.T_FIR_MFMA, reset=False)[1] == 3
        late = fe.chan_open(cr, f_new)
        fe.push(x[cuts[3]:cuts[4]])                                  # C: the new channel rides along; its first
        assert fe.timing_read(nat.T_FIR_MFMA, reset=False)[1] == 4   #    outputs are redone by one vector launch
        assert fe.timing_read(nat.T_FIR, reset=False)[1] == 2        #    (block 0 and this one)
        fe.push(x[cuts[4]:cuts[5]])                                  # E: all 11 on the matrix cores
        assert fe.timing_read(nat.T_FIR_MFMA, reset=False)[1] == 5
        assert fe.timing_read(nat.T_FIR, reset=False)[1] == 2
        ys = {c: fe.chan_read_iq(c) for c in ids if c != ids[5]}
        y_late = fe.chan_read_iq(late)

    def bank(xs, f, k0=0):
        ct, incr = OC.xlating_composite(taps, D, f, fs)
        v = G.fir_decim_cc(xs, ct, D)[k0:]
        ph, _, _ = G.rotator_phases(incr, len(v))
        return (v * ph).astype(np.complex64)

    for c, f in zip(ids, offs):
        if c in (ids[3], ids[5]):
            continue
        yo = bank(x, f)
        assert len(ys[c]) == len(yo) and rel_rms(ys[c], yo) < 1e-5, f
    # the late channel: zero history before its start, outputs on the absolute decimation grid
    start = int(cuts[3])
    xz = x.copy()
    xz[:start] = 0
    yo = bank(xz, f_new, -(-start // D))
    assert len(y_late) == len(yo) and rel_rms(y_late, yo) < 1e-5
    # the retuned channel: rotator phase carried across the retune (oracle's stateful C path)
    import ctypes as C
    L = OC.lib()
    st = OC.RotState(1.0, 0.0, 0)
    xp = np.concatenate([np.zeros(T - 1, np.complex64), x])
    base = xp.view(np.float32)[2 * (T - 1):]
    fp = C.POINTER(C.c_float)
    n_tot = (len(x) - 1) // D + 1
    k_cut = -(-int(cuts[2]) // D)
    yo = np.empty(n_tot, dtype=np.complex64)
    for (f, k0, k1) in ((offs[3], 0, k_cut), (f_retune, k_cut, n_tot)):
        ct, incr = OC.xlating_composite(taps, D, f, fs)
        inc = np.array([incr], dtype=np.complex64)
        sg = np.empty(k1 - k0, dtype=np.complex64)
        L.ro_xlating_fir_ccc(base.ctypes.data_as(fp), k0, k1 - k0, D, ct.view(np.float32).ctypes.data_as(fp),
                             T, inc.view(np.float32).ctypes.data_as(fp), C.byref(st),
                             sg.view(np.float32).ctypes.data_as(fp), 1)
        yo[k0:k1] = sg
    assert len(ys[ids[3]]) == n_tot and rel_rms(ys[ids[3]], yo) < 1e-5


def _pfb_proto(fs, nb):
    # SURVEY 8(d) cfg2 prototype: low_pass_2 rule, fc = bin/2.5, tw = bin/5, 60 dB, Blackman-Harris
    bw = fs / nb
    return G.low_pass_2(1.0, fs, bw * 0.4, bw * 0.2, 60.0, G.WIN_BLACKMAN_HARRIS)


@pytest.mark.parametrize("nb,os_", [(256, 1), (64, 1), (128, 2), (512, 1), (1024, 2), (256, 2)])
def test_pfb_equals_exact_xlating_bank(gpu_required, nb, os_):
    nat = gpu_required
    fs = 20e6
    D = nb // os_
    taps = _pfb_proto(fs, nb) if os_ == 1 else G.low_pass_2(1.0, fs, fs / nb / 4, fs / nb / 4, 20.0)
    n_frames = 150
    rng = np.random.default_rng(nb + os_)
    x = synth.awgn(rng, D * n_frames)
    bins = [0, 1, 5, nb // 2 - 1, nb // 2, nb - 3, nb - 1]
    for k in bins[1:4]:
        x = x + synth.nbfm_carrier(len(x), fs, k * fs / nb + 2000.0, 900.0, 2500.0, 1.0).astype(np.complex64)
    x = x.astype(np.complex64)
    hist = 1 << 16
    with nat.Frontend(fs, hist_capacity=max(hist, 2 * len(taps) + 2 * nb)) as fe:
        fe.pfb_open(nb, D, taps)
        cut = D * 37 + 11
        fe.push(x[:cut])                               # frames straddle a block boundary
        fe.push(x[cut:])
        assert fe.pfb_produced() == n_frames
        got = {k: fe.pfb_read_bin(k) for k in bins}
    for k in bins:
        f0 = k * fs / nb if k < nb // 2 else (k - nb) * fs / nb
        want = G.xlating_fir_exact(x, D, taps, f0, fs)
        assert len(got[k]) == n_frames
        scale = np.sqrt(np.mean(np.abs(want) ** 2))
        err = np.sqrt(np.mean(np.abs(got[k] - want) ** 2))
        assert err / max(scale, 1e-3) < 2e-5, (k, err, scale)


def test_pfb_stage2_fm_channel(gpu_required):
    """BASELINE config 2 shape: 256-bin PFB @20 Msps, stage-2 xlating FIR /3 + discriminator."""
    nat = gpu_required
    fs, nb = 20e6, 256
    x, meta = synth.cfg2(n=256 * 1500, n_active=4)
    taps = _pfb_proto(fs, nb)
    assert len(taps) == 3491
    with nat.Frontend(fs) as fe:
        fe.pfb_open(nb, nb, taps)
        chans = []
        for c in meta["carriers"]:
            b = c["bin"] % nb
            chans.append((c, b, fe.pfb_chan_open(b, 12500, c["delta"])))
        fe.push(x)
        out = [(c, b, fe.chan_read_iq(cid), fe.chan_read_fm(cid, 1.0), fe.chan_info(cid)) for c, b, cid in chans]
    bin_rate = fs / nb
    D2, taps2 = G.channel_params(bin_rate, 12500)
    assert D2 == 3 and len(taps2) == 11
    for c, b, y, fm, info in out:
        assert info["decim"] == 3 and info["ntaps"] == 11
        f0 = c["bin"] * fs / nb
        stage1 = G.xlating_fir_exact(x, nb, taps, f0, fs).astype(np.complex64)
        yo = G.xlating_fir_ccc(stage1, D2, taps2, c["delta"], bin_rate)
        fo = G.quadrature_demod_cf(yo, 1.0)
        assert len(y) == len(yo) == 500
        assert rel_rms(y, yo) < 2e-5
        assert rms(fm[20:], fo[20:]) < 1e-4


@pytest.mark.parametrize("N,F,L", [(256, 40, 10), (1024, 25, 7), (4096, 12, 5), (16384, 12, 4)])
def test_scan_chain_small(gpu_required, N, F, L):
    nat = gpu_required
    fs = 2.4e6
    rng = np.random.default_rng(N)
    x = synth.awgn(rng, N * F).astype(np.complex64)
    x += (4.0 * np.exp(2j * math.pi * 0.123 * np.arange(N * F))).astype(np.complex64)
    with nat.Frontend(fs, hist_capacity=1 << 16) as fe:
        fe.scan_start(N, F, L)
        cut = N * 3 + 17
        fe.push(x[:cut])
        assert fe.scan_result() is None
        fe.push(x[cut:])
        got = fe.scan_result()
    want = OC.scan_chain(x, N, F, L)
    assert got is not None and got.shape == (N,)
    assert np.abs(got - want).max() < 2e-3
    assert np.argmax(got) == np.argmax(want)


def test_scan_reference_size_and_peaks_bit_exact(gpu_required):
    """fs=2.4e6, N=16384, 1000 frames, 100-frame average, then the peak pick: indices bit-exact."""
    nat = gpu_required
    fs, N, fc = 2.4e6, 16384, 855.05e6
    carriers = SCAN_CARRIERS_3004
    tile = synth.scan_stream(fs, N, 125, carriers, seed=3004)
    with nat.Frontend(fs, block_capacity=N * 125, hist_capacity=1 << 16) as fe:
        fe.scan_start(N, 1000, 100)
        for _ in range(8):
            fe.push(tile)
        spec = fe.scan_result()
        lines_dev, mean_dev, _ = fe.scan_find_peaks()
    assert spec is not None
    x = np.tile(tile, 8)
    want = OC.scan_chain(x, N, 1000, 100)
    assert np.abs(spec - want).max() < 5e-3
    l_want, f_want = P.peak_detect_scipy(want, fs, fc)
    l_got, f_got = P.peak_detect_scipy(spec, fs, fc)
    np.testing.assert_array_equal(l_got, l_want)               # GPU spectrum -> same indices
    np.testing.assert_array_equal(lines_dev, l_want)           # product peak picker -> same indices
    assert list(l_want) == [2004, 5195, 9089, 15000]   # 20 kHz-wide carrier at 11000 fails the width window
    assert [nat.peak_frequency(l, fs, N, fc) for l in lines_dev] == f_want


@pytest.mark.parametrize("N,F,L", [(32768, 6, 3), (65536, 5, 2), (131072, 4, 2), (262144, 3, 2), (524288, 3, 2),
                                   (1048576, 5, 3)])
def test_scan_four_step_fft(gpu_required, N, F, L):
    """N > 16384: the four-step (N1 x N2) scan FFT against the oracle's radix-2 chain."""
    nat = gpu_required
    fs = 100e6
    rng = np.random.default_rng(N % 1000)
    x = synth.awgn(rng, N * F).astype(np.complex64)
    x += (3.0 * np.exp(2j * math.pi * 0.31 * np.arange(N * F))).astype(np.complex64)
    with nat.Frontend(fs, block_capacity=N * 2, hist_capacity=N) as fe:
        fe.scan_start(N, F, L)
        pos = 0
        step = N * 2 - 12345                      # frames straddle block boundaries
        while pos < len(x):
            fe.push(x[pos:pos + step])
            pos += step
        got = fe.scan_result()
    want = OC.scan_chain(x, N, F, L)
    assert got is not None and got.shape == (N,)
    assert np.abs(got - want).max() < 3e-3
    assert np.argmax(got) == np.argmax(want)


def test_scan_1m_point_peaks_bit_exact(gpu_required):
    """BASELINE configs[2] shape: 1M-point scan at 100 Msps, 12 carriers -> exactly 12 indices."""
    nat = gpu_required
    fs, N, fc = 100e6, 1 << 20, 860e6
    rng = np.random.default_rng(3003)
    centres = [40000 + 80000 * i + int(rng.integers(-3000, 3000)) for i in range(12)]
    carriers = [(c, float(rng.uniform(4000, 9000)), 45.0) for c in centres]
    tile = synth.scan_stream(fs, N, 8, carriers, seed=3003)
    F, L = 16, 8
    with nat.Frontend(fs, block_capacity=N * 8, hist_capacity=N) as fe:
        fe.scan_start(N, F, L)
        fe.push(tile)
        fe.push(tile)
        spec = fe.scan_result()
        lines_dev, _, _ = fe.scan_find_peaks(cap=4096)
    want = OC.scan_chain(np.tile(tile, 2), N, F, L)
    # two different float32 FFTs (oracle radix-2 vs four-step radix-16): 65 dB of dynamic range puts the
    # deepest bins ~1e-2 apart in sum-of-log10 units; the indices below are what must be bit-exact
    assert np.abs(spec - want).max() < 2e-2
    assert np.sqrt(np.mean((spec - want) ** 2)) < 2e-4
    l_want, f_want = P.peak_detect_scipy(want, fs, fc)
    l_got, _ = P.peak_detect_scipy(spec, fs, fc)
    np.testing.assert_array_equal(l_got, l_want)
    np.testing.assert_array_equal(lines_dev, l_want)
    assert len(l_want) == 12
    assert all(abs(int(a) - b) < 40 for a, b in zip(l_want, centres))


def test_p25_prefilter_chain_on_channel_output(gpu_required):
    """p25_control_demod.py:106-108,120-121 -- the channel stream goes through
    freq_xlating_fir_filter_ccc(1, low_pass_2(1, 25000, 6250, 500, 30, WIN_BLACKMAN), 0, 25000) (69 taps)
    and then quadrature_demod_cf(25000 / (2 pi 600)).  Here: a derived channel fed by a channel's ring."""
    nat = gpu_required
    x, meta = synth.cfg1(seconds=0.3, seed=1717)
    pre = G.low_pass_2(1.0, 25000.0, 6250.0, 500.0, 30.0, G.WIN_BLACKMAN)
    assert len(pre) == 69
    np.testing.assert_allclose(nat.design_low_pass_2(1.0, 25000.0, 6250.0, 500.0, 30.0, nat.WIN_BLACKMAN), pre,
                               rtol=0, atol=2e-9)
    gain = G.p25_fm_gain(25000.0)
    with nat.Frontend(meta["fs"]) as fe:
        c1 = fe.chan_open(12500, meta["offset"])
        c2 = fe.chan_open_taps(c1, 1, pre, 0.0)
        assert fe.chan_info(c2)["out_rate"] == 25000.0 and fe.chan_info(c2)["ntaps"] == 69
        for part in np.array_split(x, 7):                  # uneven blocks: the chain is block-cut invariant
            fe.push(part)
        y2 = fe.chan_read_iq(c2)
        fm = fe.chan_read_fm(c2, gain)
    yo, _ = oracle_channel(x, meta["fs"], 12500, meta["offset"], [])
    y2o = G.xlating_fir_ccc(yo, 1, pre, 0.0, 25000.0)
    fo = G.quadrature_demod_cf(y2o, gain)
    assert len(y2) == len(y2o) == len(yo)
    assert rel_rms(y2, y2o) < 1e-5
    assert rms(fm, fo) < 1e-4


@pytest.mark.parametrize("fmt_name,dtype,scale,offset", [
    ("FMT_U8", np.uint8, 1.0 / 128.0, 127.4),          # rtl-sdr wire format
    ("FMT_S16", np.int16, 1.0 / 2048.0, 0.0),          # bladeRF sc16 Q11
    ("FMT_S8", np.int8, 1.0 / 128.0, 0.0),
])
def test_wire_format_ingest_equals_host_conversion(gpu_required, fmt_name, dtype, scale, offset):
    """converting the SDR's native samples on the GPU gives bit-identical cf32 to the
    host conversion (float(raw) - offset) * scale, hence identical channel output."""
    nat = gpu_required
    rng = np.random.default_rng(44)
    info = np.iinfo(dtype)
    n = 96 * 300 + 1                                          # odd length: exercises the tail path
    raw = rng.integers(info.min, info.max + 1, size=2 * n).astype(dtype)
    x = ((raw.astype(np.float32) - np.float32(offset)) * np.float32(scale)).view(np.complex64)
    with nat.Frontend(2.4e6) as fe:
        cid = fe.chan_open(12500, 200000.0)
        fe.push_raw(raw[: 2 * 7001], getattr(nat, fmt_name), scale, offset)
        fe.push_raw(raw[2 * 7001:], getattr(nat, fmt_name), scale, offset)
        y_raw = fe.chan_read_iq(cid)
    with nat.Frontend(2.4e6) as fe:
        cid = fe.chan_open(12500, 200000.0)
        fe.push(x)
        y_ref = fe.chan_read_iq(cid)
    np.testing.assert_array_equal(y_raw.view(np.float32), y_ref.view(np.float32))
    yo, _ = oracle_channel(x, 2.4e6, 12500, 200000.0, [])
    assert rel_rms(y_raw, yo) < 1e-5


def test_p25_c4fm_front_half_symbol_filter_and_drift_probe(gpu_required):
    """pre-filter -> quadrature_demod_cf(gain) -> fir_filter_fff(1, (1/5,)*5)
    (p25_control_demod.py:106-133), plus the 10000-sample drift probe (:123-127)."""
    nat = gpu_required
    x, meta = synth.cfg1(seconds=0.6, seed=2121)
    pre = G.low_pass_2(1.0, 25000.0, 6250.0, 500.0, 30.0, G.WIN_BLACKMAN)
    gain = G.p25_fm_gain(25000.0)
    sps = 25000 // 4800
    coeffs = np.full(sps, 1.0 / sps, dtype=np.float32)
    with nat.Frontend(meta["fs"]) as fe:
        c1 = fe.chan_open(12500, meta["offset"] + 40.0)          # 40 Hz off: a visible DC term in fm
        c2 = fe.chan_open_taps(c1, 1, pre, 0.0)
        fe.chan_fm_filter(c2, gain, coeffs)
        for part in np.array_split(x, 5):
            fe.push(part)
        sym = fe.chan_read_sym(c2)
        fm = fe.chan_read_fm(c2, gain)
        level = fe.chan_fm_level(c2, gain, 10000)
    yo, _ = oracle_channel(x, meta["fs"], 12500, meta["offset"] + 40.0, [])
    y2o = G.xlating_fir_ccc(yo, 1, pre, 0.0, 25000.0)
    fo = G.quadrature_demod_cf(y2o, gain)
    so = np.convolve(fo.astype(np.float64), coeffs.astype(np.float64))[: len(fo)].astype(np.float32)
    assert len(sym) == len(so) == len(fm)
    assert rms(fm, fo) < 1e-4
    assert rms(sym, so) < 1e-4
    want_level = float(np.mean(fo[-10000:].astype(np.float64)))
    assert abs(level - want_level) < 1e-4
    assert abs(want_level - gain * 2 * math.pi * (-40.0) / 25000.0) < 0.02     # discriminator DC of a -40 Hz offset
