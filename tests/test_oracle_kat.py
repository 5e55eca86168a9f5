"""Known-answer tests pinning the CPU oracle (list 1).

The reference ships no tests or fixtures ("parity unpinned"), so the oracle is pinned analytically:
tap counts, symmetry, DC gain, tone translation, discriminator constant, window values, FFT bin
placement, moving-sum behaviour -- and by agreement between the two independent restatements
(oracle/grspec.py numpy vs oracle/rcf_oracle.c plain C).
"""
import math

import numpy as np
import pytest

from oracle import grspec as G
from oracle import cbind as OC


# (fs, cr) -> (D, T) table: SURVEY.md 8(a) row a3 / BASELINE.md section 2
TABLE = [
    (2.0e6, 12500, 80, 291), (2.4e6, 12500, 96, 349), (2.4e6, 6250, 192, 699),
    (2.4e6, 25000, 48, 175), (8e6, 12500, 320, 1163), (10e6, 12500, 400, 1455),
    (12e6, 12500, 480, 1745), (16e6, 12500, 640, 2327), (20e6, 12500, 800, 2909),
    (20e6, 6250, 1600, 5819), (25e6, 12500, 1000, 3637),
]


@pytest.mark.parametrize("fs,cr,D,T", TABLE)
def test_channel_params_table(fs, cr, D, T):
    d, taps = G.channel_params(fs, cr)
    assert d == D and len(taps) == T
    assert taps.dtype == np.float32
    np.testing.assert_array_equal(taps, taps[::-1])                  # linear phase
    assert abs(float(taps.astype(np.float64).sum()) - 1.0) < 1e-6    # DC gain 1


def test_p25_prefilter_taps():
    # p25_control_demod.py:106-108: low_pass_2(1, 25000, 6250, 500, 30, WIN_BLACKMAN) -> 69 taps
    taps = G.low_pass_2(1.0, 25000.0, 6250.0, 500.0, 30.0, G.WIN_BLACKMAN)
    assert len(taps) == 69
    assert abs(float(taps.astype(np.float64).sum()) - 1.0) < 1e-6


def test_nonintegral_decimation_rejected():
    with pytest.raises(ValueError):
        G.channel_params(10666666, 12500)      # config_denver_massive_p25.py:20 -> 426.5


def test_windows():
    w = G.hamming(349)
    assert abs(w[0] - 0.08) < 1e-6 and abs(w[-1] - 0.08) < 1e-6 and abs(w[174] - 1.0) < 1e-6
    b = G.blackman(69)
    assert abs(b[0]) < 1e-6 and abs(b[34] - 1.0) < 1e-6
    bh = G.blackman_harris(16385)
    assert abs(bh[0] - 6e-5) < 1e-6 and abs(bh[8192] - 1.0) < 1e-6
    for wt, n in ((G.WIN_HAMMING, 349), (G.WIN_BLACKMAN, 69), (G.WIN_BLACKMAN_HARRIS, 16384)):
        np.testing.assert_allclose(OC.window(wt, n), G.window(wt, n), rtol=0, atol=1.2e-7)


def test_c_and_numpy_taps_agree():
    for fs, cr, _, _ in TABLE:
        a = G.low_pass_2(1.0, fs, cr / 2, cr / 2, 20.0, G.WIN_HAMMING)
        b = OC.low_pass_2(1.0, fs, cr / 2, cr / 2, 20.0, G.WIN_HAMMING)
        assert len(a) == len(b)
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-9)


def test_composite_taps_float32_phase():
    D, taps = G.channel_params(20e6, 12500)
    ct, incr = G.xlating_composite(taps, D, 5.0125e6, 20e6)
    ct2, incr2 = OC.xlating_composite(taps, D, 5.0125e6, 20e6)
    np.testing.assert_allclose(ct.view(np.float32), ct2.view(np.float32), rtol=0, atol=3e-9)
    assert abs(incr - incr2) < 2e-7
    # the float32 phase differs measurably from the exact one: that IS the spec
    i = np.arange(len(taps))
    exact = taps * np.exp(1j * 2 * math.pi * 5.0125e6 / 20e6 * i)
    err = np.abs(ct - exact).max() / np.abs(taps).max()
    assert 1e-6 < err < 2e-3


def test_tone_translation_and_dc_gain():
    fs, cr = 2.4e6, 12500
    D, taps = G.channel_params(fs, cr)
    f0, delta = -62500.0, 1000.0
    n = 96 * 600
    t = np.arange(n) / fs
    x = np.exp(2j * math.pi * (f0 + delta) * t).astype(np.complex64)
    y = G.xlating_fir_ccc(x, D, taps, f0, fs)
    y = y[8:]                                            # past the T-1 = 348 sample transient
    # tone emerges at +delta with amplitude |H(delta)|
    H = np.abs(np.sum(taps.astype(np.float64) * np.exp(-2j * math.pi * delta / fs * np.arange(len(taps)))))
    np.testing.assert_allclose(np.abs(y), H, rtol=2e-4)
    dphi = np.angle(y[1:] * np.conj(y[:-1]))
    np.testing.assert_allclose(dphi, 2 * math.pi * delta * D / fs, atol=2e-4)


def test_fm_constant_for_tone():
    # a tone at delta FM-demodulates to gain * 2 pi delta / (2 cr)
    rate, delta, gain = 25000.0, 1000.0, 5.0
    x = np.exp(2j * math.pi * delta / rate * np.arange(500)).astype(np.complex64)
    fm = G.quadrature_demod_cf(x, gain)
    assert fm[0] == 0.0                                   # x[-1] = 0 -> fast_atan2f(0,0) = 0
    np.testing.assert_allclose(fm[1:], gain * 2 * math.pi * delta / rate, atol=2e-5)
    np.testing.assert_allclose(OC.quad_demod(x, gain), fm, rtol=0, atol=1e-6)
    assert abs(G.p25_fm_gain(25000.0) - 6.6315) < 1e-4


def test_fast_atan2f_accuracy_and_octants():
    rng = np.random.default_rng(7)
    y = rng.standard_normal(20000).astype(np.float32)
    x = rng.standard_normal(20000).astype(np.float32)
    a = G.fast_atan2f(y, x)
    assert np.abs(a - np.arctan2(y.astype(np.float64), x.astype(np.float64))).max() < 1.0e-5
    c = np.array([OC.lib().ro_fast_atan2f(float(yy), float(xx)) for yy, xx in zip(y[:2000], x[:2000])],
                 dtype=np.float32)
    np.testing.assert_allclose(c, a[:2000], rtol=0, atol=5e-7)
    # axes and origin
    for yy, xx, want in ((0, 1, 0.0), (1, 0, math.pi / 2), (0, -1, math.pi), (-1, 0, -math.pi / 2),
                         (0, 0, 0.0)):
        assert abs(float(G.fast_atan2f(np.float32(yy), np.float32(xx))) - want) < 1e-6


def test_rotator_matches_c_bitwise():
    D, taps = G.channel_params(2.4e6, 12500)
    ct, incr = G.xlating_composite(taps, D, -62500.0, 2.4e6)
    n = 1500                                              # crosses the 512 / 1024 renormalisations
    ph, last, cnt = G.rotator_phases(incr, n)
    assert cnt == n
    # drive the C FIR with an impulse train so that v[n] == ct[0] for every n -> y[n] = ct[0]*ph[n]
    x = np.zeros((n - 1) * D + 1, dtype=np.complex64)
    x[::D] = 1.0
    ctaps0 = np.zeros_like(ct)
    ctaps0[0] = 1.0
    y, _ = OC.channel_bank(x, D, ctaps0[None, :], np.array([incr]))
    np.testing.assert_array_equal(y[0].view(np.float32), ph.view(np.float32))
    assert abs(abs(complex(ph[-1])) - 1.0) < 1e-4


def test_xlating_numpy_vs_c():
    rng = np.random.default_rng(11)
    fs, cr = 2.4e6, 12500
    D, taps = G.channel_params(fs, cr)
    x = (rng.standard_normal(96 * 300) + 1j * rng.standard_normal(96 * 300)).astype(np.complex64)
    f0 = 311000.0
    y = G.xlating_fir_ccc(x, D, taps, f0, fs)
    ct, incr = OC.xlating_composite(taps, D, f0, fs)
    yc, _ = OC.channel_bank(x, D, ct[None, :], np.array([incr]), acc_double=True)
    assert y.shape == yc[0].shape == (300,)
    scale = np.sqrt(np.mean(np.abs(y) ** 2))
    assert np.abs(y - yc[0]).max() / scale < 2e-6
    # float32-accumulating (baseline) variant stays within fp32 reorder noise
    yf, _ = OC.channel_bank(x, D, ct[None, :], np.array([incr]), acc_double=False)
    assert np.sqrt(np.mean(np.abs(yf[0] - y) ** 2)) / scale < 2e-6


def test_exact_vs_faithful_difference_is_small_at_cfg1():
    rng = np.random.default_rng(12)
    fs, cr = 2.4e6, 12500
    D, taps = G.channel_params(fs, cr)
    x = (rng.standard_normal(96 * 200) + 1j * rng.standard_normal(96 * 200)).astype(np.complex64)
    a = G.xlating_fir_ccc(x, D, taps, -62500.0, fs)
    b = G.xlating_fir_exact(x, D, taps, -62500.0, fs)
    scale = np.sqrt(np.mean(np.abs(b) ** 2))
    assert np.sqrt(np.mean(np.abs(a - b) ** 2)) / scale < 1e-4      # SURVEY: ~1e-6..1e-5 here


def test_fft_bin_placement_and_shift():
    N = 1024
    k = 37
    x = np.exp(2j * math.pi * k * np.arange(N) / N).astype(np.complex64)
    X = G.fft_vcc_shift(x[None, :], np.ones(N, dtype=np.float32))[0]
    assert np.argmax(np.abs(X)) == k + N // 2
    km = -100
    x = np.exp(2j * math.pi * km * np.arange(N) / N).astype(np.complex64)
    X = G.fft_vcc_shift(x[None, :], np.ones(N, dtype=np.float32))[0]
    assert np.argmax(np.abs(X)) == km + N // 2
    assert abs(abs(X[km + N // 2]) - N) < 1e-2


def test_nlog10_and_moving_sum():
    p = np.array([1.0, 10.0, 100.0, 0.0], dtype=np.float32)
    v = G.nlog10_ff(p, 1.0, 1.0)
    np.testing.assert_allclose(v[:3], [1.0, 2.0, 3.0], atol=1e-6)
    assert abs(v[3] - (-127.0 / math.log2(10) + 1.0)) < 1e-4          # -inf -> -127 rule
    frames = np.ones((250, 8), dtype=np.float32) * np.float32(1.25)
    ms = G.moving_sum_ff(frames, 100)
    np.testing.assert_allclose(ms[0], 1.25)
    np.testing.assert_allclose(ms[98], 1.25 * 99)
    np.testing.assert_allclose(ms[99], 125.0)
    np.testing.assert_allclose(ms[249], 125.0)                         # 100 identical frames = 100x


def test_scan_chain_numpy_vs_c():
    rng = np.random.default_rng(21)
    N, F, L = 256, 30, 10
    x = (rng.standard_normal(N * F) + 1j * rng.standard_normal(N * F)).astype(np.complex64)
    x += (3.0 * np.exp(2j * math.pi * 0.2 * np.arange(N * F))).astype(np.complex64)
    a = G.scan_chain(x, N, F, L)
    b = OC.scan_chain(x, N, F, L)
    assert np.abs(a - b).max() < 2e-3                                   # float32 FFT vs float64 FFT
    assert np.argmax(a) == int(round(0.2 * N)) + N // 2
    # exact-arithmetic identity: only the last L frames matter
    c = G.scan_chain(x[(F - L) * N:], N, L, L)
    assert np.abs(a - c).max() < 1e-3


@pytest.mark.parametrize("seed", range(40))
def test_the_two_restatements_build_the_same_composite_bits(seed):
    """numpy (grspec) and C (rcf_oracle.c) restatements of build_composite_fir at random rates and OFF-GRID offsets: the
    composite taps and the rotator increment bit for bit.  Both go through libm's float cos / sin, as GNU Radio's
    exp(gr_complex(0, x)) / gr_expj(x) do: one ulp of difference in the increment -- numpy's own float32 routines or a
    correctly rounded double both produce it on some arguments -- drifts the output phase by 6e-8 rad per output, the whole
    1e-5 IQ bar after a few hundred outputs (found by cross-checking the restatements at random offsets)."""
    from oracle import cbind as OC
    rng = np.random.default_rng(41000 + seed)
    fs = float(rng.choice([2.4e6, 8e6, 10e6, 20e6]))
    cr = int(rng.choice([6250, 12500, 25000]))
    D, taps = G.channel_params(fs, cr)
    f0 = float(rng.uniform(-0.45, 0.45) * fs)
    ct_c, incr_c = OC.xlating_composite(taps, D, f0, fs)
    ct_n, incr_n = G.xlating_composite(taps, D, f0, fs)
    np.testing.assert_array_equal(ct_c.view(np.float32), ct_n.view(np.float32))
    assert np.complex64(incr_c) == np.complex64(incr_n)


def test_scan_chain_is_the_flowgraph_the_reference_builds():
    """tests/golden/demod_params.json 'fft_vector': /root/reference/fft_vector.py's own constructor run over stand-ins --
    the blocks' arguments and who is connected to whom.  The oracle's scan_chain is that chain with those numbers:
    stream -> vectors of 16384 -> forward FFT, Blackman-Harris window, shifted -> |.|^2 -> 1 log10(.) + 1 -> moving SUM of
    100 vectors (scale 1) -> the 1000th vector only (head 1000, skiphead 999)."""
    import inspect
    import json
    import os
    fv = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "demod_params.json")))["fft_vector"]
    d = inspect.signature(G.scan_chain).parameters
    ours = (d["N"].default, d["n_frames"].default, d["avg_len"].default)
    assert ours == (fv["length"], fv["head"]["args"][1], fv["moving_average_ff"]["args"][0]) == (16384, 1000, 100)
    assert fv["fft_vcc"]["args"][0] == fv["length"] and fv["fft_vcc"]["args"][1] is True and fv["fft_vcc"]["args"][3] is True
    assert fv["window_passed_is_that_one"] and fv["window_blackmanharris"]["args"] == [fv["length"]]
    assert fv["nlog10_ff"]["args"] == [1, fv["length"], 1]
    dl = inspect.signature(G.nlog10_ff).parameters
    assert (dl["n"].default, dl["k"].default) == (1.0, 1.0)
    assert fv["moving_average_ff"]["args"][:2] == [100, 1] and fv["moving_average_ff"]["args"][3] == fv["length"]
    assert inspect.signature(G.moving_sum_ff).parameters["scale"].default == 1.0
    assert fv["skiphead"]["args"][1] == fv["head"]["args"][1] - 1            # exactly one vector leaves: the last
    nxt = dict(fv["edges"])
    chain, b = [], "zeromq_sub_source_0"
    while b in nxt:
        chain.append(b)
        b = nxt[b]
    assert chain + [b] == ["zeromq_sub_source_0", "blocks_stream_to_vector_0", "fft_vxx_0", "blocks_complex_to_mag_squared_0",
                           "blocks_nlog10_ff_0", "blocks_moving_average_xx_1", "blocks_head_0", "blocks_skiphead_0",
                           "blocks_file_sink_0"]


def test_the_timed_cpu_baseline_computes_the_channel_bank():
    """bench.py's cpu_baseline times ro_bank_bench (pinned threads, private first-touched copies of a periodic tile,
    reference structure or time-tiled): both forms must BE the channel the oracle's ro_channel_bank computes -- the last
    pass over the tile equals the tail of the bank run over the tile repeated, bit for bit (same float32 order)."""
    from oracle import cbind as OC
    fs = 2.4e6
    D, taps = G.channel_params(fs, 12500)
    rng = np.random.default_rng(5)
    n = D * 400
    tile = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    offs = [-62500.0, 100e3, 412.5e3, -1e6]
    ct = np.stack([OC.xlating_composite(taps, D, f, fs)[0] for f in offs])
    inc = np.array([OC.xlating_composite(taps, D, f, fs)[1] for f in offs], dtype=np.complex64)
    g = np.full(4, 5.0, dtype=np.float32)
    passes = 3
    want, _ = OC.channel_bank(np.tile(tile, passes + 1), D, ct[:1], inc[:1], g[:1], acc_double=False)
    for tiled, block in ((False, 0), (True, D * 16), (True, D * 7)):
        t, chk, y0 = OC.bank_bench(tile, passes + 1, D, ct, inc, g, n_threads=2, cpt=2, cpu_ids=None, tiled=tiled,
                                   tile_block=block, want_y0=True)
        assert t > 0 and np.all(np.isfinite(chk))
        # the bench's first pass sees the tile's tail as history, the bank zeros: compare the LAST pass, far from either
        assert np.array_equal(y0, want[0][-len(y0):]), (tiled, block)
