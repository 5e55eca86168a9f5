"""the oracle's a4 / a6 / a8 arithmetic restates GNU Radio / VOLK ("parity unpinned"); this module
BOUNDS how far each detail it cannot pin could move a judged output (oracle/unpinned.py):  log2 polynomial -> peak
indices unchanged; atan table literals -> fm; FIR summation order -> IQ; rotator FMA contraction -> fm and IQ drift.
DESIGN.md 2 quotes the table (tools/unpinned_bounds.py prints it)."""
import numpy as np
import pytest

from oracle import cbind as OC
from oracle import grspec as G
from oracle import unpinned as U
from rcf import synth


ROTATOR_OFFSETS = (5.0125e6, 62500.0, 1234567.0, -7699667.0)


def _scan_stream(fs, N, seed, n_car):
    rng = np.random.default_rng(seed)
    step = N // (n_car + 1)
    hz = fs / N
    # occupied widths inside find_peaks' [3 kHz, 30 kHz] window (fft_peak_detection.py:46-47)
    carriers = [(step * (i + 1) + int(rng.integers(-step // 8, step // 8)), float(rng.uniform(5000, 12000)), 45.0)
                for i in range(n_car)]
    return synth.scan_stream(fs, N, 16, carriers, seed=seed), carriers


@pytest.mark.parametrize("N,fs,n_car", [(16384, 2.4e6, 5), (1 << 20, 100e6, 12)])
def test_log2_polynomial_error_cannot_move_a_peak_index(N, fs, n_car):
    """bit-exact channel indices (north_star) do not depend on which volk_32f_log2 kernel ran: +-2e-4 on every log2 value
    (70x VOLK's degree-6 polynomial error), in the worst sign patterns, leaves the index list as it is"""
    x, carriers = _scan_stream(fs, N, 3004 if N == 16384 else 3003, n_car)
    ref, got, shift = U.peaks_under_log2_error(x, N, fs, 855e6)
    assert len(ref) >= n_car - 2 and len(ref) >= 3          # the stream has findable carriers: not an empty list kept empty
    for name, idx in got.items():
        assert idx == ref, name
    # the summed spectrum itself moved by ~100 x err / log2(10) plus what the float32 running sum rounds differently
    assert 1e-3 < shift < 0.05, shift


def test_log2_margin_is_orders_of_magnitude():
    x, _ = _scan_stream(2.4e6, 16384, 3004, 5)
    e = U.log2_error_that_moves_a_peak(x, 16384, 2.4e6, 855e6, start=1e-3, stop=0.3)
    assert e is None or e >= 1e-2, e                   # indices first move somewhere above 1e-2: 3000x VOLK's error


def test_atan_table_literals_cannot_move_fm_past_the_bar():
    """fm <= 1e-4 rms (north_star): every table literal off by one unit of its last printed digit moves the P25-gain
    discriminator output by < 1e-5 at most"""
    x, meta = synth.cfg1(seconds=0.2)
    D, taps = G.channel_params(meta["fs"], 12500)
    y = G.xlating_fir_ccc(x, D, taps, meta["offset"], meta["fs"])
    gain = G.p25_fm_gain(25000.0)
    worst = U.fm_under_table_perturbation(y, gain)
    assert 0 < worst < 1e-5, worst


@pytest.mark.parametrize("fs,f0", [(2.4e6, -62500.0), (20e6, 5.0125e6)])
def test_fir_summation_order_stays_inside_the_iq_bar(fs, f0):
    """IQ <= 1e-5 relative rms is the bar of the -m gpu parity tests: sequential, 8-lane, 16-lane and pairwise float32
    sums of the same products stay within 2e-6 of each other and of the float64 sum (T = 349 and 2909)"""
    D, taps = G.channel_params(fs, 12500)
    ct, _ = OC.xlating_composite(taps, D, f0, fs)
    rng = np.random.default_rng(11)
    x = synth.awgn(rng, D * 600)
    x = (x + 3.0 * np.exp(2j * np.pi * (f0 + 900.0) * np.arange(len(x)) / fs)).astype(np.complex64)
    vs64, between, _ = U.iq_under_summation_orders(x, D, ct)
    assert max(vs64.values()) < 2e-6, vs64
    assert between < 2e-6, between


def test_rotator_fma_contraction_is_invisible_to_the_discriminator():
    """a GNU Radio built with FMA contraction turns the IQ stream by a slowly growing common phase (reported in DESIGN 2)
    but the phase STEP per output -- all the discriminator sees -- differs by < 1e-6 rad: x P25 gain 6.63 = 7e-6 << 1e-4"""
    taps = G.channel_params(20e6, 12500)[1]
    for f0 in ROTATOR_OFFSETS:                          # on-grid (incr = (+-1, delta)) and generic increments
        _, inc = OC.xlating_composite(taps, 800, f0, 20e6)
        r = U.rotator_fma_drift(inc, 1_000_000)
        assert r["max_step_difference_rad"] < 1e-6, (f0, r)
        assert r["max_phase_difference"] < 1e-3, (f0, r)    # drift of the common phase over 10^6 outputs (40 s of signal)
        assert r["magnitude_excursion_unfused"] < 1e-4 and r["magnitude_excursion_fused"] < 1e-4


# ------------------------------------------------------------------------------------------- round 5: f-2, routing budget
def _voice_stream():
    from oracle import audio as A
    x, meta = synth.cfg1(seconds=0.4)
    D, taps = G.channel_params(meta["fs"], 12500)
    y = G.xlating_fir_ccc(x, D, taps, meta["offset"], meta["fs"])
    return A.analog_chain(y, 25000.0, stages=True)


def test_voice_chain_details_stay_inside_the_audio_bar():
    """audio <= 1e-4 rms (north_star).  What the f-2 restatement cannot pin -- the order of fm_deemph's three products and
    the accumulator's type, pm_remez's grid density / convergence, the last bit of the resampler's Kaiser taps -- moves
    the 8 kHz audio by < 1e-5 rms each (signal rms 0.8)"""
    st = _voice_stream()
    forms, scale = U.audio_under_deemph_forms(st["fm"], 25000.0)
    assert 0.3 < scale < 2.0
    assert forms["feedback_first_double"] < 1e-7 and forms["transposed_df2_double"] < 1e-7, forms
    assert forms["float32_accumulator"] < 1e-6, forms                 # even a build that dropped the double accumulator
    dens, tap_move, _ = U.audio_under_remez_density(st["deemph"], 25000.0)
    assert max(dens.values()) < 1e-5, dens                            # the taps themselves move by a few 1e-4 (of 4): in the stop band
    assert max(tap_move.values()) < 1e-3, tap_move
    worst, _ = U.audio_under_resampler_tap_rounding(st["hpf"], 25000.0)
    assert worst < 1e-6, worst


def test_tap_leakage_prediction_bounds_the_measured_bin_error_also_for_a_double_fwT0_build():
    """frontend_mode = 'pfb' serves a request from a bank bin when gain * margin * |g|_2 * env (rcf_pfb_tap_leakage, margin
    2.5) is inside its budget.  Measured here on the CPU for bins across the band, in the SURVEY 8(d) cfg2 environment:
    the discriminator difference bin vs GNU Radio's channel is <= margin x the prediction -- with GNU Radio's tap phases as
    3.8 computes them AND as a build that keeps fwT0 in double would (the unpinned detail of this row)"""
    import math
    from rcf import native
    fs, NB = 20e6, 1600
    D, taps = G.channel_params(fs, 12500)
    n = D * 260
    rng = np.random.default_rng(5)
    gain = G.p25_fm_gain(25000.0)
    amp = synth.snr_amp(30.0, 12500.0, fs)
    worst = 0.0
    for k in (3, 161, 401, 641, 797, NB - 500):
        f0 = (k if k < NB // 2 else k - NB) * fs / NB
        x = synth.awgn(rng, n).astype(np.complex128)
        for j in range(32):
            f = f0 if j == 0 else float(rng.integers(-780, 780)) * 12500.0
            x += synth.nbfm_carrier(n, fs, f, 1000.0 + 37 * j, 2500.0, amp, phase0=float(rng.uniform(0, 6.28)))
        x = x.astype(np.complex64)
        leak, _ = native.pfb_tap_leakage(fs, NB, taps, k)
        pred = gain * leak * math.sqrt(float(np.mean(np.abs(x) ** 2)) / amp ** 2)
        for mode in ("float_product", "double_fwT0"):
            e = U.bin_fm_error_vs_gr(x, fs, NB, taps, D, k, gain, mode)
            worst = max(worst, e / pred)
            assert e <= 2.5 * pred + 2e-7, (k, mode, e, pred)
    assert 0.3 < worst <= 2.5, worst
