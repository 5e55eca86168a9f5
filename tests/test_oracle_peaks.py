"""Pin the find_peaks restatements (numpy + C) against the live third-party oracle
scipy.signal.find_peaks, the only arithmetic /root/reference/fft_peak_detection.py:65 delegates."""
import numpy as np
import pytest
from scipy import signal

from oracle import peaks as P
from oracle import cbind as OC


def _spectrum(rng, n, n_car, quant=None):
    x = rng.normal(100.0, 5.5, n)
    for _ in range(n_car):
        c = rng.integers(50, n - 50)
        w = rng.uniform(5, 80)
        x += rng.uniform(20, 60) * np.exp(-0.5 * ((np.arange(n) - c) / (w / 2.355)) ** 2)
    if quant:
        x = np.round(x / quant) * quant          # force plateaus / ties
    return x.astype(np.float32)


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("quant", [None, 0.5, 4.0])
def test_find_peaks_restated_equals_scipy(seed, quant):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(200, 3000))
    x = _spectrum(rng, n, int(rng.integers(0, 6)), quant)
    for (a, b, prom) in ((3.0, 40.0, 1.0), (20.48, 204.8, 1.0), (1.0, 1e9, 0.0), (2.0, 10.0, 7.5)):
        want = signal.find_peaks(x, width=[a, b], prominence=prom)[0]
        got = P.find_peaks_restated(x, a, b, prom)
        np.testing.assert_array_equal(got, want)


def test_triangle_known_width_and_prominence():
    # hand-built triangle: base 0, apex 10 at index 50, slope 1/bin -> prominence 10, width@half 10
    x = np.zeros(101, dtype=np.float32)
    x[40:61] = 10 - np.abs(np.arange(40, 61) - 50)
    pk = P.local_maxima(x.astype(np.float64))
    assert list(pk) == [50]
    prom, lb, rb = P.prominences(x.astype(np.float64), pk)
    assert prom[0] == 10.0
    w = P.widths_half(x.astype(np.float64), pk, prom, lb, rb)
    assert abs(w[0] - 10.0) < 1e-12
    assert list(P.find_peaks_restated(x, 9.0, 11.0, 1.0)) == [50]
    assert list(P.find_peaks_restated(x, 11.0, 20.0, 1.0)) == []


def test_plateau_midpoint_and_edges():
    x = np.array([0, 1, 3, 3, 3, 3, 1, 0, 5], dtype=np.float64)
    assert list(P.local_maxima(x)) == [3]            # (2+5)//2, trailing edge rise is not a peak
    assert list(signal.find_peaks(x)[0]) == [3]
    assert list(P.local_maxima(np.array([5.0, 1.0, 5.0]))) == []
    assert list(P.local_maxima(np.array([], dtype=np.float64))) == []


@pytest.mark.parametrize("seed", range(8))
def test_full_detect_three_ways(seed):
    """prologue + find_peaks + 2*mean gate: scipy-backed == numpy restatement == C restatement."""
    rng = np.random.default_rng(100 + seed)
    n = 16384
    x = _spectrum(rng, n, 6) - 480.0                   # negative minimum exercises `+ abs(min)`
    fs, fc = 2.4e6, 855e6
    l0, f0 = P.peak_detect_scipy(x, fs, fc)
    l1, f1 = P.peak_detect_restated(x, fs, fc)
    l2, mean2 = OC.peak_detect(x, fs)
    np.testing.assert_array_equal(l0, l1)
    np.testing.assert_array_equal(l0, l2)
    assert f0 == f1
    _, mean, _, _ = P.prologue(x, fs, n)
    assert mean == mean2                                # same sequential float64 sum
    if len(l0):
        hz = fs / n
        assert f0[0] == int(l0[0] * hz - fs / 2 + fc)


def _peak_goldens():
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "peaks.npz"))
    return [(i, g["spectrum_%02d" % i], [int(v) for v in g["freqs_%02d" % i]], int(g["meta"][i][0]), int(g["meta"][i][1]))
            for i in range(len(g["meta"]))]


@pytest.mark.parametrize("case", _peak_goldens(), ids=lambda c: "spectrum%02d" % c[0])
def test_detection_equals_the_references_own_run(case):
    """tests/golden/peaks.npz holds what /root/reference/fft_peak_detection.py:44-73 ITSELF found (those statements lifted
    out with ast and executed as they stand: tests/golden/make_peak_goldens.py) on quantised synthetic spectra.  The
    scipy-backed restatement, the numpy one, the C one and the product's host picker must name the same frequencies."""
    from rcf import native
    _, x, want, fs, fc = case
    l0, f0 = P.peak_detect_scipy(x, fs, fc)
    assert [int(v) for v in f0] == want
    l1, f1 = P.peak_detect_restated(x, fs, fc)
    assert [int(v) for v in f1] == want
    l2, _ = OC.peak_detect(x, fs)
    assert [native.peak_frequency(int(l), fs, len(x), fc) for l in l2] == want
    _, _, a, b = P.prologue(x, fs, len(x))
    l3, _, _ = native.find_peaks(x, a, b)
    assert [native.peak_frequency(int(l), fs, len(x), fc) for l in l3] == want
