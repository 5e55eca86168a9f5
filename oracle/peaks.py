"""CPU restatement of the peak picker: /root/reference/fft_peak_detection.py:38-73.

TEST INFRASTRUCTURE ONLY (see oracle/grspec.py header for the import rule).

PINNED: tests/golden/peaks.npz holds what fft_peak_detection.py:44-73 ITSELF found on 16 synthetic spectra (those
statements lifted out with ast and executed as they stand in the build container: tests/golden/make_peak_goldens.py);
tests/test_oracle_peaks.py holds every restatement here, and the product's pickers, to those frequencies.

Two restatements live here:
  * ``peak_detect_scipy``  -- lines 38-73 of the reference with the one third-party call it makes
    (``scipy.signal.find_peaks``, available in this image) left in place.  This is the LIVE oracle:
    it pins the restatement below and the product's own C++/HIP peak picker.
  * ``find_peaks_restated`` -- a from-the-published-algorithm restatement of
    ``find_peaks(x, width=[a,b], prominence=p)`` (local maxima with plateau midpoints, prominence by
    outward walks, width at rel_height 0.5 by linear interpolation), all in float64 like scipy,
    validated against scipy on randomised inputs (tests/test_oracle_peaks.py).

numpy-version note: builtin ``sum(data)`` over float32 scalars accumulates in float64
on the reference's numpy 1.x and in float32 on numpy >= 2.  The restatement pins float64,
sequential, left to right.
"""
from __future__ import annotations

import numpy as np


def prologue(data: np.ndarray, samp_rate: float, fft_width: int):
    """fft_peak_detection.py:44-63.  returns (shifted float32 data, mean float64, min_w, max_w)."""
    data = np.array(data, dtype=np.float32, copy=True)
    hz_per_bin = samp_rate / fft_width
    min_w = 3000 / hz_per_bin
    max_w = 30000 / hz_per_bin
    data_min = data.min() if len(data) else np.float32(0)
    data = (data + np.abs(data_min)).astype(np.float32)        # :58-59, float32 + float32
    total = 0.0
    for v in data.tolist():                                    # sequential float64 accumulate (:61)
        total += v
    mean = total / len(data) if len(data) else 0.0
    return data, mean, min_w, max_w


def peak_detect_scipy(spectrum: np.ndarray, samp_rate: float, center_freq: float,
                      fft_width: int | None = None):
    """returns (lines int64[], frequencies int[]) -- the surviving `line` values and derived Hz."""
    from scipy import signal
    fft_width = len(spectrum) if fft_width is None else fft_width
    data, mean, min_w, max_w = prologue(spectrum, samp_rate, fft_width)
    hz_per_bin = samp_rate / fft_width
    peaks = signal.find_peaks(data, width=[min_w, max_w], prominence=1)
    lines, freqs = [], []
    for line in peaks[0]:
        if float(data[line]) > mean * 2:                                        # :71
            lines.append(int(line))
            freqs.append(int((int(line) * hz_per_bin) - (samp_rate / 2) + center_freq))   # :72
    return np.array(lines, dtype=np.int64), freqs


# --------------------------------------------------------------------------------------------
# own restatement of scipy.signal.find_peaks(x, width=[a,b], prominence=p)
# --------------------------------------------------------------------------------------------
def local_maxima(x: np.ndarray) -> np.ndarray:
    """Strict local maxima; flat tops report their (floor) midpoint; edges excluded."""
    n = len(x)
    out = []
    i = 1
    i_max = n - 1
    while i < i_max:
        if x[i - 1] < x[i]:
            ahead = i + 1
            while ahead < i_max and x[ahead] == x[i]:
                ahead += 1
            if x[ahead] < x[i]:
                out.append((i + ahead - 1) // 2)
                i = ahead
        i += 1
    return np.array(out, dtype=np.int64)


def prominences(x: np.ndarray, peaks: np.ndarray):
    n = len(x)
    prom = np.empty(len(peaks), dtype=np.float64)
    lb = np.empty(len(peaks), dtype=np.int64)
    rb = np.empty(len(peaks), dtype=np.int64)
    for k, p in enumerate(peaks):
        h = x[p]
        i = p
        left_min = h
        lb[k] = p
        while i >= 0 and x[i] <= h:
            if x[i] < left_min:
                left_min = x[i]
                lb[k] = i
            i -= 1
        i = p
        right_min = h
        rb[k] = p
        while i <= n - 1 and x[i] <= h:
            if x[i] < right_min:
                right_min = x[i]
                rb[k] = i
            i += 1
        prom[k] = h - max(left_min, right_min)
    return prom, lb, rb


def widths_half(x: np.ndarray, peaks: np.ndarray, prom: np.ndarray, lb: np.ndarray, rb: np.ndarray):
    w = np.empty(len(peaks), dtype=np.float64)
    for k, p in enumerate(peaks):
        height = x[p] - prom[k] * 0.5
        i = p
        while lb[k] < i and height < x[i]:
            i -= 1
        left_ip = float(i)
        if x[i] < height:
            left_ip += (height - x[i]) / (x[i + 1] - x[i])
        i = p
        while i < rb[k] and height < x[i]:
            i += 1
        right_ip = float(i)
        if x[i] < height:
            right_ip -= (height - x[i]) / (x[i - 1] - x[i])
        w[k] = right_ip - left_ip
    return w


def find_peaks_restated(x, min_w: float, max_w: float, prominence: float = 1.0) -> np.ndarray:
    x = np.asarray(x, dtype=np.float64)
    pk = local_maxima(x)
    prom, lb, rb = prominences(x, pk)
    keep = prom >= prominence
    pk, prom, lb, rb = pk[keep], prom[keep], lb[keep], rb[keep]
    w = widths_half(x, pk, prom, lb, rb)
    keep = (min_w <= w) & (w <= max_w)
    return pk[keep]


def peak_detect_restated(spectrum: np.ndarray, samp_rate: float, center_freq: float,
                         fft_width: int | None = None):
    fft_width = len(spectrum) if fft_width is None else fft_width
    data, mean, min_w, max_w = prologue(spectrum, samp_rate, fft_width)
    hz_per_bin = samp_rate / fft_width
    pk = find_peaks_restated(data, min_w, max_w, 1.0)
    lines = [int(p) for p in pk if float(data[p]) > mean * 2]
    freqs = [int((l * hz_per_bin) - (samp_rate / 2) + center_freq) for l in lines]
    return np.array(lines, dtype=np.int64), freqs
