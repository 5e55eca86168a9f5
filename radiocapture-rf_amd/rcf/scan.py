"""Scan path: /root/reference/fft_vector.py (spectrum) + /root/reference/fft_peak_detection.py:38-73
(peak pick) on the HIP library.  `python -m rcf.scan -i <index>` mirrors fft_based_scan.sh for one
source fed from a cf32 file."""
from __future__ import annotations

import numpy as np

from . import native


def spectrum(frontend, feed, fft_len=1024 * 16, n_frames=1000, avg_len=100):
    """fft_vector.py:37-60: arm the scanner, pull blocks from `feed()` (an iterator of complex64
    arrays) until n_frames frames are consumed, return the float32[fft_len] vector the reference
    writes to /tmp/fft_source_<i>."""
    frontend.scan_start(fft_len, n_frames, avg_len)
    for block in feed:
        frontend.push(block)
        out = frontend.scan_result()
        if out is not None:
            return out
    raise RuntimeError("feed ended after %d of %d frames" % (frontend.scan_frames_done(), n_frames))


def peak_detect(data, bandwidth, center_freq, fft_width=None):
    """fft_peak_detection.py:44-73 -> (lines, frequencies): width window [3 kHz, 30 kHz] in bins,
    prominence 1, `data[line] > 2*mean` gate, frequency = int(line*hz_per_bin - bandwidth/2 + center)."""
    data = np.ascontiguousarray(data, dtype=np.float32)
    fft_width = len(data) if fft_width is None else fft_width
    hz_per_bin = bandwidth / fft_width
    min_width_in_bins = 3000 / hz_per_bin
    max_width_in_bins = 30000 / hz_per_bin
    lines, mean, count = native.find_peaks(data, min_width_in_bins, max_width_in_bins, 1.0,
                                           cap=max(1024, len(data) // 8))
    freqs = [native.peak_frequency(int(l), bandwidth, fft_width, center_freq) for l in lines]
    return lines, freqs


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="fft_vector.py + fft_peak_detection.py on MI355X")
    ap.add_argument("-i", "--index", type=int, default=0)
    ap.add_argument("--file", required=True, help="raw cf32 capture of the source")
    ap.add_argument("--samp-rate", type=float, default=2400000)
    ap.add_argument("--center-freq", type=float, default=0.0)
    ap.add_argument("--fft-len", type=int, default=1024 * 16)
    args = ap.parse_args(argv)
    x = np.fromfile(args.file, dtype=np.complex64)
    step = args.fft_len * 25
    with native.Frontend(args.samp_rate, args.center_freq, block_capacity=step,
                         hist_capacity=max(1 << 16, args.fft_len)) as fe:
        spec = spectrum(fe, (x[i:i + step] for i in range(0, len(x), step)), args.fft_len)
    spec.tofile('/tmp/fft_source_%s' % args.index)
    lines, freqs = peak_detect(spec, args.samp_rate, args.center_freq, args.fft_len)
    for f in freqs:
        print('Peak %s' % int(f))


if __name__ == "__main__":
    main()
