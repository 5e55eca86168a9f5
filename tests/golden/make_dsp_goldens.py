#!/usr/bin/env python3
"""Generate tests/golden/dsp_vectors.npz: seeded inputs + expected outputs of the hot path.

The reference ships no recorded IQ or expected outputs, and GNU Radio cannot be
run here, so these vectors are produced by the CPU oracle (oracle/grspec.py, float64 where GR's
rounding is noise, float32 where it is systematic) -- they pin the oracle AND the HIP path against
regressions, they do not pin either against GNU Radio ("parity unpinned").
The scan / peak expectations use the live third-party oracle scipy.signal.find_peaks.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radiocapture-rf_amd")]
from oracle import grspec as G      # noqa: E402
from oracle import peaks as P       # noqa: E402
from rcf import synth               # noqa: E402

out = {}
# 1. BASELINE configs[0] shape: one 12.5 kHz NBFM channel in 2.4 Msps (8 ms), gains 5 and P25
x, meta = synth.cfg1(seconds=0.008)
D, taps = G.channel_params(meta["fs"], 12500)
y = G.xlating_fir_ccc(x, D, taps, meta["offset"], meta["fs"])
out["cfg1_x"] = x
out["cfg1_y"] = y
out["cfg1_fm5"] = G.quadrature_demod_cf(y, 5.0)
out["cfg1_fm_p25"] = G.quadrature_demod_cf(y, G.p25_fm_gain(25000.0))
# 2. GR float32 phase arithmetic at wideband sizes: fs = 20 Msps, D = 800, T = 2909, offset 5.0125 MHz
rng = np.random.default_rng(2002)
xw = synth.awgn(rng, 800 * 12)
xw = (xw + synth.nbfm_carrier(len(xw), 20e6, 5.0125e6 + 400.0, 600.0, 2500.0, 0.7)).astype(np.complex64)
Dw, tw = G.channel_params(20e6, 12500)
out["wide_x"] = xw
out["wide_y"] = G.xlating_fir_ccc(xw, Dw, tw, 5.0125e6, 20e6)
# 3. 64-bin PFB (exact-phase bank) on 2.4 Msps, bins 0, 3, 40, 63
xp = synth.awgn(np.random.default_rng(64), 64 * 60).astype(np.complex64)
tp = G.low_pass_2(1.0, 2.4e6, 2.4e6 / 64 * 0.4, 2.4e6 / 64 * 0.2, 60.0, G.WIN_BLACKMAN_HARRIS)
out["pfb_x"] = xp
out["pfb_taps"] = tp
for k in (0, 3, 40, 63):
    f0 = k * 2.4e6 / 64 if k < 32 else (k - 64) * 2.4e6 / 64
    out["pfb_bin%d" % k] = G.xlating_fir_exact(xp, 64, tp, f0, 2.4e6).astype(np.complex64)
# 4. scan chain N = 512, 24 frames, 8-frame average + peak pick (scipy)
fs = 2.4e6
xs = synth.scan_stream(fs, 512, 24, [(100, 40000.0, 30.0), (350, 60000.0, 35.0)], seed=77)
spec = G.scan_chain(xs, 512, 24, 8)
out["scan_x"] = xs
out["scan_spec"] = spec
lines, freqs = P.peak_detect_scipy(spec, fs, 855.05e6)
out["scan_lines"] = lines
out["scan_freqs"] = np.array(freqs, dtype=np.int64)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dsp_vectors.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes;", "scan lines", lines)
