#!/usr/bin/env python3
"""The two 3200-bin shapes of the 6.25 kHz grid at 2^25-sample blocks (steady state, HIP events)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radiocapture-rf_amd")]
import numpy as np
from rcf import native
fs = 20e6; B = 1 << 25
for cr in (6250, 12500):
    D, T = native.channel_params(fs, cr)
    taps = native.design_low_pass_2(1.0, fs, cr / 2, cr / 2, 20.0)
    fe = native.Frontend(fs, block_capacity=B, hist_capacity=1 << 16, out_capacity=1 << (16 if D == 1600 else 17))
    fe.pfb_open(3200, D, taps)
    rng = np.random.default_rng(1)
    tile = (rng.standard_normal(1 << 20) + 1j * rng.standard_normal(1 << 20)).astype(np.complex64)
    for _ in range(2):
        for at in range(0, B, len(tile)):
            fe.ingest_write(tile, at)
        fe.commit(B)
    for _ in range(100):
        fe.commit(B)
    fe.timing_enable(True, classes=[native.T_PFB]); fe.timing_read(native.T_PFB)
    for _ in range(100):
        fe.commit(B)
    ms, n = fe.timing_read(native.T_PFB); ms /= n
    alg = (8 + 8 * 3200 / D) * B
    print("3200 bins D=%d T=%d: %.4f ms  frac %.3f" % (D, T, ms, alg / (ms * 1e-3) / 8e12))
    fe.close()
