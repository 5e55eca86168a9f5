#!/usr/bin/env python3
"""Direct GR-faithful xlating-FIR bank micro-benchmark: C channels (channel.py parameters) over a
resident 20 Msps block.  Reports kernel ms, TFLOP/s (8 T flop per output) and how many channels the
bank sustains in real time.  env: C (channels), FS, CR, BLOCK."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radiocapture-rf_amd")]
import numpy as np
from rcf import native

C_ = int(os.environ.get("C", 64)); fs = float(os.environ.get("FS", 20e6)); cr = int(os.environ.get("CR", 12500))
B = int(os.environ.get("BLOCK", 1 << 22)); steps = int(os.environ.get("STEPS", 5))
D, T = native.channel_params(fs, cr)
fe = native.Frontend(fs, block_capacity=B, hist_capacity=1 << 16, out_capacity=1 << 15)
rng = np.random.default_rng(3)
tile = (rng.standard_normal(1 << 20) + 1j * rng.standard_normal(1 << 20)).astype(np.complex64)
for _ in range(2):
    for at in range(0, B, len(tile)):
        fe.ingest_write(tile[: min(len(tile), B - at)], at)
    fe.commit(B)
offs = np.linspace(-0.45, 0.45, C_) * fs
ids = [fe.chan_open(cr, float(np.round(o / 6250) * 6250)) for o in offs]
for _ in range(2): fe.commit(B)
fe.timing_enable(True); fe.timing_read(native.T_FIR); fe.timing_read(native.T_FIR_MFMA); fe.timing_read(native.T_DISC)
import time
fe.sync(); t0 = time.perf_counter()
for _ in range(steps): fe.commit(B)
fe.sync(); wall = (time.perf_counter() - t0) / steps * 1e3
ms, n = fe.timing_read(native.T_FIR); mms, mn = fe.timing_read(native.T_FIR_MFMA); dms, dn = fe.timing_read(native.T_DISC)
kind = 'matrix-core' if mn else 'vector'
ms = (ms + mms) / max(n, mn)
n_out = B // D
flop = 8.0 * T * n_out * C_
print(kind, "C=%d fs=%.0f D=%d T=%d block=%d: fir %.3f ms (%.1f TFLOP/s, %.1f%% of 157 TF), disc %.3f ms; "
      "real-time channels at this fs: %.0f; wall %.3f ms/block%s" % (C_, fs, D, T, B, ms, flop / (ms * 1e-3) / 1e12,
      flop / (ms * 1e-3) / 157.3e12 * 100, dms / max(dn, 1), C_ * (B / fs) / (ms * 1e-3), wall,
      "  [exact rotator]" if os.environ.get("RCF_ROTATOR") == "exact" else ""))
