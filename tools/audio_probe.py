#!/usr/bin/env python3
"""Analog voice chain cost: C channels (channel.py parameters, 25 kS/s each) over a resident 20 Msps block, every
channel with the logging_receiver analog chain attached.  env: C, BLOCK, STEPS."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radiocapture-rf_amd")]
import numpy as np
from rcf import native, audio

C_ = int(os.environ.get("C", 256)); fs = 20e6; cr = 12500
B = int(os.environ.get("BLOCK", 1 << 22)); steps = int(os.environ.get("STEPS", 5))
fe = native.Frontend(fs, block_capacity=B, hist_capacity=1 << 16, out_capacity=1 << 15)
rng = np.random.default_rng(3)
tile = (rng.standard_normal(1 << 20) + 1j * rng.standard_normal(1 << 20)).astype(np.complex64)
for _ in range(2):
    for at in range(0, B, len(tile)):
        fe.ingest_write(tile[: min(len(tile), B - at)], at)
    fe.commit(B)
ids = [fe.chan_open(cr, float(np.round(o / 6250) * 6250)) for o in np.linspace(-0.45, 0.45, C_) * fs]
params = audio.analog_chain_params(25000)
for c in ids:
    fe.chan_audio_open(c, **params)
for _ in range(2): fe.commit(B)
fe.timing_enable(True)
for t in (native.T_FIR, native.T_FIR_MFMA, native.T_DISC, native.T_AUDIO): fe.timing_read(t)
for _ in range(steps): fe.commit(B)
ms, n = fe.timing_read(native.T_AUDIO); mms, mn = fe.timing_read(native.T_FIR_MFMA)
a, u = fe.chan_audio_produced(ids[0])
print("C=%d block=%d (%.2f s of signal): audio chain %.3f ms per block, bank %.3f ms; audio samples so far %d (ungated %d)"
      % (C_, B, B / fs, ms / n, mms / max(mn, 1), a, u))
