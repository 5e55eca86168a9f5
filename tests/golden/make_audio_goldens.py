#!/usr/bin/env python3
"""Generate tests/golden/audio_vectors.npz: a channel-rate IQ stream (25 kS/s, 0.25 s: NBFM tone, a stretch of exact
silence that closes the squelch, noise) and what the reference's analog voice chain makes of it, stage by stage
(logging_receiver.py:211-222), plus the filter designs.  Produced by the CPU oracle (oracle/audio.py; the equiripple
low-pass through scipy.signal.remez): pins the oracle and the HIP path against regressions, not against GNU Radio
("parity unpinned")."""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radiocapture-rf_amd")]
from oracle import audio as A       # noqa: E402

rate, n = 25000.0, 6250
rng = np.random.default_rng(8)
t = np.arange(n) / rate
iq = 0.4 * np.exp(1j * (2500.0 / 700.0) * np.sin(2 * math.pi * 700.0 * t))
iq = iq + 0.01 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
iq[2000:4500] = 0                                   # exact silence: the gate closes ~2150 samples in
iq = iq.astype(np.complex64)
st = A.analog_chain(iq, rate, stages=True)
out = dict(iq=iq, gated_len=np.int64(len(st["gated"])), deemph=st["deemph"], audio=st["audio"],
           lpf_taps=st["lpf_taps"], hpf_taps=st["hpf_taps"], rs_taps=A.design_resampler_taps(8, 25),
           deemph_b=np.array(A.fm_deemph_taps(rate)[0]), deemph_a=np.array(A.fm_deemph_taps(rate)[1]))
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "audio_vectors.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes; gated", len(st["gated"]), "of", n, "audio", len(st["audio"]))
