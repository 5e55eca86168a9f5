#!/usr/bin/env python3
"""PCIe-inclusive ingest rate: rcf_push_iq (cf32, 8 B/sample) and rcf_push_raw (u8 IQ, 2 B/sample) from host memory
through the bench's 256-bin filterbank.  env: BLOCK (samples per push), STEPS."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radiocapture-rf_amd")]
import numpy as np
from rcf import native

B = int(os.environ.get("BLOCK", 1 << 24)); steps = int(os.environ.get("STEPS", 8)); fs, nb = 20e6, 256
bw = fs / nb
taps = native.design_low_pass_2(1.0, fs, 0.4 * bw, 0.2 * bw, 60.0, native.WIN_BLACKMAN_HARRIS)
cap = 1
while cap < 2 * (B // nb): cap <<= 1
rng = np.random.default_rng(0)
x = (rng.standard_normal(B) + 1j * rng.standard_normal(B)).astype(np.complex64)
u8 = rng.integers(0, 256, size=2 * B, dtype=np.uint8)
for name, push in (("cf32 rcf_push_iq", lambda fe: fe.push(x)),
                   ("u8   rcf_push_raw", lambda fe: fe.push_raw(u8, native.FMT_U8, 1.0 / 127.5, 127.5))):
    fe = native.Frontend(fs, block_capacity=B, hist_capacity=1 << 16, out_capacity=cap)
    fe.pfb_open(nb, nb, taps)
    push(fe); push(fe); fe.sync()
    t0 = time.perf_counter()
    for _ in range(steps): push(fe)
    fe.sync()
    dt = (time.perf_counter() - t0) / steps
    print("%s: %.2f ms per %d-sample block = %.2f Gsamples/s (pageable host memory, filterbank included)"
          % (name, dt * 1e3, B, B / dt / 1e9))
    fe.close()
