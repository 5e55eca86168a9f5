#!/usr/bin/env python3
"""Micro-benchmark of the PFB kernel alone (HIP-event timing from librcf): prints ms and GB/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radiocapture-rf_amd")]
import numpy as np
from rcf import native

nb = int(os.environ.get("NB", 256)); osf = int(os.environ.get("OS", 1))
B = int(os.environ.get("BLOCK", 1 << 25)); steps = int(os.environ.get("STEPS", 10))
fs = float(os.environ.get("FS", 20e6))
D = nb // osf
bw = fs / nb
if nb % 25 == 0:
    # the reference's own channel filter (channel.py:31-33): every bin is one of its 25 kS/s channels
    cr = int(os.environ.get("CR", 12500))                 # CR=6250: the 6.25 kHz channel (D = 1600 at 20 Msps)
    D, T = native.channel_params(fs, cr)
    osf = nb // D
    taps = native.design_low_pass_2(1.0, fs, cr / 2.0, cr / 2.0, 20.0)
else:
    taps = native.design_low_pass_2(1.0, fs, 0.4 * bw, 0.2 * bw, 60.0, native.WIN_BLACKMAN_HARRIS) if osf == 1 \
        else native.design_low_pass_2(1.0, fs, bw / 4, bw / 4, 20.0)
frames = B // D
cap = 1
while cap < 2 * frames: cap <<= 1
cap *= int(os.environ.get("OUTCAP_MUL", 1))          # OUTCAP_MUL=2: bench.py's ring (two blocks + the stage-2 reach, rounded up)
fe = native.Frontend(fs, block_capacity=B, hist_capacity=1 << 17, out_capacity=cap)
fe.pfb_open(nb, D, taps)
rng = np.random.default_rng(1)
tile = (rng.standard_normal(1 << 20) + 1j * rng.standard_normal(1 << 20)).astype(np.complex64)
for _ in range(2):
    for at in range(0, B, len(tile)):
        fe.ingest_write(tile[: min(len(tile), B - at)], at)
    fe.commit(B)
ntap = int(os.environ.get("TAPS", 0))
# TAPSEQ=1: bins 0, 1, 2, ... (complete aligned runs of 16: read from the bank's ring); default: scattered bins (tap matrix)
seq = bool(int(os.environ.get("TAPSEQ", 0)))
tids = [fe.pfb_tap_open(i % nb if seq else (7 + 6 * i) % nb, gr_phase=bool(int(os.environ.get("GRPHASE", 1)))) for i in range(ntap)]
if int(os.environ.get("FMONLY", 0)):                     # FMONLY=1: the taps expose their discriminator only (rcf_chan_set_fm_only)
    for t in tids: fe.chan_set_fm_only(t, True)
# STAGE2=n: n stage-2 channels (xlating FIR /D2 + discriminator) on bins spread over the bank, as the timed configuration has 32
n2 = int(os.environ.get("STAGE2", 0))
s2 = [fe.pfb_chan_open((3 + (nb // max(n2, 1)) * i) % nb, 12500, 1000.0 + 10 * i) for i in range(n2)]
# FMFUSED=1|2: the discriminator of every bin in the bank's own kernel (rcf_pfb_fm_enable: 1 beside the bins ring, 2 instead
# of it); RCF_PFB5_FM_SPAN=n forces the chunks one workgroup walks
fmf = int(os.environ.get("FMFUSED", 0))
if fmf: fe.pfb_fm_enable(fmf, gr_phase=True)
for _ in range(int(os.environ.get("WARM", 3))): fe.commit(B)
fe.timing_enable(True, classes=None if os.environ.get('TIME_ALL') else [native.T_PFB]); fe.timing_read(native.T_PFB)
import time
fe.sync(); _t0 = time.perf_counter()
for _ in range(steps): fe.commit(B)
if os.environ.get("WALL"):
    fe.sync(); print("wall %.4f ms per step | " % ((time.perf_counter() - _t0) / steps * 1e3), end="")
ms, n = fe.timing_read(native.T_PFB)
ms /= n
extra = ""
if os.environ.get('TIME_ALL'):
    d2, n2 = fe.timing_read(native.T_FIR_DERIVED)
    d3, n3 = fe.timing_read(native.T_DISC)
    d4, n4 = fe.timing_read(native.T_TAPS)
    extra = "  derived-FIR %.4f ms/block  disc %.4f ms/block  tap-finalize %.4f ms/block" % (d2 / steps, d3 / steps, d4 / steps)
gbs = (8.0 * B + 8.0 * B * osf) / (ms * 1e-3) / 1e9
print("NB=%d OS=%d taps=%d remap=%s : %.4f ms  %.0f GB/s (%.1f%% of 8 TB/s)" % (
    nb, osf, len(taps),
    "off" if os.environ.get("RCF_PFB_NOREMAP") else "on", ms, gbs, gbs / 80.0) + extra + ("  taps=%d%s" % (ntap, " (consecutive)" if seq else "")))
