#!/usr/bin/env python3
"""Scan-path micro-benchmark (BASELINE configs[2] shape): N-point scan FFT + log-mag + running sum over
resident IQ, HIP-event kernel times, then the peak pick.  env: N (default 2^20), FRAMES, AVG, REPS (scans; the last is
timed), SAVE (npy of the emitted spectrum: RCF_SCAN_FUSED=0 / 1 runs are compared bit for bit)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radiocapture-rf_amd")]
import numpy as np
from rcf import native, synth

N = int(os.environ.get("N", 1 << 20)); F = int(os.environ.get("FRAMES", 200)); L = int(os.environ.get("AVG", 100))
fs = 100e6 if N > 16384 else 2.4e6
fpb = max(1, (1 << 25) // N)                       # frames per block (32M samples = 256 MiB)
B = N * fpb
rng = np.random.default_rng(3003)
centres = [N // 26 + (N // 13) * i for i in range(12)]
carriers = [(c, 7000.0 * fs / 100e6 if N > 16384 else 9000.0, 45.0) for c in centres]
tile = synth.scan_stream(fs, N, min(fpb, 8), carriers, seed=3003)
fe = native.Frontend(fs, 860e6, block_capacity=B, hist_capacity=max(N, 1 << 16))
for _ in range(2):
    for at in range(0, B, len(tile)):
        fe.ingest_write(tile[: min(len(tile), B - at)], at)
    fe.commit(B)
# warm-up scan (module load, attribute calls, workspace allocation), then the timed one
fe.scan_start(N, min(F, 2 * fpb), L)
while fe.scan_frames_done() < min(F, 2 * fpb):
    fe.commit(B)
fe.scan_find_peaks(cap=4096)
if not os.environ.get("NOTIME"):
    fe.timing_enable(True)
for _ in range(int(os.environ.get("REPS", 1)) - 1):        # untimed repeats: the last scan runs with the launch times settled
    fe.scan_start(N, F, L)
    while fe.scan_frames_done() < F:
        fe.commit(B)
    fe.sync()
    fe.timing_read(native.T_SCAN_FFT); fe.timing_read(native.T_SCAN_MOVSUM)
fe.scan_start(N, F, L)
fe.sync(); t0 = time.perf_counter()
while fe.scan_frames_done() < F:
    fe.commit(B)
spec = fe.scan_result()
t1 = time.perf_counter()
fft_ms, n1 = fe.timing_read(native.T_SCAN_FFT); ms_ms, n2 = fe.timing_read(native.T_SCAN_MOVSUM)
fft_ms = max(fft_ms, 1e-9)
t2 = time.perf_counter()
lines, mean, _ = fe.scan_find_peaks(cap=4096)
t3 = time.perf_counter()
samples = float(N) * F
print("N=%d frames=%d avg=%d: wall %.2f ms (%.1f Gsamples/s) | fft %.3f ms (%d launches, %.0f GB/s alg @12B) | "
      "movsum %.3f ms | peaks %d in %.2f ms (D2H + host)" % (
          N, F, L, (t1 - t0) * 1e3, samples / (t1 - t0) / 1e9, fft_ms, n1, 12.0 * samples / (fft_ms * 1e-3) / 1e9,
          ms_ms, len(lines), (t3 - t2) * 1e3))
if os.environ.get("SAVE"):
    np.save(os.environ["SAVE"], np.asarray(spec))
