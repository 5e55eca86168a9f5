#!/usr/bin/env python3
"""One front-end of the real-time leg's shape (256-bin bank + 32 FM channels, 20 ms u8 blocks) for a kernel trace:
   rocprofv3 --kernel-trace --stats -- python tools/rt_trace_probe.py      (what does a real-time block cost on the GPU?)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radiocapture-rf_amd")]
import numpy as np
import bench
from rcf import native, synth
tile, meta = synth.cfg2(n=1 << 20, seed=2002, n_bins=256, n_active=32)
blk = 400000
K = int(os.environ.get("K", 1))
raw = native.PinnedArray(2 * len(tile), np.uint8)
raw.array[:] = np.clip(np.round(tile.view(np.float32) * 32 + 127.4), 0, 255).astype(np.uint8)
fes, chans = [], []
for i in range(K):
    fe = native.Frontend(20e6, 0.0, block_capacity=blk, hist_capacity=1 << 14, out_capacity=1 << 12)
    fe.pfb_open(256, 256, bench.proto_taps(native))
    chans.append([fe.pfb_chan_open(c["bin"] % 256, 12500, c["delta"]) for c in meta["carriers"]])
    fes.append(fe)
out = np.empty((32, 1024), dtype=np.float32)
t0 = time.perf_counter()
for k in range(int(os.environ.get("BLOCKS", 100))):
    for i, fe in enumerate(fes):
        fe.push_raw(raw.array[: 2 * blk], native.FMT_U8, 1.0 / 32, 127.4)
    for i, fe in enumerate(fes):
        fe.chan_read_many(chans[i], "fm", cap_each=1024, out=out)
print("wall per front-end block: %.1f us" % ((time.perf_counter() - t0) / int(os.environ.get("BLOCKS", 100)) / K * 1e6))
