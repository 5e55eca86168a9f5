#!/usr/bin/env python3
"""The paced real-time leg of bench.py alone:  SECONDS=3 KFIRST=512 KCAP=1280 PUMPS=8 SHAPES=pfb256,grid1600 python tools/rt_probe.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radiocapture-rf_amd")]
import bench
numa = None if int(os.environ.get("NOPIN", 0)) else bench.pin_to_gpu_numa(0)      # as bench.py does, before the HIP runtime starts
from rcf import native, synth
tile, meta = synth.cfg2(n=1 << 20, seed=2002, n_bins=256, n_active=32)
r = bench.realtime_leg(native, tile, meta["carriers"], 0, seconds=float(os.environ.get("SECONDS", 3)),
                       block_ms=float(os.environ.get("BLOCK_MS", 20)), k_first=int(os.environ.get("KFIRST", 512)),
                       k_cap=int(os.environ.get("KCAP", 1280)), n_pumps=int(os.environ.get("PUMPS", 0)), window_ms=float(os.environ.get("WINDOW_MS", 1.0)), shapes=tuple(os.environ.get("SHAPES", "pfb256,grid1600").split(",")),
                       stagger=not int(os.environ.get("BURST", 0)))
r["numa"] = numa
print(json.dumps(r, indent=1))
