#!/usr/bin/env python3
"""The group_capacity leg of bench.py alone (G front-ends committed back to back as one group block):
   G=80 SECONDS=3 python tools/group_probe.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radiocapture-rf_amd")]
import bench
from rcf import native, synth
tile, meta = synth.cfg2(n=1 << 20, seed=2002, n_bins=256, n_active=32)
r = bench.group_capacity_leg(native, tile, meta["carriers"], 0, G=int(os.environ.get("G", 80)),
                             blk=int(os.environ.get("BLK", 409600)), seconds=float(os.environ.get("SECONDS", 3)))
print(json.dumps(r, indent=1))
