#!/usr/bin/env python3
"""What HIP-event timing costs the bench step: commit loop with timing off / around the PFB launch only / around every
launch (measured on MI355X: 0.155 / 0.153 / 0.165 ms per step -- the PFB-only setting bench.py uses is free)."""
import os, sys, time
ROOT='/root/repo' if os.path.isdir('/root/repo/radiocapture-rf_amd') else os.getcwd()
sys.path[:0]=[ROOT, os.path.join(ROOT,'radiocapture-rf_amd')]
import numpy as np
from rcf import native, synth
FS, NB, B = 20e6, 256, 1<<25
bw=FS/NB
taps=native.design_low_pass_2(1.0,FS,0.4*bw,0.2*bw,60.0,native.WIN_BLACKMAN_HARRIS)
fe=native.Frontend(FS,0.0,block_capacity=B,hist_capacity=1<<16,out_capacity=1<<18)
fe.pfb_open(NB,NB,taps)
tile,meta=synth.cfg2(n=1<<20,seed=2002,n_bins=NB,n_active=32)
ch=[fe.pfb_chan_open(c["bin"]%NB,12500,c["delta"]) for c in meta["carriers"]]
for _ in range(2):
    for at in range(0,B,len(tile)): fe.ingest_write(tile,at)
    fe.commit(B)
for mode in ("off","pfb","all"):
    if mode=="off": fe.timing_enable(False)
    elif mode=="pfb": fe.timing_enable(True, classes=[native.T_PFB])
    else: fe.timing_enable(True)
    for _ in range(3): fe.commit(B)
    fe.sync(); t0=time.perf_counter()
    for _ in range(40): fe.commit(B)
    fe.sync(); dt=(time.perf_counter()-t0)/40
    print("timing %s: %.4f ms per step"%(mode,dt*1e3))
