"""Untimed GPU leg: what one grouped launch per stage is worth (rcf_group_*), blocks resident in HBM."""
import time

import numpy as np

from .common import FS, NB, HBM_PEAK_GBS, proto_taps

def group_capacity_leg(native, tile, carriers, device, G=80, blk=409600, seconds=1.5):
    """What ONE grouped launch per stage is worth when the GPU is kept busy: G front-ends (each the BASELINE configs[1]
    shape: 256-bin bank + 32 FM channels) with a real-time-sized block resident in HBM, committed back to back as one
    group block (rcf_group_commit: one records + history launch, ONE filterbank launch over all G x 100 chunks, ONE stage-2
    launch over all G x 32 channels), against the same G front-ends committed one after the other.  The paced real-time
    leg runs the same launches at a duty cycle of a few per cent (the chip idles between group blocks and clocks down:
    its launches are slower than these)."""
    fes, ids = [], []
    for i in range(G):
        fe = native.Frontend(FS, 0.0, device=device, block_capacity=blk, hist_capacity=1 << 14, out_capacity=1 << 12)
        fe.pfb_open(NB, NB, proto_taps(native))
        ids.append([fe.pfb_chan_open(c["bin"] % NB, 12500, c["delta"]) for c in carriers])
        fes.append(fe)
    x = np.tile(tile, blk // len(tile) + 1)[:blk]
    out = {"front_ends": G, "block_samples": blk, "samples_per_group_block": G * blk,
           "algorithmic_bytes_per_filterbank_launch": 16.0 * G * blk}
    for mode in ("one_by_one", "grouped"):
        grp = native.Group(fes) if mode == "grouped" else None
        for _ in range(2):                               # both ping-pong buffers of every member hold data
            if grp is not None:
                grp.push([np.roll(x, 977 * i) for i in range(G)], native.FMT_CF32)
            else:
                for i, fe in enumerate(fes):
                    fe.push(np.roll(x, 977 * i))
        step = (lambda: grp.commit([blk] * G)) if grp is not None else (lambda: [fe.commit(blk) for fe in fes])
        sync = grp.sync if grp is not None else (lambda: [fe.sync() for fe in fes])
        for _ in range(30):
            step()
        sync()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < seconds / 2:
            step()
            n += 1
        sync()
        wall = (time.perf_counter() - t0) / n
        e = {"wall_ms_per_group_block": wall * 1e3, "input_Msps": G * blk / wall / 1e6, "group_blocks": n}
        # the filterbank launch itself, HIP events on the launch stream (the grouped launches are timed on member 0)
        fes[0].timing_enable(True, classes=[native.T_PFB, native.T_FIR_DERIVED])
        fes[0].timing_read(native.T_PFB)
        fes[0].timing_read(native.T_FIR_DERIVED)
        for _ in range(100):
            step()
        sync()
        ms, k = fes[0].timing_read(native.T_PFB)
        ms2, k2 = fes[0].timing_read(native.T_FIR_DERIVED)
        fes[0].timing_enable(False)
        if k:
            per = (16.0 * G * blk) if grp is not None else 16.0 * blk
            e["filterbank_launch_us"] = ms / k * 1e3
            e["filterbank_launches_timed"] = k
            e["filterbank_frac_of_hbm_peak"] = per / (ms / k * 1e-3) / 1e9 / HBM_PEAK_GBS
        if k2:
            e["stage2_launch_us"] = ms2 / k2 * 1e3
        out[mode] = e
        if grp is not None:
            # the grouped outputs are the one-by-one outputs (bit for bit: tests/test_gpu_group.py); here only that every
            # channel produced the same count either way
            e["outputs_per_channel"] = fes[G // 2].chan_produced(ids[G // 2][0])
            grp.close()
    out["grouped_over_one_by_one"] = out["one_by_one"]["wall_ms_per_group_block"] / out["grouped"]["wall_ms_per_group_block"]
    for fe in fes:
        fe.close()
    return out
