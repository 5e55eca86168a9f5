#!/usr/bin/env python3
"""Print the worst-case movement of each GNU Radio detail the oracle cannot pin (DESIGN.md 2 quotes this table):
   python tools/unpinned_bounds.py > profiles/r04_unpinned_bounds.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radiocapture-rf_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from oracle import cbind as OC, grspec as G, unpinned as U
from rcf import synth
import test_oracle_unpinned_bounds as T

out = {"log2": {}, "summation": {}}
for N, fs, n_car, seed in ((16384, 2.4e6, 5, 3004), (1 << 20, 100e6, 12, 3003)):
    x, _ = T._scan_stream(fs, N, seed, n_car)
    ref, got, shift = U.peaks_under_log2_error(x, N, fs, 855e6)
    out["log2"]["N=%d" % N] = {"per_value_error_log2_units": U.LOG2_ABS_ERR, "patterns": len(got),
                               "index_lists_changed": sum(v != ref for v in got.values()), "peaks": len(ref),
                               "largest_move_of_the_summed_spectrum": shift}
x, _ = T._scan_stream(2.4e6, 16384, 3004, 5)
out["log2"]["smallest_error_that_moves_an_index_N=16384"] = U.log2_error_that_moves_a_peak(x, 16384, 2.4e6, 855e6, 1e-3, 0.5)
x1, meta = synth.cfg1(seconds=0.2)
D, taps = G.channel_params(meta["fs"], 12500)
y = G.xlating_fir_ccc(x1, D, taps, meta["offset"], meta["fs"])
out["atan_table"] = {"entries_moved_by": "+-1 unit of the 7th significant digit, six sign patterns",
                     "max_fm_change_p25_gain": U.fm_under_table_perturbation(y, G.p25_fm_gain(25000.0))}
for fs, f0 in ((2.4e6, -62500.0), (20e6, 5.0125e6)):
    D, taps = G.channel_params(fs, 12500)
    ct, inc = OC.xlating_composite(taps, D, f0, fs)
    xx = synth.awgn(np.random.default_rng(11), D * 600)
    vs64, between, _ = U.iq_under_summation_orders(xx, D, ct)
    out["summation"]["fs=%g T=%d" % (fs, len(taps))] = {"rel_rms_vs_float64": vs64, "largest_between_two_float32_orders": between}
rot = {}
for f0 in T.ROTATOR_OFFSETS:
    _, inc = OC.xlating_composite(G.channel_params(20e6, 12500)[1], 800, f0, 20e6)
    rot["f0=%g" % f0] = U.rotator_fma_drift(inc, 1_000_000)
out["rotator_fma"] = {"by_offset": rot,
                      "max_step_difference_rad": max(r["max_step_difference_rad"] for r in rot.values()),
                      "max_phase_difference": max(r["max_phase_difference"] for r in rot.values())}
print(json.dumps(out, indent=1))
