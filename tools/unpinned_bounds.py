#!/usr/bin/env python3
"""Print the worst-case movement of each GNU Radio detail the oracle cannot pin (DESIGN.md 2 quotes this table):
   python tools/unpinned_bounds.py > profiles/r05_unpinned_bounds.json"""
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
# ---- round 5: the analog voice chain (f-2) and the routing budget of frontend_mode = 'pfb'
import math
from oracle import audio as A
from rcf import native
st = T._voice_stream()
forms, scale = U.audio_under_deemph_forms(st["fm"], 25000.0)
dens, tap_move, _ = U.audio_under_remez_density(st["deemph"], 25000.0)
rs, _ = U.audio_under_resampler_tap_rounding(st["hpf"], 25000.0)
out["voice_chain"] = {"audio_rms": scale, "fm_deemph_evaluation_order_audio_rms_change": forms,
                      "pm_remez_grid_density_audio_rms_change_vs_16": {str(k): v for k, v in dens.items()},
                      "pm_remez_largest_tap_change_vs_16": {str(k): v for k, v in tap_move.items()},
                      "resampler_taps_pm_1ulp_audio_rms_change": rs}
fs, NB = 20e6, 1600
D, taps = G.channel_params(fs, 12500)
n = D * 260
rng = np.random.default_rng(5)
gain = G.p25_fm_gain(25000.0)
amp = synth.snr_amp(30.0, 12500.0, fs)
rows = []
for k in (3, 81, 161, 241, 321, 401, 481, 561, 641, 721, 797, NB - 700, NB - 400, NB - 100):
    f0 = (k if k < NB // 2 else k - NB) * fs / NB
    x = synth.awgn(rng, n).astype(np.complex128)
    for j in range(32):
        f = f0 if j == 0 else float(rng.integers(-780, 780)) * 12500.0
        x += synth.nbfm_carrier(n, fs, f, 1000.0 + 37 * j, 2500.0, amp, phase0=float(rng.uniform(0, 6.28)))
    x = x.astype(np.complex64)
    leak, _ = native.pfb_tap_leakage(fs, NB, taps, k)
    pred = gain * leak * math.sqrt(float(np.mean(np.abs(x) ** 2)) / amp ** 2)
    rows.append({"bin": k, "leak_l2": leak, "predicted_fm_error_without_margin": pred,
                 "measured_float_fwT0": U.bin_fm_error_vs_gr(x, fs, NB, taps, D, k, gain, "float_product"),
                 "measured_double_fwT0": U.bin_fm_error_vs_gr(x, fs, NB, taps, D, k, gain, "double_fwT0")})
out["pfb_routing"] = {"rows": rows, "margin_used_by_the_receiver": 2.5,
                      "largest_measured_over_predicted": max(max(r["measured_float_fwT0"], r["measured_double_fwT0"]) / r["predicted_fm_error_without_margin"] for r in rows)}
print(json.dumps(out, indent=1))
