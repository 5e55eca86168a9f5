#!/usr/bin/env python3
"""Generate tests/golden/peaks.npz by RUNNING the reference's own peak detection.

/root/reference/fft_peak_detection.py is a script: argument parsing, file and Redis access at module level.  This
generator (build container only) lifts the statements that ARE the detection -- from `fft_width = ...` through the
`for line in peaks[0]` loop, fft_peak_detection.py:44-73 -- out with `ast` and executes them as they stand on synthetic
spectra, with the demodulator constructor replaced by a recorder.  Written: the spectra (quantised to 1/8 so that they
compress, and so that plateaus and ties occur) and, for each, the frequencies the reference printed -- data only.
"""
import ast
import os
import sys
import types

import numpy
import numpy as np
from scipy import signal

REF = "/root/reference/fft_peak_detection.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "peaks.npz")

tree = ast.parse(open(REF).read())
body, take = [], False
for node in tree.body:
    if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", None) == "fft_width":
        take = True
    if take:
        body.append(node)
    if take and isinstance(node, ast.For) and isinstance(node.iter, ast.Subscript):
        break
assert take and isinstance(body[-1], ast.For)
code = compile(ast.Module(body=body, type_ignores=[]), REF, "exec")


def run_reference(data, fs, fc):
    found = []

    class demod:
        def __init__(self, cfg, *a, **k):
            found.append(cfg["channels"][0])

        def start(self):
            pass

    g = {"numpy": numpy, "signal": signal, "data": data.copy(),
         "config": types.SimpleNamespace(sources={0: {"samp_rate": fs, "center_freq": fc}}),
         "args": types.SimpleNamespace(index=0), "p25_control_demod": demod, "site_uuid": "s", "overseer_uuid": "o",
         "rcm": None, "print": lambda *a: None}
    exec(code, g)
    return found


out = {}
meta = []
N = 16384                                          # the reference's fft_width (fft_peak_detection.py:44)
for case in range(16):
    rng = np.random.default_rng(52000 + case)
    fs = int(rng.choice([2400000, 8000000, 10000000, 20000000]))
    fc = int(rng.choice([855050000, 770000000, 460000000]))
    hz = fs / N
    x = rng.normal(100.0, float(rng.uniform(0.5, 6.0)), N)
    for _ in range(int(rng.integers(1, 12))):
        c = int(rng.integers(0, N))
        w = float(rng.uniform(500.0, 60000.0)) / hz
        d = (np.arange(N) - c) / max(w / 2.355, 0.5)
        x += float(rng.uniform(5, 60)) * (np.exp(-0.5 * d ** 2) if rng.random() < 0.7 else (np.abs(np.arange(N) - c) < w / 2))
    x = (np.round(x * 8) / 8 - float(rng.choice([0.0, 480.0]))).astype(np.float32)
    freqs = run_reference(x, fs, fc)
    out["spectrum_%02d" % case] = x
    out["freqs_%02d" % case] = np.array(freqs, dtype=np.int64)
    meta.append((fs, fc, len(freqs)))
out["meta"] = np.array(meta, dtype=np.int64)
np.savez_compressed(OUT, **out)
print("wrote", OUT, os.path.getsize(OUT), "bytes;", [m[2] for m in meta], "peaks per spectrum")
