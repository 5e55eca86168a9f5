#!/usr/bin/env python3
"""Generate tests/golden/reference_configs.json: what the reference's site configurations (configs/*.py) ask of the
channelizer -- per file: receiver_split2, frontend_mode, scan_mode and every source's type / centre / rate.  Build
container only (needs /root/reference).  Most of those files mix tabs and spaces the Python-2 way and no longer import
under Python 3; they are read with rcf.frontend.load_config, which falls back to Python 2's tab rule.  Numbers only."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "radiocapture-rf_amd")]
from rcf import frontend  # noqa: E402

out = {}
for f in sorted(glob.glob("/root/reference/configs/*.py")):
    c = frontend.load_config(f)
    try:
        compile(open(f).read(), f, "exec")
        py3 = True
    except TabError:
        py3 = False
    out[os.path.basename(f)] = {
        "imports_under_python3": py3,
        "receiver_split2": bool(getattr(c, "receiver_split2", False)),
        "frontend_mode": getattr(c, "frontend_mode", None),
        "scan_mode": bool(getattr(c, "scan_mode", False)),
        "sources": [{"index": k, "type": s.get("type"), "center_freq": s.get("center_freq"), "samp_rate": s.get("samp_rate")}
                    for k, s in sorted(c.sources.items())],
    }
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_configs.json"), "w") as fh:
    json.dump(out, fh, indent=1, sort_keys=True)
print(len(out), "configs")
