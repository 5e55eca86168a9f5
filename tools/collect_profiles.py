#!/usr/bin/env python3
"""Copy what tools/make_profiles.sh left under gpurun_out/ into profiles/ (run where gpurun_out/ was merged):
   python tools/collect_profiles.py r01"""
import shutil, os, re
import collections, csv, glob, json, os, sys
R = sys.argv[1]
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for f in sorted(glob.glob("gpurun_out/%s_pmc_%s/**/*counter_collection.csv" % (R, c), recursive=True),
                    key=os.path.getmtime)[-1:]:                            # newest run only
        for r in csv.DictReader(open(f)):
            if "pfb_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                v.append(float(r["Counter_Value"]))
                name = r["Kernel_Name"]
    v = v[1:] if len(v) > 2 else v      # drop the first (zero-history instantiation / cold) launch
    vals[c] = (sum(v) / len(v), len(v)) if v else (None, 0)
if vals["FETCH_SIZE"][0] is not None and vals["WRITE_SIZE"][0] is not None:
    fetch = vals["FETCH_SIZE"][0] * 1024 * 2       # KiB -> B, gfx950 x2 correction (MI355X_MICROARCH.md)
    write = vals["WRITE_SIZE"][0] * 1024
    B = 1 << 25
    json.dump({"block": B, "kernel": (re.search(r"pfb_kernel\w*<[^>]*>", name) or re.search(r".*", name)).group(0), "launches_averaged": vals["FETCH_SIZE"][1],
               "FETCH_SIZE_KiB_raw": vals["FETCH_SIZE"][0], "WRITE_SIZE_KiB_raw": vals["WRITE_SIZE"][0],
               "fetch_bytes_corrected_x2": fetch, "write_bytes": write, "hbm_bytes_per_launch": fetch + write,
               "algorithmic_bytes_per_launch": 16.0 * B,
               "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of `rocprofv3 --pmc X --kernel-trace -- python "
                       "bench.py --steps 5 --warmup 1 --no-cpu-baseline`; KiB units and the gfx950 FETCH_SIZE x2 "
                       "correction per MI355X_MICROARCH.md; counters sit at the L2<->fabric boundary, so Infinity-Cache "
                       "hits are included"}, open("profiles/pfb_traffic.json", "w"), indent=1)
    print("traffic: fetch %.1f MB write %.1f MB (algorithmic %.1f MB)" % (fetch / 1e6, write / 1e6, 16.0 * B / 1e6))
else:
    print("traffic: counters missing", vals)

for src, dst in (("gpurun_out/%s_bench.json" % R, "profiles/%s_bench.json" % R),):
    if os.path.exists(src):
        with open(src) as f:
            lines = [l for l in f.read().splitlines() if l.startswith("{")]
        if lines:
            open(dst, "w").write(lines[-1] + "\n")
fs = sorted(glob.glob("gpurun_out/%s_trace/**/*kernel_stats.csv" % R, recursive=True), key=os.path.getmtime)
if fs:
    shutil.copy(fs[-1], "profiles/%s_bench_kernel_stats.csv" % R)       # newest run
