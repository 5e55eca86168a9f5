#!/usr/bin/env python3
"""Kernel trace of the real-time leg (rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/rt_probe.py):
steady-state figures per kernel -- dispatches after the first SKIP_S seconds of pump activity only -- and, for the grouped
filterbank launches, the bytes each moved (grid workgroups x one chunk's algorithmic bytes) over its duration.
  python tools/rt_trace_outliers.py DIR [SKIP_S] -> JSON on stdout"""
import collections, csv, glob, json, sys
d = sys.argv[1]
skip_s = float(sys.argv[2]) if len(sys.argv) > 2 else 1.5
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
grp = [r for r in rows if "group_prep_kernel" in r["Kernel_Name"]]
t_first = min(int(r["Start_Timestamp"]) for r in grp)
t_lo = t_first + int(skip_s * 1e9)
# algorithmic bytes of one workgroup's chunk: 16 frames x 256 bins x (8 read + 8 written) for the 256-bin bank; the 1600-bin
# bank: 4 frames x (800 x 8 read + 1600 x 8 written)
chunk_bytes = {"pfb_group_kernel_os<256": 16 * 256 * 16, "pfb5_group_kernel<20, 4, 2, 2>": 4 * (800 * 8 + 1600 * 8),
               "pfb5_fmlb_group_kernel<20, 4, 2, 2, 2>": 4 * (800 * 8 + 1600 * 4)}      # (fused discriminator, fm ring only)
out = {"trace": f, "steady_state_from_s_after_first_group_block": skip_s, "kernels": {}}
acc = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for r in rows:
    if int(r["Start_Timestamp"]) < t_lo:
        continue
    n = r["Kernel_Name"]
    dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    key = n.split("(anonymous namespace)::")[-1].split("(rcfx")[0].replace("void ", "")[:60]
    a = acc[key]
    a[0] += 1
    a[1] += dur
    a[2] = max(a[2], dur)
    for k, cb in chunk_bytes.items():
        if k in n:
            wgs = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
            a[3] += wgs * cb
for k, a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    e = {"launches": a[0], "avg_us": a[1] / a[0] / 1e3, "max_us": a[2] / 1e3, "total_ms": a[1] / 1e6}
    if a[3]:
        e["algorithmic_GB_per_launch_mean"] = a[3] / a[0] / 1e9
        e["achieved_GBps"] = a[3] / a[1]
        e["frac_of_8TBps"] = a[3] / a[1] / 8000.0
    out["kernels"][k] = e
print(json.dumps(out, indent=1))
