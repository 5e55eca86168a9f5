#!/usr/bin/env python3
"""Copy what tools/make_profiles.sh left under gpurun_out/ into profiles/ (run where gpurun_out/ was merged):
   python tools/collect_profiles.py r02"""
import csv, glob, json, os, re, shutil, sys
R = sys.argv[1]


def newest(pattern):
    fs = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return fs[-1] if fs else None


def counters(d, kernel, skip_first=True):
    """average per dispatch of every counter for kernels whose name contains `kernel`; + the dispatch durations"""
    acc, dur, name = {}, [], None
    f = newest(os.path.join(d, "**", "*counter_collection.csv"))
    if f:
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                name = r["Kernel_Name"]
    f = newest(os.path.join(d, "**", "*kernel_trace.csv"))
    if f:
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {}
    for k, v in acc.items():
        v = v[1:] if skip_first and len(v) > 2 else v
        out[k] = sum(v) / len(v)
    if dur:
        dur = dur[1:] if skip_first and len(dur) > 2 else dur
        out["kernel_us"] = sum(dur) / len(dur)
        out["launches_averaged"] = len(dur)
    return out, name


def short(name):
    m = re.search(r"(pfb5?_kernel\w*<[^>]*>|fir_mfma_kernel<[^>]*>|[A-Za-z_0-9]+_kernel[A-Za-z_0-9<>, ]*)", name or "")
    return m.group(0) if m else name


# ---- timed configurations: PFB traffic (cfg4 -> pfb_traffic.json, cfg5 -> pfb512_traffic.json)
when = open("gpurun_out/%s_when.txt" % R).read().strip() if os.path.exists("gpurun_out/%s_when.txt" % R) else "?"
for tag, fname, flag in (("pmc", "pfb_traffic.json", ""), ("pmc5", "pfb512_traffic.json", " --config cfg5")):
    v = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        v[c], name = counters("gpurun_out/%s_%s_%s" % (R, tag, c), "pfb_kernel")
    if v["FETCH_SIZE"].get("FETCH_SIZE") is not None and v["WRITE_SIZE"].get("WRITE_SIZE") is not None:
        fetch = v["FETCH_SIZE"]["FETCH_SIZE"] * 1024 * 2       # KiB -> B, gfx950 x2 correction (MI355X_MICROARCH.md)
        write = v["WRITE_SIZE"]["WRITE_SIZE"] * 1024
        B = 1 << 25
        alg = 16.0 * B
        try:        # cfg4's launch also carries the previous block's stage-2 workgroups (round 5): the bench line states the bytes
            bl = [l for l in open("gpurun_out/%s_bench%s.json" % (R, "_cfg5" if tag == "pmc5" else "")).read().splitlines() if l.startswith("{")]
            alg = float(json.loads(bl[-1])["roofline"].get("algorithmic_bytes_per_launch", alg))
        except Exception:
            pass
        json.dump({"block": B, "kernel": short(name), "measured": "%s make_profiles run of %s" % (R, when),
                   "launches_averaged": v["FETCH_SIZE"].get("launches_averaged"),
                   "FETCH_SIZE_KiB_raw": v["FETCH_SIZE"]["FETCH_SIZE"], "WRITE_SIZE_KiB_raw": v["WRITE_SIZE"]["WRITE_SIZE"],
                   "fetch_bytes_corrected_x2": fetch, "write_bytes": write, "hbm_bytes_per_launch": fetch + write,
                   "algorithmic_bytes_per_launch": alg,
                   "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of `rocprofv3 --pmc X --kernel-trace -- python "
                           "bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-sustained%s`; KiB units and the "
                           "gfx950 FETCH_SIZE x2 correction per MI355X_MICROARCH.md; counters sit at the L2<->fabric "
                           "boundary, so Infinity-Cache hits are included" % flag}, open("profiles/" + fname, "w"), indent=1)
        print("%s: fetch %.1f MB write %.1f MB (algorithmic %.1f MB)" % (fname, fetch / 1e6, write / 1e6, alg / 1e6))
    else:
        print("%s: counters missing" % fname)


def pmc_record(tag, kernel, out, extra):
    rec = {}
    for i in (1, 2, 3, 4):
        c, name = counters("gpurun_out/%s_pmc_%d" % (tag, i), kernel)
        if not c:
            continue
        us = c.pop("kernel_us", None)
        c.pop("launches_averaged", None)
        rec["pass%d" % i] = dict(c, kernel_us_in_this_pass=us)
        rec["kernel"] = short(name)
    if rec:
        rec.update(extra)
        alg = extra.get("algorithmic_read_bytes")
        f, w = rec.get("pass3", {}).get("FETCH_SIZE"), rec.get("pass4", {}).get("WRITE_SIZE")
        if alg and f is not None and w is not None:
            rec["fetch_x2_bytes"] = f * 1024 * 2
            rec["write_bytes"] = w * 1024
            rec["fetch_x2_over_algorithmic_read"] = f * 1024 * 2 / alg
            rec["hbm_bytes_over_algorithmic"] = (f * 1024 * 2 + w * 1024) / extra["algorithmic_bytes"]
        json.dump(rec, open(out, "w"), indent=1)
        print("wrote", out)


pmc_record("%s_fir4096" % R, "fir_mfma_kernel<", "profiles/%s_fir_mfma_pmc.json" % R, {
    "workload": "tools/fir_probe.py C=4096: 4096 reference-shaped channels (D=800, T=2909), 20 Msps, block 2^22",
    "ideal_mfma_instructions": 4096 * 5243 * 2909 * 8 / 2048.0,
    "how_to_read": "MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); clock = "
                   "GRBM_GUI_ACTIVE / 8 / kernel_us; SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles; "
                   "FETCH_SIZE in KiB and x2 on gfx950 (MI355X_MICROARCH.md)"})
pmc_record("%s_pfb512" % R, "pfb_kernel", "profiles/%s_pfb512_traffic.json" % R, {
    "workload": "tools/pfb_probe.py NB=512: 512-bin critically sampled bank (two-branch form pfb_kernel_2b<256, 14, 3>), block 2^25; algorithmic 16 B/sample = 536.9 MB",
    "algorithmic_read_bytes": 8.0 * (1 << 25), "algorithmic_bytes": 16.0 * (1 << 25),
    "how_to_read": "HBM bytes = FETCH_SIZE KiB x 1024 x 2 + WRITE_SIZE KiB x 1024"})
pmc_record("%s_pfb1024" % R, "pfb_kernel", "profiles/%s_pfb1024_traffic.json" % R, {
    "workload": "tools/pfb_probe.py NB=1024: 1024-bin critically sampled bank (two-branch form pfb_kernel_2b<512, 14, 4>), block 2^25; algorithmic 16 B/sample = 536.9 MB",
    "algorithmic_read_bytes": 8.0 * (1 << 25), "algorithmic_bytes": 16.0 * (1 << 25),
    "how_to_read": "HBM bytes = FETCH_SIZE KiB x 1024 x 2 + WRITE_SIZE KiB x 1024"})
pmc_record("%s_pfb1600" % R, "pfb5_kernel", "profiles/%s_pfb1600_pmc.json" % R, {
    "workload": "tools/pfb_probe.py NB=1600 BLOCK=2^25: 1600-bin bank, D = 800, 2909 taps; algorithmic 24 B/sample = 805.3 MB",
    "algorithmic_read_bytes": 8.0 * (1 << 25), "algorithmic_bytes": 24.0 * (1 << 25),
    "how_to_read": "HBM bytes = FETCH_SIZE KiB x 1024 x 2 + WRITE_SIZE KiB x 1024"})
pmc_record("%s_pfb3200a" % R, "pfb5_kernel", "profiles/%s_pfb3200_d1600_pmc.json" % R, {
    "workload": "tools/pfb_probe.py NB=3200 CR=6250 BLOCK=2^25: 3200-bin bank, D = 1600, 5819 taps (3 launches after idling: the duration here is not the steady-state one); algorithmic 24 B/sample = 805.3 MB",
    "algorithmic_read_bytes": 8.0 * (1 << 25), "algorithmic_bytes": 24.0 * (1 << 25),
    "how_to_read": "HBM bytes = FETCH_SIZE KiB x 1024 x 2 + WRITE_SIZE KiB x 1024"})
pmc_record("%s_pfb3200b" % R, "pfb5_kernel", "profiles/%s_pfb3200_d800_pmc.json" % R, {
    "workload": "tools/pfb_probe.py NB=3200 CR=12500 BLOCK=2^25: 3200-bin bank, D = 800, 2909 taps (3 launches after idling); algorithmic 40 B/sample = 1342.2 MB",
    "algorithmic_read_bytes": 8.0 * (1 << 25), "algorithmic_bytes": 40.0 * (1 << 25),
    "how_to_read": "HBM bytes = FETCH_SIZE KiB x 1024 x 2 + WRITE_SIZE KiB x 1024"})
pmc_record("%s_pfb1600fm" % R, "pfb5_fmlb_kernel", "profiles/%s_pfb1600_fused_pmc.json" % R, {
    "workload": "tools/pfb_probe.py NB=1600 FMFUSED=2 BLOCK=2^25: the 1600-bin bank with the discriminator of every bin fused in, discriminator ring only (pfb5_fmlb_kernel); algorithmic 8 + 8 B/sample = 536.9 MB",
    "algorithmic_read_bytes": 8.0 * (1 << 25), "algorithmic_bytes": 16.0 * (1 << 25),
    "how_to_read": "HBM bytes = FETCH_SIZE KiB x 1024 x 2 + WRITE_SIZE KiB x 1024; the hand-over rows (one frame per chunk, written through L2 and read back by the next chunk's workgroup) are part of the traffic: 4 + 4 B per input sample at most"})
pmc_record("%s_tapfin" % R, "tap_finalize", "profiles/%s_tap_finalize_pmc.json" % R, {
    "workload": "tools/pfb_probe.py NB=1600 TAPS=1600 BLOCK=2^25: tap_finalize_kernel with every bin of the 1600-bin bank open as a channel (41943 frames x 1600 taps per launch; 3 launches after idling)",
    "algorithmic_read_bytes": 8.0 * 1600 * 41943, "algorithmic_bytes": 20.0 * 1600 * 41943,
    "how_to_read": "HBM bytes = FETCH_SIZE KiB x 1024 x 2 + WRITE_SIZE KiB x 1024; since round 5 a tile reads exactly the rows its outputs reach (r_need_lo..r_need_hi of tap_finalize_tile), not the 32-aligned 161 of round 4"})

# ---- shader clock of the filterbank launches over the sustained leg (first / last 100 dispatches)
def clock_series(d, kernel):
    cnt, dur = {}, {}
    f = newest(os.path.join(d, "**", "*counter_collection.csv"))
    if f:
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                cnt[int(r["Dispatch_Id"])] = cnt.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    f = newest(os.path.join(d, "**", "*kernel_trace.csv"))
    if f:
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    ids = sorted(i for i in cnt if i in dur and dur[i] > 0)
    return [(cnt[i] / 8.0 / dur[i] / 1e3, dur[i]) for i in ids]      # (GHz, us)


ser = clock_series("gpurun_out/%s_pmc_clock" % R, "pfb_kernel_os<256")
if len(ser) > 400:
    w = lambda xs: sum(xs) / len(xs)
    json.dump({"kernel": "pfb_kernel_os<256, 1, 14, 4, false>", "dispatches": len(ser),
               "clock_ghz_first_100": w([c for c, _ in ser[:100]]), "clock_ghz_last_100": w([c for c, _ in ser[-100:]]),
               "kernel_us_first_100_under_pmc": w([u for _, u in ser[:100]]),
               "kernel_us_last_100_under_pmc": w([u for _, u in ser[-100:]]),
               "how": "rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace over `bench.py --steps 20 --warmup 5 --no-extras "
                      "--no-cpu-baseline` (prewarm + timed region + the 2 s sustained leg); clock = GRBM_GUI_ACTIVE / 8 XCDs / "
                      "dispatch duration; durations under counter collection are longer than un-profiled ones"},
              open("profiles/%s_sustained_clock.json" % R, "w"), indent=1)
    print("clock: first 100 %.3f GHz, last 100 %.3f GHz over %d dispatches" % (
        w([c for c, _ in ser[:100]]), w([c for c, _ in ser[-100:]]), len(ser)))

for shape in ("pfb256", "grid1600"):
    for tag in ("rt_%s_steady_state" % shape, "rt_%s_under_rocprof" % shape):
        src = "gpurun_out/%s_%s.json" % (R, tag)
        if os.path.exists(src) and open(src).read().strip():
            shutil.copy(src, "profiles/%s_%s.json" % (R, tag))
    f = newest("gpurun_out/%s_trace_rt_%s/**/*kernel_stats.csv" % (R, shape))
    if f:
        shutil.copy(f, "profiles/%s_rt_%s_kernel_stats.csv" % (R, shape))
src = "gpurun_out/%s_group_capacity_under_rocprof.json" % R
if os.path.exists(src) and open(src).read().strip():
    shutil.copy(src, "profiles/%s_group_capacity_under_rocprof.json" % R)
f = newest("gpurun_out/%s_trace_group/**/*kernel_stats.csv" % R)
if f:
    shutil.copy(f, "profiles/%s_group_capacity_kernel_stats.csv" % R)
for tag in ("pfb1600_limiter_pmc", "pfb1600_fused_limiter_pmc", "fused_discriminator_timings"):
    src = "gpurun_out/%s_%s.txt" % (R, tag)
    if os.path.exists(src) and open(src).read().strip():
        shutil.copy(src, "profiles/%s_%s.txt" % (R, tag))
for tag in ("bench", "bench_line", "bench_cfg5", "bench_head_under_rocprof", "bench_2ranks_1gpu", "sustained_60s", "unpinned_bounds", "hbm_mix_probe"):
    src = "gpurun_out/%s_%s.json" % (R, tag)
    if os.path.exists(src):
        text = open(src).read()
        lines = [l for l in text.splitlines() if l.startswith("{") and l.rstrip().endswith("}")]
        if tag in ("unpinned_bounds", "hbm_mix_probe") and text.strip():
            open("profiles/%s_%s.json" % (R, tag), "w").write(text)
        elif lines:
            open("profiles/%s_%s.json" % (R, tag), "w").write(lines[-1] + "\n")
f = newest("gpurun_out/%s_trace/**/*kernel_stats.csv" % R)
if f:
    shutil.copy(f, "profiles/%s_bench_kernel_stats.csv" % R)
fn = newest("gpurun_out/%s_trace_nolag/**/*kernel_stats.csv" % R)
if fn:
    shutil.copy(fn, "profiles/%s_bench_nolag_kernel_stats.csv" % R)
f5 = newest("gpurun_out/%s_trace_cfg5/**/*kernel_stats.csv" % R)
if f5:
    shutil.copy(f5, "profiles/%s_bench_cfg5_kernel_stats.csv" % R)
f = newest("gpurun_out/%s_trace_legs/**/*kernel_stats.csv" % R)
if f:
    shutil.copy(f, "profiles/%s_bench_legs_kernel_stats.csv" % R)
    rows = [r for r in csv.DictReader(open(f)) if re.search(r"scan|movsum|k_pick|k_prep|k_min|k_tables|k_sort", r["Name"])]
    if rows:
        with open("profiles/%s_scan_kernel_stats.csv" % R, "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
    print("kernel stats:", f)
