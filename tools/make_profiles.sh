#!/bin/bash
# Regenerate profiles/ for a round on the GPU box:  tools/make_profiles.sh r01
#   1. bench.py JSON line
#   2. rocprofv3 --kernel-trace --stats of the same command (kernel summary CSV)
#   3. separate --pmc passes (FETCH_SIZE, WRITE_SIZE) -> profiles/pfb_traffic.json
# Counter passes never share a run with trace domains other than --kernel-trace.
set -u
R=${1:-r01}
cd "$(dirname "$0")/.."
ROOT=$PWD
mkdir -p profiles gpurun_out
export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 20 --warmup 3"

python bench.py --steps 20 --warmup 3 > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err
tail -1 gpurun_out/${R}_bench.json > profiles/${R}_bench.json

(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${R}_trace -- $CMD --no-cpu-baseline > /dev/null 2>&1)
S=$(find gpurun_out/${R}_trace -name "*kernel_stats.csv" | head -1)
[ -n "$S" ] && cp "$S" profiles/${R}_bench_kernel_stats.csv

for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $ROOT/gpurun_out/${R}_pmc_$c -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
done
python - "$R" <<'EOF'
import collections, csv, glob, json, sys
R = sys.argv[1]
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for f in glob.glob("gpurun_out/%s_pmc_%s/**/*counter_collection.csv" % (R, c), recursive=True):
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
    json.dump({"block": B, "kernel": name.split("(")[0][:80], "launches_averaged": vals["FETCH_SIZE"][1],
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
EOF
cat profiles/${R}_bench.json
head -12 profiles/${R}_bench_kernel_stats.csv
