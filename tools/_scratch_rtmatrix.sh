mkdir -p gpurun_out/r05g
show() { python - "$1" <<'PY'
import json,sys
r=json.load(open(sys.argv[1]))
for k,v in r.items():
    if isinstance(v, dict):
        print(k, "K_max", v["K_max"], "first_attempt", v["K_max_first_attempt"])
        for p in v["points"]:
            print("   ", {a: (round(b,2) if isinstance(b,float) else b) for a,b in p.items() if a not in ("errors","seconds","gpu_busy_percent_est") or (a=="errors" and b)})
PY
}
for cfg in "4 512 1.0" "8 512 1.0" "4 256 1.0"; do
  set -- $cfg
  echo "== grid1600 pumps $1 prep_wgs $2 window $3"
  PUMPS=$1 RCF_PREP_WGS=$2 WINDOW_MS=$3 SECONDS=4 KFIRST=768 KCAP=1024 SHAPES=grid1600 timeout 500 python tools/rt_probe.py > gpurun_out/r05g/rt_grid_p$1_$3.json 2> gpurun_out/r05g/err.txt || tail -3 gpurun_out/r05g/err.txt
  show gpurun_out/r05g/rt_grid_p$1_$3.json
done
cd /tmp && export TMPDIR=/tmp
PUMPS=4 WINDOW_MS=1.0 SECONDS=3 KFIRST=768 KCAP=768 SHAPES=grid1600 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05g/trace_grid1600 -- python $GRAFT_REPO_ROOT/tools/rt_probe.py > $GRAFT_REPO_ROOT/gpurun_out/r05g/rt_prof_grid1600.json 2>/dev/null
cd $GRAFT_REPO_ROOT/gpurun_out/r05g; f=$(find trace_grid1600 -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-180; cp $f trace_grid1600_kernel_stats.csv; rm -rf trace_grid1600
