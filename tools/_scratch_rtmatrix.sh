mkdir -p gpurun_out/r05e
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
for cfg in "4 512 1.0" "4 256 1.0" "2 1024 1.0" "8 512 0.5" "4 512 2.0"; do
  set -- $cfg
  echo "== pumps $1 prep_wgs $2 window $3"
  PUMPS=$1 RCF_PREP_WGS=$2 WINDOW_MS=$3 SECONDS=4 KFIRST=1024 KCAP=1280 SHAPES=pfb256 timeout 400 python tools/rt_probe.py > gpurun_out/r05e/rt_p$1_w$2_$3.json 2> gpurun_out/r05e/err.txt || tail -3 gpurun_out/r05e/err.txt
  show gpurun_out/r05e/rt_p$1_w$2_$3.json
done
echo "== grid1600 pumps 4"
PUMPS=4 RCF_PREP_WGS=512 WINDOW_MS=1.0 SECONDS=4 KFIRST=512 KCAP=1024 SHAPES=grid1600 timeout 600 python tools/rt_probe.py > gpurun_out/r05e/rt_grid_p4.json 2> gpurun_out/r05e/err.txt || tail -3 gpurun_out/r05e/err.txt
show gpurun_out/r05e/rt_grid_p4.json
