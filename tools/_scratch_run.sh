mkdir -p gpurun_out/r05i
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null; nproc; python -c "import os; print(len(os.sched_getaffinity(0)))"
show() { python - "$1" <<'PY'
import json,sys
r=json.load(open(sys.argv[1]))
for k,v in r.items():
    if isinstance(v, dict):
        print(k, "K_max", v["K_max"], "first_attempt", v["K_max_first_attempt"])
        for p in v["points"]:
            print("   ", {a: (round(b,2) if isinstance(b,float) else b) for a,b in p.items() if a not in ("errors","seconds","gpu_busy_percent_est","ok") or (a=="errors" and b)})
PY
}
for cfg in "grid1600 768 1024" "pfb256 1024 1280"; do
set -- $cfg
echo "== $cfg"
RCF_PUMP_DEBUG=1 PUMPS=4 WINDOW_MS=1.0 SECONDS=4 KFIRST=$2 KCAP=$3 SHAPES=$1 timeout 600 python tools/rt_probe.py > gpurun_out/r05i/c_$1.json 2> gpurun_out/r05i/err.txt || tail -3 gpurun_out/r05i/err.txt
grep "^pump:" gpurun_out/r05i/err.txt | sort | tail -3
show gpurun_out/r05i/c_$1.json
done
