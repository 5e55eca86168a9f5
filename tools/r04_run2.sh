#!/bin/bash
# round 4, GPU call: full GPU suite on the two-branch filterbank kernel, PMC passes at 512 / 1024 bins, cfg5 line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputests.txt 2>&1; echo "rc=$?" >> gpurun_out/r04_gputests.txt
tail -5 gpurun_out/r04_gputests.txt
PROBE=tools/pfb_probe.py KERNEL=pfb_kernel tools/fir_pmc.sh r04_pfb512 NB=512 > /dev/null 2>&1
PROBE=tools/pfb_probe.py KERNEL=pfb_kernel tools/fir_pmc.sh r04_pfb1024 NB=1024 > /dev/null 2>&1
python bench.py --config cfg5 --no-extras --no-cpu-baseline > gpurun_out/r04_cfg5_quick.json 2> gpurun_out/r04_cfg5_quick.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_cfg5_quick.json")); print("cfg5", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d.get("sustained",{}).get("frac_last_window"))
PY
grep -E "FETCH|WRITE|kernel_us" gpurun_out/r04_pfb512_pmc.txt gpurun_out/r04_pfb1024_pmc.txt
