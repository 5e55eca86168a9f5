#!/bin/bash
cd /root/repo
python -m pytest tests -m gpu -q -k "pfb or tap or reference_grid or smoke" 2>&1 | tail -2
for i in 1 2; do
NB=1600 BLOCK=33554432 python tools/pfb_probe.py 2>&1 | tail -1 | cut -c1-90
done
NB=3200 BLOCK=16777216 python tools/pfb_probe.py 2>&1 | tail -1 | cut -c1-90
NB=1600 BLOCK=33554432 TAPS=256 python tools/pfb_probe.py 2>&1 | tail -1 | cut -c1-90
