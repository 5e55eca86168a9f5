#!/bin/bash
cd /root/repo
python -m pytest tests -m gpu -q -k "pfb or tap" 2>&1 | tail -3
for nb in 1600 3200 800; do
  NB=$nb BLOCK=33554432 python tools/pfb_probe.py 2>&1 | tail -1 | cut -c1-100
done
NB=1600 BLOCK=16777216 python tools/pfb_probe.py 2>&1 | tail -1 | cut -c1-100
NB=1600 BLOCK=33554432 TAPS=256 python tools/pfb_probe.py 2>&1 | tail -1 | cut -c1-100
