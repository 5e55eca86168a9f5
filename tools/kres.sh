#!/bin/bash
# kernel resource table of one .hip file:  tools/kres.sh pfb.hip [filter]   (name, VGPRs, spills, SGPRs, occupancy)
cd "$(dirname "$0")/../radiocapture-rf_amd/csrc"
F=${1:-pfb.hip}; PAT=${2:-.}
EXTRA=""; case $F in pfb.hip|scan.hip) EXTRA="-fno-slp-vectorize";; pfb5.hip|tapfin.hip) EXTRA="-fno-slp-vectorize -ffp-contract=off";; fir.hip|peaks.hip|audio.hip) EXTRA="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form $EXTRA $KFLAGS -c $F -o /tmp/kres_$$.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re,subprocess
cur=None; rows={}
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)|remark: [^ ]+ Name: (\S+)|  Name: (\S+)",l)
    if "Name:" in l:
        nm=l.split("Name:")[1].split()[0]; cur=nm; rows[cur]={}
    for k in ("VGPRs:","VGPRs Spill:","TotalSGPRs:","Occupancy [waves/SIMD]:","ScratchSize [bytes/lane]:"):
        if cur and k in l: rows[cur][k]=l.split(k)[1].split()[0]
names=list(rows)
dem=subprocess.run(["c++filt"]+names,capture_output=True,text=True).stdout.split("\n")
for n,d in zip(names,dem):
    r=rows[n]; d=d.replace("rcfx::(anonymous namespace)::","").replace("(rcfx::PfbLaunch, int)","")
    print("%-60s vgpr %4s spill %3s sgpr %3s occ %s scratch %s"%(d[:60],r.get("VGPRs:"),r.get("VGPRs Spill:"),r.get("TotalSGPRs:"),r.get("Occupancy [waves/SIMD]:"),r.get("ScratchSize [bytes/lane]:")))
' | grep -E "$PAT"
rm -f /tmp/kres_$$.o
