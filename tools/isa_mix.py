#!/usr/bin/env python3
"""Compile one .hip file for gfx950 with -save-temps and print registers + instruction mix of kernels
whose mangled name contains the given substring.  usage: isa_mix.py file.hip substring"""
import collections, os, re, subprocess, sys
src, pat = sys.argv[1], sys.argv[2]
d = "/tmp/asm"; os.makedirs(d, exist_ok=True)
base = os.path.splitext(os.path.basename(src))[0]
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.dirname(os.path.abspath(src)),
                    *os.environ.get("XFLAGS","").split(), "-c", os.path.abspath(src), "-o", base + ".o", "-save-temps", "-Rpass-analysis=kernel-resource-usage"],
                   cwd=d, capture_output=True, text=True)
cur = None
for l in r.stderr.splitlines():
    if "Function Name:" in l: cur = l.split("Function Name:")[1].split()[0]
    if cur and pat in cur and any(k in l for k in ("VGPRs:", "ScratchSize", "Occupancy", "LDS Size")):
        print(cur[-60:], l.split("remark:")[1].split("[-R")[0].strip().split(None, 1)[1] if "remark:" in l else l)
s = open(os.path.join(d, base + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
for m in re.finditer(r'\n(_Z\w+):[^\n]*\n(.*?)s_endpgm', s, re.S):
    if pat not in m.group(1): continue
    ins = [l.strip().split()[0] for l in m.group(2).split('\n') if l.strip() and not l.strip().startswith(('.', ';')) and not l.strip().endswith(':')]
    c = collections.Counter(ins)
    print(m.group(1)[-70:], len(ins), "instrs")
    print("  " + ', '.join(f"{k}:{v}" for k, v in c.most_common(24)))
