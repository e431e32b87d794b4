#!/usr/bin/env python
"""SGPR / VGPR / occupancy / LDS per kernel from hipcc's -Rpass-analysis=kernel-resource-usage remarks:
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fno-gpu-rdc --cuda-device-only -c csrc/cmax_fused.hip \
         -o /tmp/x.o -Rpass-analysis=kernel-resource-usage 2> /tmp/res.txt
   python tools/kernel_resources.py /tmp/res.txt [regex on the demangled name]
(The rule of profiles/r02_ablation.txt: <= 80 SGPRs keeps 8 waves per SIMD; check the t512 / t1024 k_grad after every change.)"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read().split("\n")
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
cur, rows = None, {}
for l in txt:
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = m.group(1)
        rows[cur] = {}
    for key, short in (("SGPRs", "sgpr"), ("VGPRs", "vgpr"), ("Occupancy", "occ"), ("LDS Size", "lds"), ("ScratchSize", "scratch")):
        m = re.search(key + r"[^:]*: (\d+)", l)
        if m and cur:
            rows[cur].setdefault(short, int(m.group(1)))
names = list(rows)
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for k, d in zip(names, dem):
    d = d.split("(")[0].replace("void cmax::", "")
    if pat is None or pat.search(d):
        print(f"{d:64s} {rows[k]}")
