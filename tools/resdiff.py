#!/usr/bin/env python
"""Side-by-side kernel resources of two -Rpass-analysis remark files (tools/resources.sh): resdiff.py old.txt new.txt [regex]
Only kernels whose SGPR / VGPR / occupancy / LDS / scratch differ are printed."""
import re, subprocess, sys

def parse(path):
    cur, rows = None, {}
    for l in open(path):
        m = re.search(r"Function Name: (\S+)", l)
        if m:
            cur = m.group(1); rows[cur] = {}
        for key, short in (("SGPRs", "s"), ("VGPRs", "v"), ("Occupancy", "occ"), ("LDS Size", "lds"), ("ScratchSize", "scr")):
            m = re.search(key + r"[^:]*: (\d+)", l)
            if m and cur: rows[cur].setdefault(short, int(m.group(1)))
    return rows
a, b = parse(sys.argv[1]), parse(sys.argv[2])
pat = re.compile(sys.argv[3]) if len(sys.argv) > 3 else None
names = sorted(set(a) | set(b))
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for k, d in zip(names, dem):
    d = d.split("(")[0].replace("void cmax::", "")
    if pat and not pat.search(d): continue
    if a.get(k) != b.get(k): print(f"{d:56s} {a.get(k)} -> {b.get(k)}")
