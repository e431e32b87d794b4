#!/usr/bin/env python
"""Static instruction mix per kernel from a gfx950 assembly listing (hipcc --cuda-device-only -S):
   python tools/isa_mix.py /tmp/fused.s 'k_voteILi1ELb0' ['--dump' to print the body]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read().split("\n")
pat = re.compile(sys.argv[2])
dump = "--dump" in sys.argv
i = 0
while i < len(txt):
    m = re.match(r"^(_ZN4cmax\S+):\s*; @", txt[i])
    if m and pat.search(m.group(1)):
        j = i + 1
        c = collections.Counter()
        while j < len(txt) and not txt[j].startswith(".Lfunc_end"):
            l = txt[j].strip()
            if l and not l.startswith((".", ";")) and not l.endswith(":"):
                op = l.split()[0]
                kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else \
                    "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
                c[kind] += 1
                if dump:
                    print(txt[j])
            elif dump and l.endswith(":"):
                print(txt[j])
            j += 1
        print(m.group(1)[:70], dict(c), "(static counts; loops count once)")
        i = j
    i += 1
