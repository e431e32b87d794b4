#!/usr/bin/env python
"""Compact view of bench.py's JSON line(s): python tools/bench_compact.py <log> [label]"""
import json
import sys

label = sys.argv[2] if len(sys.argv) > 2 else ""
for line in open(sys.argv[1]):
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d["roofline"]
    ks = {k: round(v.get("launch_us", v.get("single_launch_bracket_us")), 2) for k, v in r["kernels"].items()}
    print("%s %s: %.2f us/eval (min %.2f max %.2f) %.3e ev/s eval-frac %.3f kernels %s" % (
        label, d["config"]["workload"].split(":")[0], d["ms_per_step"] * 1e3, d["timing"]["min"] * 1e3, d["timing"]["max"] * 1e3, d["value"], r["frac"], ks))
    for k, a in d.get("also", {}).items():
        print("%s   %s: %.2f us/eval %.3e ev/s eval-frac %.3f kernels %s" % (
            label, k, a["ms_per_step"] * 1e3, a["value"], a["evaluation_frac"], {q: round(v, 2) for q, v in a["kernels_us"].items()}))
