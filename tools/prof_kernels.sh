#!/bin/bash
# usage (on the GPU box): tools/prof_kernels.sh <tag> [bench args...]
# rocprofv3 kernel trace of bench.py -> per-kernel mean durations (csv kept under gpurun_out/prof_<tag>/)
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p $out
timeout 240 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python bench.py --verbose --no-cpu-baseline --no-also --no-pmc --windows 3 --ramp 0 "$@" > $out/bench.log 2>&1
echo "[$tag] rocprofv3 rc=$?"
python - "$out" "$tag" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
files = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)
if not files:
    print("no kernel trace csv under", out); sys.exit(0)
d = collections.defaultdict(list)
for row in csv.DictReader(open(files[0])):
    d[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1000.0)
rows = sorted(d.items(), key=lambda kv: -sum(kv[1]))
with open(out + "/kernel_stats.txt", "w") as f:
    f.write("%-96s %6s %9s %9s %9s %11s\n" % ("kernel", "calls", "avg_us", "min_us", "max_us", "total_us"))
    for k, v in rows:
        f.write("%-96s %6d %9.2f %9.2f %9.2f %11.1f\n" % (k[:96], len(v), sum(v) / len(v), min(v), max(v), sum(v)))
import re
hot = [(k, v) for k, v in rows if re.search(r"cmax::(?:[tbm]\d+::)?k_", k) and len(v) >= 20 or "fillBuffer" in k]
import re
print("[%s] " % tag + "  ".join("%s %.2f" % (re.search(r"cmax::(?:[tbm]\d+::)?(k_\w+)", k).group(1) if "cmax" in k else "memset", sum(v) / len(v)) for k, v in hot))
import json
for line in open(out + "/bench.log"):
    if line.startswith("{"):
        j = json.loads(line); print("[%s] ms/step %.4f value %.3e" % (tag, j["ms_per_step"], j["value"]))
PY
