#!/bin/bash
# usage (GPU box): tools/prof_set_events.sh <tag> [HxW:n ...]   -- rocprofv3 kernel trace of tools/probe_set_events.py, one line per
# kernel of the per-batch pipeline (average over the calls of a case is not separated: run one case per invocation for clean numbers)
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/prof_se_$tag
mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out -o t -- python tools/probe_set_events.py "$@" > $out/probe.log 2>&1
echo "[$tag] rocprofv3 rc=$?"; grep "events:" $out/probe.log
python - "$out" "$tag" <<'PY'
import csv, glob, sys, collections, re
out, tag = sys.argv[1], sys.argv[2]
files = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)
if not files:
    print("no kernel trace csv under", out); sys.exit(0)
d = collections.defaultdict(list)
for row in csv.DictReader(open(files[0])):
    d[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1000.0)
rows = [(k, v) for k, v in d.items() if re.search(r"cmax::(\w+::)?k_(sort|bucket|scan|tile|run|slab|tmm|rs|pack)", k)]
rows.sort(key=lambda kv: -sum(kv[1]))
with open(out + "/kernel_stats.txt", "w") as f:
    for k, v in rows:
        name = re.search(r"cmax::(?:\w+::)?(k_\w+)", k).group(1)
        v = sorted(v)
        line = "[%s] %-20s calls %4d  median %9.2f us  min %9.2f  max %9.2f" % (tag, name, len(v), v[len(v) // 2], v[0], v[-1])
        print(line); f.write(line + "\n")
PY
