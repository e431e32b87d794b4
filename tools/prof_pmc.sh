#!/bin/bash
# usage (GPU box): tools/prof_pmc.sh <tag> [bench args]   -- HBM traffic counters, one pass per counter
# (FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc is never combined with tracing options)
tag=$1; shift
export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/pmc_${tag}_$ctr
  mkdir -p $out
  timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $out -o c -- python bench.py --no-cpu-baseline --no-also --no-pmc --windows 2 --steps 20 --warmup 3 --ramp 0 "$@" > $out/bench.log 2>&1
  echo "[$tag $ctr] rc=$?"
done
python - "$tag" <<'PY'
import csv, glob, sys, collections, json
tag = sys.argv[1]
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob("gpurun_out/pmc_%s_%s/**/*counter_collection.csv" % (tag, ctr), recursive=True)
    if not files:
        print("no counter csv for", ctr); continue
    d = collections.defaultdict(list)
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] == ctr:
            d[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    for k, v in d.items():
        import re
        if re.search(r"cmax::(?:[tbm]\d+::)?k_", k) and len(v) >= 10:
            name = re.search(r"cmax::(?:[tbm]\d+::)?(k_\w+)", k).group(1)
            res.setdefault(name, {})[ctr] = sum(v) / len(v)
print(json.dumps(res, indent=1))
json.dump(res, open("gpurun_out/pmc_%s.json" % tag, "w"), indent=1)
PY
