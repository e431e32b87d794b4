#!/bin/bash
# usage (GPU box): tools/prof_sq.sh <tag> [bench args]   -- SQ counters of the event kernels, 8 per pass (the SQ block
# has 8 slots on gfx950; --pmc is never combined with tracing options).  Result: gpurun_out/sq_<tag>.json =
# {kernel: {counter: mean per launch}} plus derived ratios.
tag=$1; shift
export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU"
P3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES"
i=0
for set in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  out=gpurun_out/sq_${tag}_p$i
  mkdir -p $out
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $out -o c -- python bench.py --no-cpu-baseline --no-also --no-pmc --steps 20 --warmup 3 --windows 2 --ramp 0 "$@" > $out/bench.log 2>&1
  echo "[$tag pass $i] rc=$?"
done
python - "$tag" <<'PY'
import csv, glob, sys, collections, json, re
tag = sys.argv[1]
res = collections.defaultdict(dict)
for p in (1, 2, 3):
    files = glob.glob("gpurun_out/sq_%s_p%d/**/*counter_collection.csv" % (tag, p), recursive=True)
    if not files:
        print("no counter csv for pass", p); continue
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(files[0])):
        d[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, ctrs in d.items():
        m = re.search(r"cmax::(?:([tbm]\d+)::)?(k_\w+)", k)
        if not m:
            continue
        name = (m.group(1) + "::" if m.group(1) else "") + m.group(2)
        for c, v in ctrs.items():
            if len(v) >= 10:
                res[name][c] = sum(v) / len(v)
out = {}
for k, c in res.items():
    o = dict(c)
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        for a in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM",
                  "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_LDS"):
            if a in c:
                o["frac_of_wave_cycles:" + a] = c[a] / wc
    if c.get("SQ_LDS_IDX_ACTIVE"):
        o["lds_bank_conflict_rate"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    if c.get("SQ_WAVES"):
        for a in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_INSTS_LDS_ATOMIC"):
            if a in c:
                o["per_wave:" + a] = c[a] / c["SQ_WAVES"]
    out[k] = o
json.dump(out, open("gpurun_out/sq_%s.json" % tag, "w"), indent=1, sort_keys=True)
for k, o in out.items():
    print(k, {a: round(v, 3) for a, v in o.items() if a.startswith(("frac_", "per_wave", "lds_"))})
PY
