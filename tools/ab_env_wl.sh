#!/bin/bash
# usage (GPU box): tools/ab_env_wl.sh <workload> "ENV=1 ..." "ENV2=..." ...  -- one workload, once per environment setting ("" = default)
wl=$1; shift
mkdir -p gpurun_out
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout 600 python bench.py --verbose --no-cpu-baseline --no-also --workload $wl --steps 100 --windows 11 > gpurun_out/abenvwl_$i.log 2>&1 || tail -5 gpurun_out/abenvwl_$i.log
  python tools/bench_compact.py gpurun_out/abenvwl_$i.log "[${e:-default}]"
done
