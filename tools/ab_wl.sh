#!/bin/bash
# usage (GPU box): tools/ab_wl.sh <workload> libA.so libB.so ...  -- one workload per library, compact lines ("default" = the in-tree library)
wl=$1; shift
mkdir -p gpurun_out
for lib in "$@"; do
  tag=$(basename $lib .so)
  if [ "$lib" = "default" ]; then unset CMAX_LIB; else export CMAX_LIB=$lib; fi
  timeout 600 python bench.py --verbose --no-cpu-baseline --no-also --no-pmc --workload $wl --steps 100 --windows 11 > gpurun_out/abwl_$tag.log 2>&1 || tail -5 gpurun_out/abwl_$tag.log
  python tools/bench_compact.py gpurun_out/abwl_$tag.log "[$tag]"
done
