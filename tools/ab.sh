#!/bin/bash
# usage (GPU box): tools/ab.sh libA.so libB.so ...   -- bench.py (cfg2 + also) once per library, compact lines
mkdir -p gpurun_out
for lib in "$@"; do
  tag=$(basename $lib .so)
  CMAX_LIB=$lib timeout 600 python bench.py --verbose --no-cpu-baseline --no-pmc --steps 100 --windows 11 > gpurun_out/ab_$tag.log 2>&1 || tail -5 gpurun_out/ab_$tag.log
  python tools/bench_compact.py gpurun_out/ab_$tag.log "[$tag]"
done
