#!/bin/bash
# usage (GPU box): tools/ab_cfg2.sh libA.so libB.so ...   -- bench.py cfg2 only (no also rows), three runs per library interleaved
mkdir -p gpurun_out
for rep in 1 2 3; do
for lib in "$@"; do
  tag=$(basename $lib .so)
  CMAX_LIB=$lib timeout 300 python bench.py --verbose --no-cpu-baseline --no-also --steps 100 --windows 21 > gpurun_out/abc_${tag}_$rep.log 2>&1 || tail -5 gpurun_out/abc_${tag}_$rep.log
  python tools/bench_compact.py gpurun_out/abc_${tag}_$rep.log "[$tag #$rep]"
done
done
