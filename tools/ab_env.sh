#!/bin/bash
# usage (GPU box): tools/ab_env.sh "ENV=1 ..." "ENV2=..." ...  -- bench.py (cfg2 + also) once per environment setting ("" = default)
mkdir -p gpurun_out
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout 600 python bench.py --verbose --no-cpu-baseline --steps 100 --windows 11 > gpurun_out/abenv_$i.log 2>&1 || tail -5 gpurun_out/abenv_$i.log
  python tools/bench_compact.py gpurun_out/abenv_$i.log "[${e:-default}]"
done
