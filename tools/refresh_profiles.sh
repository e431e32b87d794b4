#!/bin/bash
# Runs on the GPU box (via gpurun): regenerates every artifact under profiles/ that quotes a kernel duration.
# Outputs land in gpurun_out/refresh/; copy them into profiles/ afterwards (tools/collect_profiles.py).
export TMPDIR=/tmp
out=gpurun_out/refresh
mkdir -p $out
for w in cfg2 cfg3 cfg4 cfg5; do
  bash tools/prof_kernels.sh $w --workload $w --steps 100
  cp gpurun_out/prof_$w/kernel_stats.txt $out/kernel_stats_$w.txt
done
# the official summary of the default bench command (csv output: the default rocpd database made --stats hang here)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats -o r01 -- python bench.py --no-cpu-baseline --steps 200 --warmup 20 > $out/rocprofv3_bench_line_cfg2.txt 2>&1
echo "rocprofv3 --stats rc=$?"
f=$(find gpurun_out/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/rocprofv3_kernel_stats_cfg2.csv
bash tools/prof_pmc.sh cfg2 --workload cfg2
cp gpurun_out/pmc_cfg2.json $out/pmc_cfg2_raw.json
timeout 600 python bench.py > $out/bench_cfg2.json 2> $out/bench_cfg2.err; echo "bench rc=$?"; tail -c 600 $out/bench_cfg2.json
python tools/bench_solver_objective.py > $out/solver_objective.txt 2>&1; tail -8 $out/solver_objective.txt
