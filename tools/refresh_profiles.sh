#!/bin/bash
# Runs on the GPU box (via gpurun): regenerates every artifact under profiles/ that quotes a kernel duration.
# Outputs land in gpurun_out/refresh/; copy them into profiles/ afterwards (tools/collect_profiles.py <tag>).
export TMPDIR=/tmp
tag=${1:-r03}
out=gpurun_out/refresh
mkdir -p $out
for w in cfg2 cfg3 cfg4 cfg5 cfg5_strong hbm; do
  st=100; [ $w = hbm ] && st=20; [ $w = cfg5_strong ] && st=40
  bash tools/prof_kernels.sh $w --workload $w --steps $st
  cp gpurun_out/prof_$w/kernel_stats.txt $out/kernel_stats_$w.txt
done
# the official summary of the default bench command (csv output: the default rocpd database made --stats hang here)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats -o $tag -- python bench.py --verbose --no-cpu-baseline --no-also --no-pmc --steps 200 --warmup 20 --ramp 0 > $out/rocprofv3_bench_line_cfg2.txt 2>&1
echo "rocprofv3 --stats rc=$?"
f=$(find gpurun_out/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/rocprofv3_kernel_stats_cfg2.csv
for w in cfg2 cfg3 cfg4 cfg5 cfg5_strong hbm; do
  bash tools/prof_pmc.sh $w --workload $w
  cp gpurun_out/pmc_$w.json $out/pmc_${w}_raw.json
done
for w in cfg2 cfg3 cfg4 cfg5 cfg5_strong hbm; do
  bash tools/prof_sq.sh $w --workload $w > $out/sq_$w.txt 2>&1
  cp gpurun_out/sq_$w.json $out/sq_$w.json
done
timeout 900 python bench.py --verbose > $out/bench_cfg2.json 2> $out/bench_cfg2.err; echo "bench rc=$?"; python tools/bench_compact.py $out/bench_cfg2.json "[bench]"
timeout 600 python bench.py --verbose --no-cpu-baseline --no-pmc --deterministic --steps 50 --windows 5 > $out/bench_deterministic.json 2>&1; python tools/bench_compact.py $out/bench_deterministic.json "[deterministic]"
./tools/microbench_launch > $out/launch_floor.txt 2>&1; cat $out/launch_floor.txt
python tools/bench_solver_objective.py > $out/solver_objective.txt 2>&1; tail -8 $out/solver_objective.txt
python tools/bench_solver_optimize.py > $out/solver_optimize.txt 2>&1; tail -6 $out/solver_optimize.txt
# the N > 1 code path on one GPU: two ranks sharing cuda:0 over gloo (RCCL refuses two ranks per device)
timeout 900 python bench.py --verbose --gpus 2 --share-gpu --backend gloo --steps 20 --warmup 5 --windows 5 --ramp 0.2 --no-cpu-baseline --no-pmc > $out/bench_2ranks_shared_gpu.json 2> $out/bench_2ranks.err; echo "2-rank bench rc=$?"; python tools/bench_compact.py $out/bench_2ranks_shared_gpu.json "[2 ranks, one GPU, gloo]"
# raw traces are large (kernel-trace csv of the 64M-event runs): gpurun merges at most 64 MiB back -- keep the summaries only
rm -rf gpurun_out/prof_* gpurun_out/pmc_*_FETCH_SIZE gpurun_out/pmc_*_WRITE_SIZE gpurun_out/sq_*_p? gpurun_out/stats
du -sh gpurun_out
