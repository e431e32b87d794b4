#!/bin/bash
# Runs on the GPU box (via gpurun): smoke, GPU parity tests, bench.  Logs go to gpurun_out/.
mkdir -p gpurun_out
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 ${PYTEST_ARGS} > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; tail -${PYTEST_TAIL:-15} gpurun_out/pytest.log
for wl in ${WORKLOADS:-cfg2}; do
  timeout 600 python bench.py --workload $wl --steps ${STEPS:-200} --warmup 20 ${BENCH_ARGS} > gpurun_out/bench_$wl.log 2>&1; echo "bench $wl rc=$?"
  python - <<PY
import json
for line in open("gpurun_out/bench_$wl.log"):
    if line.startswith("{"):
        d = json.loads(line)
        r = d["roofline"]
        print("$wl value %.3e ev/s  ms/step %.4f  dominant %s %.1f us frac %.3f  all %s  eval_frac %.3f  cpu %s" % (
            d["value"], d["ms_per_step"], r["kernel"].split()[0], r["launch_us"], r["frac"],
            {k: round(v, 1) for k, v in r["all_kernels_us"].items()}, r["evaluation_frac"],
            ("%.2e" % d["cpu_baseline"]["value"]) if "cpu_baseline" in d else "-"))
        break
else:
    print(open("gpurun_out/bench_$wl.log").read()[-2000:])
PY
done
