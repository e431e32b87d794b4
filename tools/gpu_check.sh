#!/bin/bash
# Runs on the GPU box (via gpurun): smoke, GPU parity tests, bench.  Logs go to gpurun_out/.
mkdir -p gpurun_out
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 ${PYTEST_ARGS} > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; tail -${PYTEST_TAIL:-15} gpurun_out/pytest.log
for wl in ${WORKLOADS:-cfg2}; do
  timeout 600 python bench.py --verbose --workload $wl --steps ${STEPS:-200} --warmup 20 ${BENCH_ARGS} > gpurun_out/bench_$wl.log 2>&1; echo "bench $wl rc=$?"
  python - <<PY
import json
for line in open("gpurun_out/bench_$wl.log"):
    if line.startswith("{"):
        d = json.loads(line)
        r = d["roofline"]
        print("$wl value %.3e ev/s  ms/step %.4f (min %.4f max %.4f)  eval frac %.3f  dominant %s %.1f us frac %.3f  kernels %s  cpu %s" % (
            d["value"], d["ms_per_step"], d["timing"]["min"], d["timing"]["max"], r["frac"], r["dominant"]["kernel"].split()[0],
            r["dominant"]["launch_us"], r["dominant"]["frac"],
            {k: round(v.get("launch_us", v.get("single_launch_bracket_us")), 1) for k, v in r["kernels"].items()},
            ("%.2e" % d["cpu_baseline"]["value"]) if "cpu_baseline" in d else "-"))
        for k, a in d.get("also", {}).items():
            print("   also %s: ms/step %.4f value %.3e eval frac %.3f dominant %s %.1f us frac %.3f  kernels %s" % (
                k, a["ms_per_step"], a["value"], a["evaluation_frac"], (a["dominant_kernel"] or "-").split()[0], a["dominant_kernel_us"] or 0,
                a["dominant_kernel_frac"] or 0, {q: round(v, 1) for q, v in a["kernels_us"].items()}))
        break
else:
    print(open("gpurun_out/bench_$wl.log").read()[-2000:])
PY
done
