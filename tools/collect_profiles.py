#!/usr/bin/env python
"""Copies what tools/refresh_profiles.sh / prof_pmc.sh left under gpurun_out/ into profiles/ (tracked), with the
command lines as headers and the PMC counters converted to bytes per launch as MI355X_MICROARCH.md prescribes
(FETCH_SIZE: KiB, doubled for the 16 B/lane coalesced event stream; WRITE_SIZE as reported)."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")

for wl in ("cfg2", "cfg3", "cfg4", "cfg5", "cfg5_strong", "hbm"):
    f = os.path.join(src, "refresh", "kernel_stats_%s.txt" % wl)
    if os.path.exists(f):
        with open(os.path.join(dst, "%s_kernel_stats_%s.txt" % (tag, wl)), "w") as out:
            out.write("# rocprofv3 --kernel-trace --output-format csv -- python bench.py --no-cpu-baseline --no-also --windows 3 --workload %s "
                      "--steps 100; per-kernel durations in microseconds, computed from the kernel-trace csv by tools/prof_kernels.sh\n" % wl)
            out.write(open(f).read())
for name in ("rocprofv3_kernel_stats_cfg2.csv", "bench_cfg2.json", "launch_floor.txt", "solver_optimize.txt"):
    f = os.path.join(src, "refresh", name)
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, "%s_%s" % (tag, name)))
f = os.path.join(src, "refresh", "rocprofv3_bench_line_cfg2.txt")
if os.path.exists(f):
    with open(os.path.join(dst, "%s_rocprofv3_bench_line_cfg2.txt" % tag), "w") as out:
        out.write("# command: rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats -o %s -- python bench.py "
                  "--no-cpu-baseline --no-also --steps 200 --warmup 20\n# (under the profiler the host-side launch path is slower, so ms_per_step "
                  "here is NOT the bench number -- see %s_bench_cfg2.json; kernel durations are unaffected)\n" % (tag, tag))
        out.writelines(l for l in open(f) if l.startswith("{"))
f = os.path.join(src, "refresh", "solver_objective.txt")
if os.path.exists(f):
    lines = [l for l in open(f) if "ms per call" in l]
    with open(os.path.join(dst, "%s_solver_objective.txt" % tag), "w") as out:
        out.write("# python tools/bench_solver_objective.py -- cfg1-shaped objective through the optimiser boundary (30k events, 260x346,\n"
                  "# shipped YAML hybrid cost, 16x16 patches = 512 DoF), wall time per call incl. the host round trip.\n"
                  "# native = one cmax_patch_plan_* call; autograd = the same kernels chained by torch.autograd.\n"
                  "# reference (torch-CPU fp64, BASELINE.md): 126 / 231 ms plain, 294 / 1098 ms Burgers (value+grad / hvp)\n")
        out.writelines(lines)
short = {"k_vote": "vote", "k_stats": "stats", "k_gimage": "gimage", "k_grad": "grad", "k_finish": "finish", "k_finish_deferred": "finish", "k_finish_raw": "finish",
         "k_stats_gimage_gm": "stats", "k_blur_stats_gimage_gm": "stats", "k_blur_stats_var": "stats", "k_gimage_blur_adj_var": "gimage", "k_blur_stats_adj_var": "stats"}
for wl in ("cfg2", "cfg3", "cfg4", "cfg5", "cfg5_strong", "hbm"):
    f = os.path.join(src, "refresh", "pmc_%s_raw.json" % wl)
    if not os.path.exists(f):
        f = os.path.join(src, "pmc_%s.json" % wl)
    if not os.path.exists(f):
        continue
    raw = json.load(open(f))
    traffic = {}
    for k, v in raw.items():
        if k in short and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            traffic[short[k]] = int(round((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024))
    doc = {
        "_comment": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing), python bench.py --no-cpu-baseline "
                    "--steps 20 --warmup 3 --workload %s. Counter values are KiB per launch, averaged over launches. FETCH_SIZE is "
                    "doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of a wide 16 B/lane coalesced stream; the event stream is "
                    "read with 16-byte loads); WRITE_SIZE is uncalibrated and taken as reported." % wl,
        "raw_KiB": raw,
        "traffic_bytes_per_launch": {wl: traffic},
    }
    json.dump(doc, open(os.path.join(dst, "%s_pmc_%s.json" % (tag, wl)), "w"), indent=1)
    print(wl, traffic)

# SQ counters of the event kernels (tools/prof_sq.sh: three passes of 8 SQ counters, no tracing)
for wl in ("cfg2", "cfg3", "cfg4", "cfg5", "cfg5_strong", "hbm"):
    f = os.path.join(src, "refresh", "sq_%s.json" % wl)
    if os.path.exists(f):
        doc = {"_comment": "rocprofv3 --pmc <8 SQ counters> x 3 passes -- python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 3 "
                           "--windows 2 --workload %s (tools/prof_sq.sh); values are means per launch summed over the chip; SQ_*CYCLES and "
                           "SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md); frac_of_wave_cycles:X = X / SQ_WAVE_CYCLES; "
                           "per_wave:X = X / SQ_WAVES; lds_bank_conflict_rate = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE" % wl,
               "kernels": json.load(open(f))}
        json.dump(doc, open(os.path.join(dst, "%s_sq_%s.json" % (tag, wl)), "w"), indent=1, sort_keys=True)
for name, header in (("bench_deterministic.json", "# python bench.py --no-cpu-baseline --deterministic --steps 50 --windows 5 (cmax_set_deterministic: integer accumulation)\n"),
                     ("bench_2ranks_shared_gpu.json", "# python bench.py --gpus 2 --share-gpu --backend gloo --steps 20 --warmup 5 --windows 5 --no-cpu-baseline: the N > 1 CODE PATH on a\n"
                                                     "# 1-GPU box (self-launched ranks, both on cuda:0, gloo + torch collectives because RCCL refuses two ranks per device).\n"
                                                     "# NOT a scaling number: two processes time-share one GPU and every all-reduce goes through the host.\n")):
    f = os.path.join(src, "refresh", name)
    if os.path.exists(f):
        with open(os.path.join(dst, "%s_%s" % (tag, name.replace(".json", ".txt"))), "w") as out:
            out.write(header)
            out.writelines(l for l in open(f) if l.startswith("{"))
