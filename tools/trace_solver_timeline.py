#!/usr/bin/env python
"""Kernel timeline of ONE solver-objective call (value+gradient or exact HVP; plain or time-aware Burgers) on the cfg1 shape
(30k events, 260x346, YAML hybrid cost, 16x16 patches): runs itself under `rocprofv3 --kernel-trace` and prints start offset,
duration and name of every kernel of the last call.  This is how the launch sequence of cmax_patch_plan_* was trimmed
(profiles/r01_ablation.txt).   usage (GPU box):  python tools/trace_solver_timeline.py [--burgers] [--hvp]"""
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(burgers: bool, hvp: bool):
    import numpy as np
    import torch

    import event_based_optical_flow_amd as E
    from event_based_optical_flow_amd.solver import PatchFlowObjective
    from event_based_optical_flow_amd.solver.scipy_autograd import TorchWrapper

    H, W, N = 260, 346, 30000
    ev = E.utils.generate_events(N, H, W, 0.0, 0.05, seed=46)
    cww = {"multi_focal_normalized_gradient_magnitude": 1.0, "total_variation": 0.01}
    h = E.CMaxHandle((H, W)).set_events(ev, time_bin=10 if burgers else 0)
    obj = PatchFlowObjective(h, 0.05, (16, 16), (16, 21), (16, 21), (2, 5), cost="hybrid", cost_with_weight=cww, blur_sigma=1,
                             time_aware=burgers, time_bin=10)
    w = TorchWrapper(obj, precision="float64", device="cuda")
    x = w.get_input(np.random.default_rng(0).uniform(-100, 100, 512))
    v = np.random.default_rng(1).normal(size=512)
    for _ in range(40):
        w.get_hvp(x, v) if hvp else w.get_value_and_grad(x)
    torch.cuda.synchronize()
    print("MARK", flush=True)  # everything after the last k_patch_tail but one belongs to the last call


def main():
    burgers, hvp = "--burgers" in sys.argv, "--hvp" in sys.argv
    if "--child" in sys.argv:
        return child(burgers, hvp)
    out = os.path.join(ROOT, "gpurun_out", "timeline")
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "t", "--", sys.executable, os.path.abspath(__file__), "--child"]
    cmd += [a for a in ("--burgers", "--hvp") if a in sys.argv]
    subprocess.run(cmd, env=env, timeout=300, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    files = sorted(glob.glob(out + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
    rows = sorted(csv.DictReader(open(files[-1])), key=lambda r: int(r["Start_Timestamp"]))
    tails = [i for i, r in enumerate(rows) if "k_patch_tail" in r["Kernel_Name"]]
    last = rows[tails[-2] + 1: tails[-1] + 1]
    t0 = int(last[0]["Start_Timestamp"])
    print("%s %s: %d launches, %.1f us from the first start to the last end (under the profiler the host enqueues more slowly)" % (
        "burgers" if burgers else "plain", "hvp" if hvp else "value+grad", len(last), (int(last[-1]["End_Timestamp"]) - t0) / 1e3))
    for r in last:
        print("%8.2f %7.2f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:90]))


if __name__ == "__main__":
    main()
