#!/usr/bin/env python
"""optimize() of the pyramid solver on a 1M-event batch (260x346) whose scene moves by 12 px and by ~150 px over the batch
(time slabs and candidate batches reach the solver classes).  patch.initialize = "global-best": the 30 x 30 grid of
src/solver/patch_contrast_base.py:164-187 through cmax_objective_batch; every later evaluation asks PatchFlowObjective.ensure_time_slabs.
Prints wall time of optimize() (same solver, repeated), of the grid search alone, the slab count per scale and the end-point error;
`--no-slabs` switches the automatic slab order off (what round 4's solver did)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import event_based_optical_flow_amd as E  # noqa: E402
from event_based_optical_flow_amd import solver  # noqa: E402
from event_based_optical_flow_amd.solver import patch_objective, translation_search  # noqa: E402

H, W, N = 260, 346, 1_000_000
t_scale = 1.0  # seconds: the grid's +-150 px/s are +-150 px of displacement
cost = sys.argv[sys.argv.index("--cost") + 1] if "--cost" in sys.argv else "image_variance"
no_slabs = "--no-slabs" in sys.argv
if no_slabs:
    E.CMaxHandle.auto_time_slabs = lambda self, px: self.time_slabs

rows = []
for v in ((12.0, -8.0), (140.0, -90.0)):
    rng = np.random.default_rng(7)
    # dots spread over the sensor + the margin they cross during the batch; as many raw events as it takes for N of them to fall on
    # the sensor (both scenes are timed on N events)
    area = (H + 2 * abs(v[0])) * (W + 2 * abs(v[1])) / (H * W)
    n_raw, n_dots = int(N * area * 1.15), int(4000 * area)
    tau = np.sort(rng.uniform(0, 1, n_raw))
    dot = rng.integers(0, n_dots, n_raw)
    cx, cy = rng.uniform(-abs(v[0]), H + abs(v[0]), n_dots), rng.uniform(-abs(v[1]), W + abs(v[1]), n_dots)
    x = np.round(cx[dot] + tau * v[0] + rng.normal(0, 0.4, n_raw))
    y = np.round(cy[dot] + tau * v[1] + rng.normal(0, 0.4, n_raw))
    keep = np.nonzero((x >= 0) & (x < H) & (y >= 0) & (y < W))[0]
    keep = np.sort(rng.choice(keep, N, replace=False)) if len(keep) > N else keep
    ev = np.stack([x, y, tau * t_scale, rng.integers(0, 2, n_raw).astype(float)], 1)[keep]
    slv_cfg = {"method": "pyramidal_patch_contrast_maximization", "time_aware": False,
               "patch": {"initialize": "global-best", "scale": 3, "crop_height": 256, "crop_width": 336, "filter_type": "bilinear", "search_grid": 0},
               "motion_model": "2d-translation", "warp_direction": "first", "parameters": ["trans_x", "trans_y"],
               "cost": cost, "outer_padding": 0, "iwe": {"method": "bilinear_vote", "blur_sigma": 0 if cost == "image_variance" else 1}}
    if cost == "hybrid":
        slv_cfg["cost_with_weight"] = {"multi_focal_normalized_gradient_magnitude": 1.0, "total_variation": 0.01}
    opt_cfg = {"n_iter": 40, "method": "Newton-CG", "max_iter": 25,
               "parameters": {"trans_x": {"min": -150, "max": 150}, "trans_y": {"min": -150, "max": 150}}}
    slv = solver.collections["pyramidal_patch_contrast_maximization"]((H, W), {}, slv_cfg, opt_cfg, {}, None)
    t_grid = []
    inner = slv.initialize_guess_from_whole_image

    def timed(handle, ts, inner=inner):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        out = inner(handle, ts)
        torch.cuda.synchronize()
        t_grid.append(time.perf_counter() - t1)
        return out

    slv.initialize_guess_from_whole_image = timed
    ev_dev = torch.from_numpy(ev).cuda()
    times = []
    for rep in range(3):
        t_grid.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        best = slv.optimize(ev)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    nfev = sum(r.nfev for _, r in slv.history)
    nhev = sum(getattr(r, "nhev", 0) for _, r in slv.history)
    flow = slv.motion_to_dense_flow(best) * t_scale
    med = np.median(flow[:, 40:-40, 40:-40].reshape(2, -1), axis=1)
    guess = slv.search_history[0][3]
    rows.append((v, min(times), t_grid[-1], nfev, nhev, slv.slab_history, guess, med, len(ev)))
    print("scene %s px over the batch, %d events, cost %s%s: optimize() %.1f ms (runs: %s), of which the 900-candidate grid %.1f ms; "
          "f / Hv callbacks %d / %d; slabs per scale %s; grid pick %s px/s; median flow %s px (truth %s)" % (
              v, len(ev), cost, " [automatic slabs OFF]" if no_slabs else "", 1e3 * min(times), ", ".join("%.1f" % (1e3 * t) for t in times),
              1e3 * t_grid[-1], nfev, nhev, slv.slab_history, guess, np.round(med, 2), v))
print("large / small motion: optimize() %.2fx, grid search %.2fx" % (rows[1][1] / rows[0][1], rows[1][2] / rows[0][2]))
