#!/usr/bin/env python
"""End-to-end timing of the coarse-to-fine solver with the parameters of the shipped YAML configs
(configs/mvsec_indoor_no_timeaware.yaml / mvsec_indoor_burgers.yaml: 260x346, 30k events per batch, crop 256x336,
pyramid scales 1..4, hybrid cost = multi-focal normalised gradient magnitude + 0.01 TV, blur sigma 1, Newton-CG with
max_iter 25) on a synthetic scene: dots moving along a smooth ground-truth flow.  Reports wall time of
`optimize(events)` (pack + sort, every scale, host round trips included), optimiser callbacks and the end-point error."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import event_based_optical_flow_amd as E  # noqa: E402
from event_based_optical_flow_amd import solver  # noqa: E402
from event_based_optical_flow_amd.solver import scipy_autograd  # noqa: E402

H, W, N = 260, 346, 30000
rng = np.random.default_rng(11)
V = E.utils.generate_smooth_flow((H, W), 12.0, grid=3, seed=12)  # pixel displacement over the batch
n_dots = 1500
cx, cy = rng.uniform(8, H - 8, n_dots), rng.uniform(8, W - 8, n_dots)
dot = rng.integers(0, n_dots, N)
tau = np.sort(rng.uniform(0, 1, N))
vx, vy = V[0, cx.astype(int), cy.astype(int)][dot], V[1, cx.astype(int), cy.astype(int)][dot]
x = np.clip(np.round(cx[dot] + tau * vx + rng.normal(0, 0.4, N)), 0, H - 1)
y = np.clip(np.round(cy[dot] + tau * vy + rng.normal(0, 0.4, N)), 0, W - 1)
t_scale = 0.05
ev = np.stack([x, y, tau * t_scale, rng.integers(0, 2, N).astype(float)], 1)
mask = np.zeros((H, W), bool)
mask[x.astype(int), y.astype(int)] = True
mask[:8] = mask[-8:] = False
mask[:, :8] = mask[:, -8:] = False

for time_aware, grid in ((False, None), (False, 0), (False, 16), (True, None), (True, 0)):
    slv_cfg = {"method": "pyramidal_patch_contrast_maximization", "time_aware": time_aware,
               "patch": {"initialize": "random", "scale": 5, "crop_height": 256, "crop_width": 336, "filter_type": "bilinear"},
               "motion_model": "2d-translation", "warp_direction": "first", "parameters": ["trans_x", "trans_y"],
               "cost": "hybrid", "outer_padding": 0,
               "cost_with_weight": {"multi_focal_normalized_gradient_magnitude": 1.0, "total_variation": 0.01},
               "iwe": {"method": "bilinear_vote", "blur_sigma": 1}}
    if time_aware:
        slv_cfg.update({"time_bin": 10, "flow_interpolation": "burgers", "t0_flow_location": "middle"})
    opt_cfg = {"n_iter": 40, "method": "Newton-CG", "max_iter": 25,
               "parameters": {"trans_x": {"min": -150, "max": 150}, "trans_y": {"min": -150, "max": 150}}}
    if grid is not None:
        slv_cfg["patch"]["search_grid"] = grid  # None: the reference's trial budget; 0: no per-patch re-initialisation
    times, t_search = [], []
    # one solver for all repetitions, like main.py's loop over a sequence: the first call creates the device workspaces
    # and the per-scale objectives, the later ones reuse them
    slv = solver.collections["pyramidal_patch_contrast_maximization"]((H, W), {}, slv_cfg, opt_cfg, {}, None)
    inner = slv.initialize_guess_from_patch_search

    def timed(handle, s, m0, inner=inner):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        out = inner(handle, s, m0)
        torch.cuda.synchronize()
        t_search.append(time.perf_counter() - t1)
        return out

    slv.initialize_guess_from_patch_search = timed
    for rep in range(4):
        np.random.seed(46)
        t_search.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        best = slv.optimize(ev)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    nfev = sum(r.nfev for _, r in slv.history)
    njev = sum(getattr(r, "njev", 0) for _, r in slv.history)
    nhev = sum(getattr(r, "nhev", 0) for _, r in slv.history)
    flow = slv.motion_to_dense_flow(best) * t_scale
    if flow.ndim == 4:  # time-aware: the voxel's slice at the original time
        flow = slv.get_original_flow_from_time_aware_flow_voxel(flow)
    aee = np.sqrt(((flow - V) ** 2).sum(0))[mask].mean()
    aee0 = np.sqrt((V ** 2).sum(0))[mask].mean()
    n_pairs = sum(c.shape[0] * c.shape[1] for _, c, _, _ in slv.search_history)
    print("%-8s search_grid %-4s optimize(): %.3f s (first call, then the same solver again: %s)  scales %s  f/g/Hv callbacks %d/%d/%d  end-point error %.2f px (zero flow: %.2f px)"
          "  per-patch search: %d (patch, candidate) pairs in %.2f ms over %d scales" % (
              "burgers" if time_aware else "plain", grid, min(times), ", ".join("%.3f" % t for t in times), sorted(best), nfev, njev, nhev,
              aee, aee0, n_pairs, 1e3 * sum(t_search), len(t_search)))
