#!/usr/bin/env python
"""Does the evaluation time follow the NUMBER OF ROUNDS of workgroups (1024 resident 512-thread workgroups on 256 CUs) rather than the
number of events?  One evaluation (K1 + image kernel + K3) for growing batches of the cfg5-shard shape (720p, dense flow, variance) and
of the cfg4 shape (260x346, voxel T = 10, blurred variance); CMAX_DEBUG_SEGS=1 prints the size of each work list.
us per evaluation (median of 9 windows of 50) | K1 | K3 (8-launch HIP-event brackets)."""
import os
import sys
import time

import numpy as np
import torch

os.environ.setdefault("CMAX_DEBUG_SEGS", "1")
sys.path.insert(0, ".")
import event_based_optical_flow_amd as E

shapes = {
    "dense720p": dict(H=720, W=1280, model="dense-flow", cost="image_variance", sigma=0.0, T=0, ns=[1.0, 1.25, 1.5, 1.75, 2.0, 2.25, 2.5, 2.75, 3.0, 3.5, 4.0]),
    "dense240p": dict(H=260, W=346, model="dense-flow", cost="image_variance", sigma=0.0, T=0, ns=[0.5, 1.0, 2.0]),
    "dense480p": dict(H=480, W=640, model="dense-flow", cost="image_variance", sigma=0.0, T=0, ns=[0.5, 1.0, 2.0, 3.0]),
    "dense720p_smallflow": dict(H=720, W=1280, model="dense-flow", cost="image_variance", sigma=0.0, T=0, flow=4, ns=[1.0, 2.5]),
    "dense720p_hi": dict(H=720, W=1280, model="dense-flow", cost="image_variance", sigma=0.0, T=0, ns=[3.0, 4.0, 5.0, 6.0, 7.0]),
    "dense480p_hi": dict(H=480, W=640, model="dense-flow", cost="image_variance", sigma=0.0, T=0, ns=[3.0, 4.0, 5.0, 6.0, 7.0]),
    "voxel_hi": dict(H=260, W=346, model="dense-flow-voxel", cost="image_variance", sigma=1.0, T=10, ns=[3.0, 4.0, 5.0, 6.0, 7.0]),
    "voxel": dict(H=260, W=346, model="dense-flow-voxel", cost="image_variance", sigma=1.0, T=10, ns=[1.0, 1.25, 1.5, 1.75, 2.0, 2.25, 2.5, 3.0]),
}
which = sys.argv[1:] or list(shapes)
for name in which:
    c = shapes[name]
    H, W, T = c["H"], c["W"], c["T"]
    if T:
        f0 = E.utils.generate_smooth_flow((H, W), 20, seed=1046)
        motion = torch.from_numpy(np.stack([f0 * (1.0 + 0.02 * b) for b in range(T)]).astype(np.float32)).cuda()
    else:
        motion = torch.from_numpy(E.utils.generate_smooth_flow((H, W), c.get("flow", 20), seed=1046).astype(np.float32)).cuda()
    for nm in c["ns"]:
        n = int(nm * 1e6)
        ev = torch.from_numpy(E.utils.generate_events(n, H, W, 0.0, 0.05, seed=46)).cuda()
        h = E.CMaxHandle((H, W)).set_events(ev, time_bin=T)
        desc = E.make_descriptor(c["cost"], c["model"], sigma=c["sigma"], time_bin=T)
        call, _res, _grad = h.prepare(desc, motion)
        for _ in range(100):
            call()
        torch.cuda.synchronize()
        ts = []
        for _ in range(9):
            t0 = time.perf_counter()
            for _ in range(50):
                call()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 50 * 1e6)
        h.set_profiling(True, repeat=8)
        for _ in range(10):
            call()
        torch.cuda.synchronize()
        p = h.read_profile()
        h.set_profiling(False)
        k = {q: v[0] / max(v[1], 1) * 1e3 for q, v in p.items() if v[1]}
        print("%s n = %.2fM: %6.1f us per evaluation (K1 %5.1f, image %4.1f, K3 %5.1f)  %.2f us per M events" % (
            name, nm, float(np.median(ts)), k.get("vote", 0), k.get("stats", 0), k.get("grad", 0), float(np.median(ts)) / nm), flush=True)
        h.close()
        del ev
