#!/usr/bin/env python
"""Large displacements (cfg2 shape: 1M uniform events, 260x346, 2-DoF, image variance): one evaluation in its raw form (K1 + K3)
for growing |theta| -- when a segment's LDS window (source tiles + displacement range) stops fitting, the event kernels fall back to a
window clipped to the centre of the box, test every vote and send the overflow to global atomics (workgroup-uniform slow path).
us per evaluation (median of 15 windows of 100 evaluations) | K1 | K3 (8-launch HIP-event brackets)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import event_based_optical_flow_amd as E

H, W, n = 260, 346, 1_000_000
ev = torch.from_numpy(E.utils.generate_events(n, H, W, 0.0, 0.05, seed=46)).cuda()
for theta in ((12.3, -7.7), (30.0, -20.0), (50.0, -40.0), (80.0, -60.0), (150.0, -100.0)):
    row = []
    for tb, slabs in ((0, 0), (8, 0), (0, 2), (0, 4), (0, 8)):
        h = E.CMaxHandle((H, W)).set_events(ev, time_bin=tb)
        if slabs:
            h.set_time_slabs(slabs)
        desc = E.make_descriptor("image_variance", "2d-translation")
        m = torch.tensor(theta, dtype=torch.float32, device="cuda")
        call, raw, fin = h.prepare_raw(desc, m)
        for _ in range(300):
            call()
        torch.cuda.synchronize()
        ts = []
        for _ in range(15):
            t0 = time.perf_counter()
            for _ in range(100):
                call()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 100 * 1e6)
        h.set_profiling(True, repeat=8)
        for _ in range(20):
            call()
        torch.cuda.synchronize()
        p = h.read_profile()
        h.set_profiling(False)
        k = {q: v[0] / max(v[1], 1) * 1e3 for q, v in p.items() if v[1]}
        res, grad = h.evaluate(desc, m)
        row.append("%6.1f us (K1 %5.1f K3 %5.1f) segs %4d loss %.9g g0 %.7g" % (float(np.median(ts)), k.get("vote", 0), k.get("grad", 0), h.work_list_info()["segments"], float(res[0]), float(grad[0])))
        h.close()
    print("theta = (%6.1f, %6.1f) px per batch:" % theta)
    for name, r in zip(("un-binned ", "time_bin 8", "slabs 2   ", "slabs 4   ", "slabs 8   "), row):
        print("    %s %s" % (name, r), flush=True)
