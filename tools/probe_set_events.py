#!/usr/bin/env python
"""Where cmax_set_events spends its time (per batch): host wall time until the call returns and until the stream is idle.
    python tools/probe_set_events.py [HxW:n ...]        default: 260x346:1M 480x640:5M 720x1280:20M 720x1280:64M
Events are drawn on the device (sorted uniform timestamps, integer pixels); CMAX_NO_RUN_SORT=1 in the environment drops the ordering
of the pixel runs by time.  Under rocprofv3 --kernel-trace (tools/prof_set_events.sh) the per-kernel durations of the pipeline follow."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import event_based_optical_flow_amd as E
from event_based_optical_flow_amd import _lib, functional as F

lib = _lib.load()
cases = [a for a in sys.argv[1:] if ":" in a] or ["260x346:1000000", "480x640:5000000", "720x1280:20000000", "720x1280:64000000"]
reps = 12
for c in cases:
    hw, n = c.split(":")
    H, W = (int(v) for v in hw.split("x"))
    n = int(float(n))
    g = torch.Generator(device="cuda")
    g.manual_seed(46)
    ev = torch.empty((n, 4), dtype=torch.float64, device="cuda")
    ev[:, 0] = torch.randint(0, H, (n,), generator=g, device="cuda")
    ev[:, 1] = torch.randint(0, W, (n,), generator=g, device="cuda")
    ev[:, 2] = torch.sort(torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * 0.05).values
    ev[:, 3] = 1.0
    h = E.CMaxHandle((H, W))
    h.set_events(ev)
    torch.cuda.synchronize()
    ret, idle = [], []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = lib.cmax_set_events(h._h, ev.data_ptr(), 1, n, 0, 0.0, 0.0, 0, F._stream())
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        assert rc == 0
        ret.append((t1 - t0) * 1e6)
        idle.append((t2 - t0) * 1e6)
    print("%dx%d %d events: cmax_set_events returns after %.1f us (median), stream idle after %.1f us; work list %s" % (
        H, W, n, np.median(ret), np.median(idle), h.work_list_info()), flush=True)
    h.close()
    del ev
