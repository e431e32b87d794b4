#!/usr/bin/env python
"""Where cmax_set_events spends its time (per batch): host wall time until the call returns, until the stream is idle, for
1M / 5M events; CMAX_NO_RUN_SORT=1 in the environment drops the ordering of the pixel runs by time."""
import ctypes
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import event_based_optical_flow_amd as E
from event_based_optical_flow_amd import _lib, functional as F

lib = _lib.load()
for H, W, n in ((260, 346, 1_000_000), (480, 640, 5_000_000)):
    ev = torch.from_numpy(E.utils.generate_events(n, H, W, 0.0, 0.05, seed=46)).cuda()
    h = E.CMaxHandle((H, W))
    h.set_events(ev)
    torch.cuda.synchronize()
    ret, idle = [], []
    for _ in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = lib.cmax_set_events(h._h, ev.data_ptr(), 1, n, 0, 0.0, 0.0, 0, F._stream())
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        assert rc == 0
        ret.append((t1 - t0) * 1e6)
        idle.append((t2 - t0) * 1e6)
    print("%dx%d %d events: cmax_set_events returns after %.1f us (median), stream idle after %.1f us" % (H, W, n, np.median(ret), np.median(idle)), flush=True)
