#!/usr/bin/env python
"""Phase timeline of the event kernels K1 / K3 for one bench workload (library built with -DCMAX_TIMELINE, see csrc/cmax_fused.hip):
    CMAX_LIB=gpurun_variants/libtimeline.so python tools/timeline.py cfg2 [cfg4 cfg5 ...]
Thread 0 of every workgroup stamps the 100 MHz wall clock at its phase boundaries; printed per kernel: when workgroups start
(dispatch skew), how long each phase takes (median / p90 / max over the workgroups), when they end, and the kernel's span."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import bench  # noqa: E402
import event_based_optical_flow_amd as E  # noqa: E402
from event_based_optical_flow_amd import _lib  # noqa: E402

K1 = ["entry", "events+flows arrived, warped", "box reduced", "barrier 1 + window + LDS zero issued", "barrier 2 (+ offsets)", "votes done", "barrier 3", "flush issued"]
K3 = ["entry", "window staged + events warped", "barrier", "gather done", "barrier (accumulators)", "flush / scan done", "sums reduced", "-"]
# round 6: the fused image kernels between K1 and K3 (k_blur_stats_adj_var, k_stats_gimage_gm / k_blur_stats_gimage_gm), kernel index 2
KI = ["entry", "tile loads arrived", "barrier 1 (tile staged)", "clears issued + inner tile", "barrier 2", "outputs stored", "block sum (2 barriers)", "statistics' atomics issued"]


def run(name):
    cfg = bench.WORKLOADS[name]
    ev, motion, T = bench.make_inputs(cfg, 0, 1)
    dev = torch.device("cuda", 0)
    h = E.CMaxHandle((cfg["H"], cfg["W"]))
    h.set_events(torch.from_numpy(ev).to(dev) if not isinstance(ev, torch.Tensor) else ev, time_bin=T)
    if cfg["model"] == "dense-flow-voxel":
        f0 = torch.from_numpy(E.utils.generate_smooth_flow((cfg["H"], cfg["W"]), 20, seed=1046)).to(dev)
        m = E.utils.construct_dense_flow_voxel_torch(f0 / 20.0, T, "burgers", "middle").float() * 20.0
    else:
        m = torch.from_numpy(np.asarray(motion)).to(dev).float().contiguous()
    desc = E.make_descriptor(cfg["cost"], cfg["model"], sigma=cfg["sigma"], time_bin=T)
    call, res, grad = h.prepare(desc, m)
    for _ in range(50):
        call()
    torch.cuda.synchronize()
    buf = torch.zeros(3 * 4096 * 8, dtype=torch.int64, device=dev)
    lib = _lib.load()
    _lib.check(lib.cmax_debug_timeline(ctypes.c_void_p(buf.data_ptr())))
    call()
    torch.cuda.synchronize()
    _lib.check(lib.cmax_debug_timeline(None))
    t = buf.cpu().numpy().reshape(3, 4096, 8).astype(np.float64)
    info = h.work_list_info()
    print(f"== {name}: {cfg['desc']} -- {info['segments']} segments of <= {info['segment_events']} events")
    t00 = None
    for k, names in ((0, K1), (2, KI), (1, K3)):
        a = t[k]
        live = a[:, 0] > 0
        a = a[live]
        if not len(a):
            continue
        start = a[:, 0].min()
        if t00 is None:
            t00 = start
        last = np.max(np.where(a > 0, a, 0), axis=1)
        print(f" {('K1', 'K3', 'image kernel')[k]}: {len(a)} workgroups stamped; first starts at {(start - t00) / 100:.2f} us, starts spread over {(a[:, 0].max() - start) / 100:.2f} us, "
              f"last stamp at {(last.max() - t00) / 100:.2f} us (span {(last.max() - start) / 100:.2f} us)")
        st = np.sort((a[:, 0] - start) / 100.0)
        print(f"    starts: {int((st <= 1.0).sum())} within 1 us, {int((st <= 2.0).sum())} within 2 us, {int((st > 4.0).sum())} later than 4 us (latest {st[-1]:.2f} us)")
        prev = 0
        for i in range(1, 8):
            ok = a[:, i] > 0
            if not ok.any():
                continue
            d = (a[ok, i] - a[ok, prev]) / 100.0
            print(f"    {names[prev]:>40s} -> {names[i]:<40s} median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f} us   (reached by {ok.sum()})")
            prev = i
        tot = (last - a[:, 0]) / 100.0
        print(f"    workgroup lifetime (entry -> last stamp): median {np.median(tot):.2f}  p90 {np.percentile(tot, 90):.2f}  max {tot.max():.2f} us")
    h.close()


if __name__ == "__main__":
    for n in sys.argv[1:] or ["cfg2"]:
        run(n)
