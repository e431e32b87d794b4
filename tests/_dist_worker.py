"""Rank program of tests/test_gpu_dist.py::test_eight_ranks_sharing_one_gpu (run under torch.distributed.run).

Every rank builds the SAME seeded batch, keeps its contiguous time slice (one rank's slice is empty in the "empty" cases),
and evaluates the time-sliced objective with the other ranks: global extremes by one MIN all-reduce, C1 (images) and C2
(gradient) as torch.distributed all-reduces around the phase-split calls -- RCCL refuses several ranks on one device, so the
transport is gloo; the sharding, both exchange steps and the rank-local HIP kernels are the ones of an N-GPU run.  Rank 0
also evaluates the whole batch behind ONE handle and prints one JSON line with the differences."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import event_based_optical_flow_amd as E  # noqa: E402
from event_based_optical_flow_amd.distributed import TimeSlicedObjective, time_slice_bounds  # noqa: E402

CASES = [
    # model, cost, sigma, time bins, one rank without events
    ("2d-translation", "image_variance", 0.0, 0, False),
    ("2d-translation", "multi_focal_normalized_gradient_magnitude", 1.0, 0, True),
    ("dense-flow", "gradient_magnitude", 0.0, 0, False),
    ("dense-flow", "image_variance", 0.0, 0, True),
    ("dense-flow-voxel", "image_variance", 1.0, 5, False),
]


def motion_of(model, size, Tn):
    if model == "2d-translation":
        return np.array([9.0, -6.0])
    if model == "dense-flow":
        return E.utils.generate_smooth_flow(size, 10, seed=5)
    return np.stack([E.utils.generate_smooth_flow(size, 10, seed=5 + b) for b in range(Tn)])


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    size, n = (96, 128), 160_000
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=19)
    out = []
    for model, cost, sigma, Tn, empty in CASES:
        motion = motion_of(model, size, Tn)
        desc = E.make_descriptor(cost, model, sigma=sigma, time_bin=Tn)
        if empty:  # the last rank holds nothing: the others share the batch
            lo, hi = time_slice_bounds(n, world - 1, rank) if rank < world - 1 else (n, n)
        else:
            lo, hi = time_slice_bounds(n, world, rank)
        obj = TimeSlicedObjective(E.CMaxHandle(size), in_library=False)
        obj.set_local_events(torch.from_numpy(ev[lo:hi]).cuda(), time_bin=Tn, device="cpu")
        assert obj.local.n_events == hi - lo
        for _ in range(2):  # double-buffered images, cached un-warped image: twice
            res, grad = obj.evaluate(desc, motion)
        torch.cuda.synchronize()
        # every rank must hold the same loss and the same (reduced) gradient
        both = torch.cat([res[:1].double().cpu(), grad.double().cpu().reshape(-1)[:64]])
        gathered = [torch.empty_like(both) for _ in range(world)]
        dist.all_gather(gathered, both)
        spread = max(float((g[1:] - gathered[0][1:]).abs().max()) for g in gathered)  # the all-reduced gradient: identical bits
        loss_spread = max(float((g[0] - gathered[0][0]).abs()) for g in gathered)  # fp64 atomics in another order on every rank
        if rank == 0:
            h1 = E.CMaxHandle(size).set_events(ev, time_bin=Tn)
            res1, grad1 = h1.evaluate(desc, motion)
            g, g1 = grad.double().cpu().numpy(), grad1.double().cpu().numpy()
            out.append({"model": model, "cost": cost, "empty": empty, "loss": float(res[0]), "loss_single": float(res1[0]),
                        "grad_rel_diff": float(np.abs(g - g1).max() / np.abs(g1).max()), "spread_over_ranks": spread, "loss_spread_over_ranks": loss_spread,
                        "slice": [lo, hi]})
        obj.local.close()
    if rank == 0:
        print(json.dumps({"world": world, "cases": out}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
