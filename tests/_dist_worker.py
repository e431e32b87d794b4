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


def patch_main():
    """CMAX_DIST_CASE=patch: the SOLVER's objective (PatchFlowObjective) on a time-sliced batch through its torch.distributed fall-back
    (gloo: two ranks cannot share a device under RCCL) with UNEQUAL slices and with an EMPTY one -- loss, gradient and the
    difference-quotient Hessian-vector product must be the same on every rank and equal the single-handle native plan's."""
    from event_based_optical_flow_amd.solver import PatchFlowObjective
    from event_based_optical_flow_amd.solver.scipy_autograd import TorchWrapper

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    size, n = (96, 128), 90_000
    ev = E.utils.generate_structured_events(n, size[0], size[1], (7.0, -5.0), n_dots=300, seed=23)
    t_scale = float(ev[:, 2].max() - ev[:, 2].min())
    pis, ps = (3, 4), (32, 32)
    rng = np.random.default_rng(9)
    x = (np.array([[-7.0], [5.0]]) * (1.0 + 0.2 * rng.uniform(-1, 1, (2, 12)))).reshape(-1) / t_scale
    v = rng.normal(0.0, 1.0, x.size)
    hybrid = {"multi_focal_normalized_gradient_magnitude": 1.0, "total_variation": 0.01}
    out = []
    for tag in ("unequal", "empty"):
        cuts = [0] + [int(n * (0.7 if tag == "unequal" else 1.0) * (r + 1) / (world - 1)) for r in range(world - 1)] + [n]
        cuts = [min(c, n) for c in cuts]
        lo, hi = cuts[rank], cuts[rank + 1]
        for cost, cww, sigma in (("image_variance", None, 0.0), ("hybrid", hybrid, 1.0)):
            handle = E.CMaxHandle(size)
            sliced = TimeSlicedObjective(handle, in_library=False)
            sliced.set_local_events(torch.from_numpy(ev[lo:hi]).cuda(), device="cpu")
            obj = PatchFlowObjective(handle, t_scale, pis, ps, ps, (0, 0), cost=cost, cost_with_weight=cww, blur_sigma=sigma, sliced=sliced)
            assert not obj.has_native_plan and not obj.has_exact_hvp and handle.n_events == hi - lo
            w = TorchWrapper(obj, precision="float64", device="cuda", hvp_eps=0.02)  # (difference quotient: a step of ~0.1 px of displacement)
            w.get_input(x)
            for _ in range(2):
                loss, grad = w.get_value_and_grad(x)
            hv = w.get_hvp(x, v)
            mine = torch.from_numpy(np.concatenate([[float(loss)], grad, hv]))
            gathered = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(gathered, mine)
            spread = max(float((g - gathered[0]).abs().max()) for g in gathered)
            if rank == 0:
                h1 = E.CMaxHandle(size).set_events(ev)
                obj1 = PatchFlowObjective(h1, t_scale, pis, ps, ps, (0, 0), cost=cost, cost_with_weight=cww, blur_sigma=sigma)
                loss1, grad1 = obj1.value_and_grad_numpy(x)
                hv1 = obj1.hvp_numpy(x, v)
                out.append({"slices": tag, "cost": cost, "slice": [lo, hi], "loss": float(loss), "loss_single": float(loss1),
                            "grad_rel_diff": float(np.abs(grad - grad1).max() / np.abs(grad1).max()),
                            "hvp_cosine": float(np.dot(hv, hv1) / (np.linalg.norm(hv) * np.linalg.norm(hv1))), "spread_over_ranks": spread})
                h1.close()
            handle.close()
    if rank == 0:
        print(json.dumps({"world": world, "cases": out}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    patch_main() if os.environ.get("CMAX_DIST_CASE") == "patch" else main()
