"""Time-sliced multi-GPU objective: one process per GPU (torch.distributed, backend "nccl" = RCCL
over xGMI), each rank owns a contiguous time slice of the event batch.

The reference has no distributed code at all; this is the natural sharding
of the path: the IWE is a sum over events and the gradient is a sum over events given the global
dL/dIWE.  Per objective evaluation there are exactly two exchange steps:

    C1  all-reduce(sum) of the raw vote images   [n_images, Hp, Wp] fp32   (one grouped call)
    C2  all-reduce(sum) of the gradient          double[2] | fp32 [2,H,W] | fp32 [T,2,H,W]

Between them every rank redundantly evaluates the (image-space, microseconds) contrast on the
reduced image; its scalars (result[8]: loss, statistics) leave the evaluation as RANK 0'S BITS on every rank -- they ride in
C2 (one grouped RCCL call; x + 0 + ... + 0 is exact), so replicated optimisers take identical decisions.  Both collectives are enqueued BY THE LIBRARY, on the stream
of its kernels (cmax_comm_init + cmax_objective_dist: RCCL bound inside libcmax_hip.so): one ctypes
call per evaluation, no Python between the phases.  torch.distributed only ships the RCCL rendezvous
id once; if the library cannot bring up its communicator (no librccl), or for a `local` without one
(the CPU test's stand-in), the same two exchange steps run as torch.distributed all-reduces around
the phase-split calls.  t_min / t_max are agreed once per batch with one MIN
all-reduce over (t_min, -t_max) because dt normalisation and the voxel bin edges are defined on the whole batch
(src/warp.py:216-224, 254-259, 342-345).

The per-rank compute is injected (`local`): production uses CMaxHandle (HIP); the CPU test
(tests/test_distributed_gloo.py) injects a checker-backed stand-in to exercise the sharding and the
collectives under gloo.
"""
from typing import Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def time_slice_bounds(n: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous split of n time-sorted events into world_size slices (sizes differ by <= 1)."""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def agree_time_extremes(t_local_min: float, t_local_max: float, group=None, device="cpu") -> Tuple[float, float]:
    """Global (t_min, t_max) of the batch from per-slice extremes: ONE all-reduce, MIN over (t_min, -t_max)."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return float(t_local_min), float(t_local_max)
    pair = torch.tensor([t_local_min, -t_local_max], dtype=torch.float64, device=device)
    dist.all_reduce(pair, op=dist.ReduceOp.MIN, group=group)
    lo, neg_hi = pair.tolist()
    return float(lo), float(-neg_hi)


class TimeSlicedObjective:
    """Objective evaluation over a batch sharded by time slice across the ranks of `group`.

    local: object with
        set_events(events, tmin, tmax, time_bin)
        objective_vote(desc, motion)            -> images [n_images, Hp, Wp]
        objective_finish(desc, motion, images, want_grad) -> (result[8], grad)
    (CMaxHandle implements it.)"""

    def __init__(self, local, group=None, in_library: bool = True):
        self.local = local
        self.group = group
        self.collectives = "none" if self.world_size == 1 else "torch.distributed"
        if in_library and self.world_size > 1 and hasattr(local, "comm_init") and dist.get_backend(group) == "nccl":
            # every rank must take the same branch: agree on success before trusting the communicator
            ok = 1
            try:
                local.comm_init(group)
            except Exception as e:  # RCCL not loadable / communicator refused: the torch path is the same RCCL underneath
                import warnings

                warnings.warn(f"in-library RCCL communicator unavailable ({e}); using torch.distributed all-reduces")
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=getattr(local, "device", "cpu"))
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            if int(flag.item()) == 1:
                self.collectives = "in-library RCCL (cmax_objective_dist)"
            elif ok:
                local.comm_destroy()

    @property
    def world_size(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def set_local_events(self, events_slice, time_bin: int = 0, device="cpu"):
        """`events_slice`: this rank's contiguous time slice [n_local, 4] (may be empty)."""
        ev = events_slice
        if self.world_size == 1 and getattr(self.local, "finds_time_extremes", False):
            # the whole batch is here: the handle reduces t_min / t_max on the device, no host round trip
            self.local.set_events(ev, None, None, time_bin)
            return None, None
        if len(ev) > 0:
            t = ev[:, 2]
            lo, hi = float(t.min()), float(t.max())
        else:
            lo, hi = float("inf"), float("-inf")
        tmin, tmax = agree_time_extremes(lo, hi, self.group, device)
        self.local.set_events(ev, tmin, tmax, time_bin)
        return tmin, tmax

    def _all_reduce(self, t: torch.Tensor):
        if self.world_size > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def _rank0_value(self, t: torch.Tensor):
        """Scalars every rank computed redundantly on the reduced image (loss, statistics: fp64 sums in each rank's own order, equal
        to ~1e-16, not bit for bit) become RANK 0'S on every rank.  Replicated optimisers -- every rank runs the same SciPy loop
        (src/solver/scipy_autograd/scipy_minimize.py:100-117) -- must see identical values: one differing line-search comparison means
        different call sequences and mismatched collectives.  (The library's own path lets result[8] ride in the gradient's all-reduce,
        cmax_objective_dist; this fall-back spends a broadcast.)"""
        if self.world_size > 1:
            src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
            dist.broadcast(t, src=src, group=self.group)
        return t

    def prepare(self, desc, motion, want_grad: bool = True):
        """(call, result, grad): `call()` = one evaluation of the whole batch with preallocated outputs (CMaxHandle.prepare);
        the torch-collectives fallback and stand-in locals get a closure over `evaluate` that copies into the same buffers."""
        if hasattr(self.local, "prepare") and (self.world_size == 1 or self.collectives.startswith("in-library")):
            return self.local.prepare(desc, motion, want_grad, dist=self.world_size > 1)
        result, grad = self.evaluate(desc, motion, want_grad)

        def call():
            r, g = self.evaluate(desc, motion, want_grad)
            result.copy_(r)
            if g is not None:
                grad.copy_(g)

        return call, result, grad

    def evaluate(self, desc, motion, want_grad: bool = True):
        if self.world_size == 1 and hasattr(self.local, "evaluate"):
            return self.local.evaluate(desc, motion, want_grad)  # no exchange step: one cmax_objective call
        if self.collectives.startswith("in-library"):
            return self.local.evaluate_dist(desc, motion, want_grad)  # vote, C1, finish, C2 enqueued by one library call
        images = self.local.objective_vote(desc, motion)
        self._all_reduce(images)  # C1
        result, grad = self.local.objective_finish(desc, motion, images, want_grad)
        if grad is not None:
            self._all_reduce(grad)  # C2
        self._rank0_value(result)  # rank-consistent scalars
        return result, grad
