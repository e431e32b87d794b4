"""Fused contrast-maximization objective: the hot path behind the reference's
`calculate_cost` (src/solver/patch_contrast_base.py:273-352).

`CMaxHandle` owns a `cmax_handle_t`: the batch is packed and sorted once (`set_events`), then every
objective evaluation runs warp+vote -> contrast -> gather-gradient on the GPU without
materialising warped events.  `ContrastObjective` exposes it as a differentiable callable
`loss = objective(motion)` for `scipy_autograd.minimize` / `torch.autograd.grad`.
"""
import copy
import ctypes
import logging
from typing import Dict, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib
from . import functional as F
from ._lib import CmaxObjective, check
from .array_types import to_device_tensor
from .costs.hybrid import combine

logger = logging.getLogger(__name__)

_COST_TABLE = {
    # name: (cost code, normalized, [(direction, multiplier), ...])  -- which IWEs get_arg_for_cost builds
    "image_variance": (_lib.COST_VARIANCE, 0, (("first", 1.0),)),
    "gradient_magnitude": (_lib.COST_GRADMAG, 0, (("first", 1.0),)),
    "normalized_image_variance": (_lib.COST_VARIANCE, 1, (("first", 1.0),)),
    "normalized_gradient_magnitude": (_lib.COST_GRADMAG, 1, (("first", 1.0),)),
    # forward = "last", backward = "first", middle x2 (multi_focal_*.py; patch_contrast_base.py:308-347)
    "multi_focal_normalized_image_variance": (_lib.COST_VARIANCE, 1, (("last", 1.0), ("first", 1.0), ("middle", 2.0))),
    "multi_focal_normalized_gradient_magnitude": (_lib.COST_GRADMAG, 1, (("last", 1.0), ("first", 1.0), ("middle", 2.0))),
}
FUSED_COSTS = tuple(_COST_TABLE)


def make_descriptor(cost: str, motion_model: str, direction: str = "minimize", sigma: float = 0.0,
                    omit_boundary: bool = True, normalize_t: bool = True, time_bin: int = 0,
                    warp_direction: Union[str, float] = "first") -> CmaxObjective:
    """Build the cmax_objective_t for one named cost (include/cmax_hip.h)."""
    if cost not in _COST_TABLE:
        raise KeyError(f"cost {cost!r} has no fused kernel; fused costs: {FUSED_COSTS}")
    if motion_model not in F.MODEL_CODES:
        raise KeyError(motion_model)
    if direction not in ("minimize", "maximize", "natural"):
        raise ValueError(f"direction should be minimize, maximize, and natural. Got {direction}.")
    code, normalized, refs = _COST_TABLE[cost]
    d = CmaxObjective()
    d.model = F.MODEL_CODES[motion_model]
    d.cost = code
    d.normalized = normalized
    d.minimize = int(direction == "minimize")
    # multi_focal_* return -loss for "maximize" (they sum iwe/orig ratios first)
    d.negate = int(len(refs) > 1 and direction == "maximize")
    d.omit_boundary = int(bool(omit_boundary))
    d.normalize_t = int(bool(normalize_t))
    d.n_ref = len(refs)
    for k, (ref_dir, mult) in enumerate(refs):
        if len(refs) == 1:
            ref_dir = warp_direction
        mode, frac = F.direction_to_ref(ref_dir)
        d.ref_mode[k] = mode
        d.ref_frac[k] = frac
        d.mult[k] = mult
    d.sigma = float(sigma)
    d.T = int(time_bin)
    return d


class CMaxHandle:
    """One GPU workspace for one event batch (cmax_create / cmax_set_events / cmax_objective)."""

    def __init__(self, image_size: Tuple[int, int], outer_padding: Union[int, Tuple[int, int]] = 0):
        _lib.require_gpu()
        self._lib = _lib.load()
        if isinstance(outer_padding, (int, float)):
            outer_padding = (int(outer_padding), int(outer_padding))
        self.image_size = (int(image_size[0]), int(image_size[1]))
        self.outer_padding = (int(outer_padding[0]), int(outer_padding[1]))
        self.padded_size = (self.image_size[0] + 2 * self.outer_padding[0], self.image_size[1] + 2 * self.outer_padding[1])
        self._h = ctypes.c_void_p()
        check(self._lib.cmax_create(self.image_size[0], self.image_size[1], self.outer_padding[0], self.outer_padding[1],
                                    ctypes.byref(self._h)))
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.time_bin = 0
        self.time_slabs = 0  # cmax_set_time_slabs order of the current batch (0: none); reset by set_events / set_time_bins

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            if getattr(self, "_comm_init_thread", None) is not None and self._comm_init_pending():
                return  # deliberately leaked: a timed-out cmax_comm_init still holds the native handle (comm_init)
            self._lib.cmax_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------
    finds_time_extremes = True  # set_events(ev, None, None, ...) reduces t_min / t_max on the device

    def set_keep_outside(self, on: bool = True):
        """Finite events whose source pixel lies OFF the sensor (cmax_set_keep_outside; takes effect with the next set_events).
        True -- THE DEFAULT since round 5 -- keeps them: the reference's 2-DoF warp has no bounds test on the source and lets such an
        event vote wherever it warps into the padded image (src/warp.py:506-515); a dense / voxel objective (a flow has no value off
        the sensor; the reference's gather indexes out of bounds there, src/warp.py:303-307) and the patch search then REFUSE the
        batch.  False drops them while packing (counted: batch_info()["dropped"], on_dropped) -- what a caller of a dense objective
        asks for explicitly; the pyramid solver does, with on_dropped="raise"."""
        check(self._lib.cmax_set_keep_outside(self._h, int(bool(on))))
        return self

    def set_events(self, events, tmin: Optional[float] = None, tmax: Optional[float] = None, time_bin: int = 0,
                   on_dropped: str = "warn"):
        """Pack + sort one [n,4] batch (numpy or tensor, fp32/fp64).  (tmin, tmax): global batch
        extremes when this handle only holds a time slice of the batch (multi-GPU).
        on_dropped: what to do when events were NOT packed because their source pixel lies off the sensor (or is NaN) --
        the fused path indexes source tiles and the flow field with it; the reference's 2-DoF warp would still let such an
        event vote if it warps into a padded image, its dense warp indexes out of bounds there -- "warn" (log),
        "raise" (a solver that must not diverge from the reference silently: ValueError) or "ignore"."""
        if on_dropped not in ("warn", "raise", "ignore"):
            raise ValueError(f"on_dropped should be warn, raise or ignore. Got {on_dropped}.")
        ev = to_device_tensor(events, "events")
        if ev.dim() != 2 or ev.shape[1] != 4:
            raise ValueError(f"events must be [n, 4], got {tuple(ev.shape)}")
        ev = ev.contiguous()
        have = tmin is not None and tmax is not None
        check(self._lib.cmax_set_events(self._h, ev.data_ptr(), F._code(ev), ev.shape[0], int(have),
                                        float(tmin) if have else 0.0, float(tmax) if have else 0.0, int(time_bin),
                                        F._stream()))
        self.time_bin = int(time_bin)
        self.time_slabs = 0
        info = self.batch_info() if on_dropped != "ignore" else {"dropped": 0, "outside": 0}
        dropped = info["dropped"]
        if info["outside"]:
            # said HERE, at the batch, not at the first dense evaluation behind the sort: keeping off-sensor events is the
            # default since round 5 and only the 2-DoF model is defined for them
            msg = (f"cmax_set_events kept {info['outside']} of {ev.shape[0]} events whose source pixel lies off the "
                   f"{self.image_size[0]} x {self.image_size[1]} sensor (the reference's 2-DoF warp has no bounds test, src/warp.py:506-515); "
                   "dense / voxel objectives and the patch search refuse such a batch -- set_keep_outside(False) before set_events drops them")
            if on_dropped == "raise":
                raise ValueError(msg)
            logger.warning(msg)
        if dropped and on_dropped == "raise":
            raise ValueError(f"cmax_set_events dropped {dropped} of {ev.shape[0]} events whose source pixel is outside the "
                             f"{self.image_size[0]} x {self.image_size[1]} sensor (or NaN); use the leaf operators (Warp + "
                             "EventImageConverter) for such batches, set_keep_outside() for a 2-DoF objective, or crop / pad the sensor")
        if dropped:
            logger.warning(f"cmax_set_events dropped {dropped} of {ev.shape[0]} events: source pixel outside the "
                           f"{self.image_size[0]} x {self.image_size[1]} sensor (or NaN); the fused path cannot keep them")
        return self

    def batch_info(self) -> Dict[str, int]:
        """{"packed", "dropped", "fractional", "owned_groups", "outside"} of the last set_events (cmax_batch_info, cmax_batch_outside)."""
        n, d = ctypes.c_int64(0), ctypes.c_int64(0)
        f, o = ctypes.c_int(0), ctypes.c_int(0)
        check(self._lib.cmax_batch_info(self._h, ctypes.byref(n), ctypes.byref(d), ctypes.byref(f), ctypes.byref(o)))
        out = ctypes.c_int64(0)
        check(self._lib.cmax_batch_outside(self._h, ctypes.byref(out)))
        return {"packed": n.value, "dropped": d.value, "fractional": bool(f.value), "owned_groups": bool(o.value), "outside": out.value}

    def packed_events(self):
        """(packed [n, 2] int64 on the host: word 0 = row | col << 12 | top byte << 24, word 1 = bits of the fp32 normalised time;
        group starts [n_groups + 1]) of the current batch in its sorted order (cmax_debug_packed_events) -- for tests of the order."""
        n = self.n_events
        ev = torch.empty((max(n, 1), 2), dtype=torch.int32, device=self.device)
        ng = ctypes.c_int(0)
        tiles = ((self.image_size[0] + 15) // 16) * ((self.image_size[1] + 15) // 16)
        gs = torch.empty(tiles * 256 + 1, dtype=torch.int32, device=self.device)
        check(self._lib.cmax_debug_packed_events(self._h, ev.data_ptr(), gs.data_ptr(), ctypes.byref(ng), F._stream()))
        return (ev[:n].cpu().numpy().astype(np.int64) & 0xFFFFFFFF), gs[: ng.value + 1].cpu().numpy()

    def work_list_info(self) -> Dict[str, int]:
        """{"segments", "segment_events", "small_accumulators"} of the work list the last set_events / set_time_bins cut
        (cmax_work_list_info): one workgroup of the event kernels per segment; segment_events is 2040 or 4088 (big segments)."""
        n, e, a = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        check(self._lib.cmax_work_list_info(self._h, ctypes.byref(n), ctypes.byref(e), ctypes.byref(a)))
        return {"segments": n.value, "segment_events": e.value, "small_accumulators": bool(a.value)}

    def set_time_bins(self, time_bin: int):
        check(self._lib.cmax_set_time_bins(self._h, int(time_bin), F._stream()))
        self.time_bin = int(time_bin)
        self.time_slabs = 0

    def set_time_slabs(self, n_slab: int):
        """Large motions of a 2-DoF / dense objective: order the batch in `n_slab` time slabs (cmax_set_time_slabs); <= 1 undoes it."""
        check(self._lib.cmax_set_time_slabs(self._h, int(n_slab), F._stream()))
        self.time_bin = 0
        self.time_slabs = int(n_slab) if n_slab > 1 else 0
        return self

    def auto_time_slabs(self, displacement_px: float) -> int:
        """Put the batch into the slab order `suggest_time_slabs(displacement_px)` asks for -- with hysteresis, so that an optimiser whose
        motion hovers around a threshold does not re-sort the batch every iteration: more slabs as soon as the displacement asks for
        them (an un-slabbed 150-px evaluation costs 4x a slabbed one), fewer only once it has fallen to 0.6 of the threshold that
        would keep the current count.  What the solver classes call before every evaluation (the motion is theirs, on the host; the
        library never reads a motion back).  Binned (voxel) handles are left alone.  Returns the slab count in force."""
        if self.time_bin > 0:
            return 0
        want = self.suggest_time_slabs(displacement_px)
        cur = max(self.time_slabs, 1)
        if want > cur or (want < cur and self.suggest_time_slabs(displacement_px / 0.6) < cur):
            self.set_time_slabs(want)
        return self.time_slabs

    def suggest_time_slabs(self, displacement_px: float) -> int:
        """Slab count for a motion of up to `displacement_px` over the batch (profiles/r04_large_motion.txt: 1M events @346x260 --
        none below ~25 px, 2 up to ~45 px, 4 beyond), capped so that a (tile, slab) group keeps >= ~600 events: smaller groups make
        segments of three groups too short to pay for a workgroup."""
        want = 1 if displacement_px < 25 else (2 if displacement_px < 45 else 4)
        tiles = ((self.image_size[0] + 15) // 16) * ((self.image_size[1] + 15) // 16)
        while want > 1 and self.n_events < 600 * tiles * want:
            want //= 2
        return want

    @property
    def n_events(self) -> int:
        n = ctypes.c_int64()
        check(self._lib.cmax_handle_info(self._h, ctypes.byref(n), None))
        return n.value

    def launch_floor_us(self, pairs: int = 200) -> float:
        """Microseconds per pair of dependent EMPTY launches with the grids of K1 and K3 on the current batch (cmax_debug_launch_floor),
        bracketed by events on the current stream: the floor of the headline evaluation's launch structure, measured in this run."""
        stream = F._stream()
        check(self._lib.cmax_debug_launch_floor(self._h, 20, stream))  # warm-up
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        check(self._lib.cmax_debug_launch_floor(self._h, int(pairs), stream))
        t1.record()
        t1.synchronize()
        return t0.elapsed_time(t1) * 1e3 / pairs

    @property
    def workspace_bytes(self) -> int:
        b = ctypes.c_int64()
        check(self._lib.cmax_handle_info(self._h, None, ctypes.byref(b)))
        return b.value

    def _motion32(self, motion) -> torch.Tensor:
        return to_device_tensor(motion, "motion").detach().to(torch.float32).contiguous()

    def _motion_arg(self, desc: CmaxObjective, motion):
        """(device tensor, descriptor) for one objective call.  A 2-DoF theta that arrives in fp64 -- the dtype the
        reference's solver optimises in (src/solver/patch_contrast_pyramid.py:186) -- is handed over as it is
        (cmax_objective_t::motion_dtype = CMAX_F64): the kernels round it to fp32 for the bulk arithmetic and keep the
        doubles for the events that lie on a cell border.  Everything else is fp32 on the device."""
        m = to_device_tensor(motion, "motion").detach()
        want = _lib.F64 if (desc.model == _lib.MODEL_2DOF and m.dtype == torch.float64) else _lib.F32
        m = m.contiguous() if want == _lib.F64 else m.to(torch.float32).contiguous()
        if desc.motion_dtype != want:
            desc = CmaxObjective.from_buffer_copy(desc)
            desc.motion_dtype = want
        return m, desc

    def iwe(self, motion, motion_model: Optional[str], direction: Union[str, float] = "first",
            normalize_t: bool = True, sigma: float = 0.0) -> torch.Tensor:
        """Image of warped events, fp32 [Hp, Wp].  motion_model None -> un-warped image (orig_iwe)."""
        out = torch.empty(self.padded_size, dtype=torch.float32, device=self.device)
        if motion_model is None:
            model, mptr, T = -1, None, 0
        else:
            m = self._motion32(motion)
            model, mptr = F.MODEL_CODES[motion_model], m.data_ptr()
            T = int(m.shape[0]) if model == _lib.MODEL_VOXEL else 0
        mode, frac = F.direction_to_ref(direction)
        check(self._lib.cmax_iwe(self._h, model, mptr, T, mode, frac, int(bool(normalize_t)), float(sigma),
                                 out.data_ptr(), F._stream()))
        return out

    def evaluate(self, desc: CmaxObjective, motion, want_grad: bool = True):
        """One cmax_objective call.  Returns (result double[8] on device, grad or None).
        result[0] = loss, result[1..n_ref] = raw contrasts, result[5] = contrast of orig_iwe."""
        m, desc = self._motion_arg(desc, motion)
        result = torch.empty(8, dtype=torch.float64, device=self.device)
        grad = None
        if want_grad:
            if desc.model == _lib.MODEL_2DOF:
                grad = torch.empty(2, dtype=torch.float64, device=self.device)
            else:
                grad = torch.empty(tuple(m.shape), dtype=torch.float32, device=self.device)
        check(self._lib.cmax_objective(self._h, ctypes.byref(desc), m.data_ptr(), result.data_ptr(),
                                       grad.data_ptr() if grad is not None else None, F._stream()))
        return result, grad

    def prepare(self, desc: CmaxObjective, motion, want_grad: bool = True, dist: bool = False):
        """A prepared cmax_objective (dist: cmax_objective_dist) call for an inner loop that evaluates the SAME motion
        buffer again and again (an optimiser updating it in place, a benchmark): outputs allocated once, pointers
        resolved once.  Returns (call, result, grad); `call()` enqueues one evaluation on the current stream and
        overwrites result / grad.  The call reads `call.motion`: that IS `motion` when it is already a contiguous device
        tensor of the dtype the kernels take (fp32; fp64 for a 2-DoF theta) -- otherwise it is a private converted copy, and an
        optimiser that updates its own tensor in place must write into `call.motion` instead (call.motion_is_callers tells).  `evaluate` spends ~6 us per call in Python (two allocations, tensor checks) -- as
        much as the three launches of a 1M-event 2-DoF evaluation leave the host to spare (profiles/r02_ablation.txt)."""
        m, desc = self._motion_arg(desc, motion)
        result = torch.empty(8, dtype=torch.float64, device=self.device)
        grad = None
        if want_grad:
            grad = (torch.empty(2, dtype=torch.float64, device=self.device) if desc.model == _lib.MODEL_2DOF
                    else torch.empty(tuple(m.shape), dtype=torch.float32, device=self.device))
        fn = self._lib.cmax_objective_dist if dist else self._lib.cmax_objective
        h, dref, mp, rp, gp, stream = self._h, ctypes.byref(desc), m.data_ptr(), result.data_ptr(), grad.data_ptr() if grad is not None else None, F._stream

        def call():
            rc = fn(h, dref, mp, rp, gp, stream())
            if rc:
                check(rc)

        call.keepalive = (m, desc, result, grad)  # the pointers above stay valid as long as the callable lives
        call.motion = m
        call.motion_is_callers = isinstance(motion, torch.Tensor) and motion.is_cuda and motion.data_ptr() == m.data_ptr()
        return call, result, grad

    # -- results on the host / raw sums (no finishing kernel) ---------------------------------------------------
    def evaluate_host(self, desc: CmaxObjective, motion, want_grad: bool = True):
        """One cmax_objective_host call: the evaluation with its results delivered to the host, the way an optimiser
        consumes them (the reference's wrapper ends in .cpu().numpy()).  Returns (result float64[8], grad ndarray or None);
        blocks.  For the 2-DoF image-variance objective the finishing kernel writes into pinned host memory and the call polls a
        run counter behind the results: no copy engine, no driver call on the way back."""
        m, desc = self._motion_arg(desc, motion)
        result = np.empty(8, dtype=np.float64)
        grad = None
        if want_grad:
            grad = np.empty(2, dtype=np.float64) if desc.model == _lib.MODEL_2DOF else np.empty(tuple(m.shape), dtype=np.float32)
        check(self._lib.cmax_objective_host(self._h, ctypes.byref(desc), m.data_ptr(), result.ctypes.data,
                                            grad.ctypes.data if grad is not None else None, F._stream()))
        return result, grad

    def prepare_host(self, desc: CmaxObjective, motion, want_grad: bool = True):
        """Prepared cmax_objective_host call: (call, result, grad) with `result` / `grad` numpy arrays that every `call()`
        overwrites (it returns when they are filled)."""
        m, desc = self._motion_arg(desc, motion)
        result = np.empty(8, dtype=np.float64)
        grad = None
        if want_grad:
            grad = np.empty(2, dtype=np.float64) if desc.model == _lib.MODEL_2DOF else np.empty(tuple(m.shape), dtype=np.float32)
        fn = self._lib.cmax_objective_host
        h, dref, mp, rp, gp, stream = self._h, ctypes.byref(desc), m.data_ptr(), result.ctypes.data, grad.ctypes.data if grad is not None else None, F._stream

        def call():
            rc = fn(h, dref, mp, rp, gp, stream())
            if rc:
                check(rc)

        call.keepalive = (m, desc, result, grad)
        call.motion = m
        return call, result, grad

    def has_raw(self, desc: CmaxObjective) -> bool:
        """Whether `desc` on the current batch has a raw form (cmax_objective_has_raw): every 2-DoF objective in the default
        (non-deterministic) mode on a non-empty batch, except a normalised plain variance."""
        return bool(self._lib.cmax_objective_has_raw(self._h, ctypes.byref(desc)))

    def prepare_raw(self, desc: CmaxObjective, motion):
        """Prepared cmax_objective_raw call: (call, raw, finalize).  `call()` enqueues K1 + K3 of one evaluation, which leave
        their partial sums in `raw` (device float64 [n_ref, RAW_DOUBLES]); `finalize()` copies `raw` to the host and returns
        (result float64[8], grad float64[2]) = cmax_finalize_raw_host -- the consumer's share of the evaluation, a few
        hundred additions."""
        m, desc = self._motion_arg(desc, motion)
        if not self.has_raw(desc):
            raise _lib.CmaxError(-1, "this objective has no raw form (2-DoF, default mode, non-empty batch; not a normalised plain variance)")
        raw = torch.empty((desc.n_ref, _lib.RAW_DOUBLES), dtype=torch.float64, device=self.device)
        host = torch.empty((desc.n_ref, _lib.RAW_DOUBLES), dtype=torch.float64).pin_memory()
        fn, fin = self._lib.cmax_objective_raw, self._lib.cmax_finalize_raw_host
        h, dref, mp, rp, stream = self._h, ctypes.byref(desc), m.data_ptr(), raw.data_ptr(), F._stream

        def call():
            rc = fn(h, dref, mp, rp, stream())
            if rc:
                check(rc)

        def finalize():
            host.copy_(raw, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            result, grad = np.empty(8, dtype=np.float64), np.empty(2, dtype=np.float64)
            check(fin(h, dref, host.data_ptr(), result.ctypes.data, grad.ctypes.data))
            return result, grad

        call.keepalive = (m, desc, raw, host)
        call.motion = m
        return call, raw, finalize

    def prepare_batch(self, desc: CmaxObjective, motions):
        """Prepared cmax_objective_batch call: (call, results, grads) for K candidate motions `motions` [K, ...] (2-DoF: [K, 2]; an
        fp64 theta crosses the ABI in fp64 like in `evaluate`).  results float64 [K, 8], grads float64 [K, 2] | float32 [K, ...].
        The reference's gradient-free paths evaluate batches of sampled motions (src/solver/base.py:738-758); for the 2-DoF
        image-variance objective the K evaluations share one launch of each kernel (blockIdx.z = candidate)."""
        mt = to_device_tensor(motions, "motions").detach()
        K = int(mt.shape[0])
        desc = CmaxObjective.from_buffer_copy(desc)
        if desc.model == _lib.MODEL_2DOF and mt.dtype == torch.float64:
            desc.motion_dtype = _lib.F64
            m = mt.contiguous()
        else:
            desc.motion_dtype = _lib.F32
            m = mt.to(torch.float32).contiguous()
        results = torch.empty((K, 8), dtype=torch.float64, device=self.device)
        if desc.model == _lib.MODEL_2DOF:
            grads = torch.empty((K, 2), dtype=torch.float64, device=self.device)
        else:
            grads = torch.empty((K,) + tuple(m.shape[1:]), dtype=torch.float32, device=self.device)
        fn, h, dref, mp, rp, gp, stream = self._lib.cmax_objective_batch, self._h, ctypes.byref(desc), m.data_ptr(), results.data_ptr(), grads.data_ptr(), F._stream

        def call():
            rc = fn(h, dref, mp, K, rp, gp, stream())
            if rc:
                check(rc)

        call.keepalive = (m, desc, results, grads)
        return call, results, grads

    def evaluate_batch(self, desc: CmaxObjective, motions):
        """(results [K, 8], grads [K, ...]) of K candidate motions: one cmax_objective_batch call."""
        call, results, grads = self.prepare_batch(desc, motions)
        call()
        return results, grads

    def hvp(self, desc: CmaxObjective, motion, tangent) -> torch.Tensor:
        """Exact Hessian-vector product H @ tangent of the objective w.r.t. the motion (cmax_objective_hvp):
        what torch.autograd.functional.vhp gives the reference's Newton-CG.  Returns fp64 [2] (2-DoF) or
        fp32 with the motion's shape."""
        m = self._motion32(motion)
        if desc.motion_dtype != _lib.F32:
            desc = CmaxObjective.from_buffer_copy(desc)
            desc.motion_dtype = _lib.F32
        u = to_device_tensor(tangent, "tangent").detach().to(torch.float64)
        umax = u.abs().max()
        if float(umax) == 0.0:
            return torch.zeros(2, dtype=torch.float64, device=self.device) if desc.model == _lib.MODEL_2DOF else torch.zeros_like(m)
        un = (u / umax).to(torch.float32).contiguous()  # unit max-norm: the derivative votes are fixed point
        if desc.model == _lib.MODEL_2DOF:
            hv = torch.empty(2, dtype=torch.float64, device=self.device)
        else:
            hv = torch.empty(tuple(m.shape), dtype=torch.float32, device=self.device)
        check(self._lib.cmax_objective_hvp(self._h, ctypes.byref(desc), m.data_ptr(), un.data_ptr(), hv.data_ptr(), F._stream()))
        return hv * umax.to(hv.dtype)

    # -- phase-split form (time-sliced multi-GPU, see distributed.py) --------------------------------
    def objective_vote(self, desc: CmaxObjective, motion) -> torch.Tensor:
        """Raw votes of this handle's events: fp32 [n_images, Hp, Wp] (n_ref images, plus the un-warped
        image when a normalised cost needs it).  To be all-reduced (sum) across time slices."""
        m, desc = self._motion_arg(desc, motion)
        images = torch.empty((5,) + self.padded_size, dtype=torch.float32, device=self.device)
        n_images = ctypes.c_int(0)
        check(self._lib.cmax_objective_vote(self._h, ctypes.byref(desc), m.data_ptr(), images.data_ptr(),
                                            ctypes.byref(n_images), F._stream()))
        return images[: n_images.value]

    def objective_finish(self, desc: CmaxObjective, motion, images: torch.Tensor, want_grad: bool = True):
        """Loss from the (globally reduced) images, gradient contribution of this handle's events."""
        m, desc = self._motion_arg(desc, motion)
        images = images.contiguous()
        result = torch.empty(8, dtype=torch.float64, device=self.device)
        grad = None
        if want_grad:
            if desc.model == _lib.MODEL_2DOF:
                grad = torch.empty(2, dtype=torch.float64, device=self.device)
            else:
                grad = torch.empty(tuple(m.shape), dtype=torch.float32, device=self.device)
        check(self._lib.cmax_objective_finish(self._h, ctypes.byref(desc), m.data_ptr(), images.data_ptr(),
                                              int(images.shape[0]), result.data_ptr(),
                                              grad.data_ptr() if grad is not None else None, F._stream()))
        return result, grad

    # -- deterministic mode ------------------------------------------------------------------------------------
    def set_deterministic(self, enable: bool = True):
        """Bit-identical IWE / loss / gradient from run to run (cmax_set_deterministic): integer accumulation wherever the
        order of events, workgroups or atomics would otherwise show in the last bits.  Slower; see include/cmax_hip.h."""
        check(self._lib.cmax_set_deterministic(self._h, int(bool(enable))))
        return self

    @property
    def deterministic(self) -> bool:
        v = ctypes.c_int(0)
        check(self._lib.cmax_get_deterministic(self._h, ctypes.byref(v)))
        return bool(v.value)

    # -- in-library collectives (RCCL over xGMI, see distributed.py) -------------------------------------
    def comm_available(self) -> Tuple[bool, str]:
        """(RCCL can be bound in this process, path of the shared object it was bound from) -- cmax_comm_available, a LOCAL
        call: ranks agree on it before any of them enters the blocking cmax_comm_init."""
        buf = ctypes.create_string_buffer(1024)
        rc = self._lib.cmax_comm_available(buf, len(buf))
        return rc == 0, buf.value.decode("utf-8", "replace")

    def comm_init(self, group=None, force_rccl: bool = False, timeout_s: float = 120.0):
        """Give this handle an RCCL communicator over the ranks of the torch.distributed `group` (default: the
        world): rank 0 draws the rendezvous id (cmax_comm_unique_id), torch.distributed ships it, every rank
        calls cmax_comm_init.  After that `evaluate_dist` needs no torch.distributed call at all.
        World size 1: no communicator (evaluate_dist == evaluate) unless force_rccl, which makes a real
        1-rank communicator so that the N > 1 enqueue sequence can be exercised on one GPU.
        No rank can be left waiting for another: (1) every rank first checks LOCALLY that RCCL can be bound and the
        ranks agree on that with one MIN all-reduce -- nobody enters ncclCommInitRank unless everybody will; (2) the blocking
        cmax_comm_init runs under a watchdog (`timeout_s`): a rank whose peers never arrive raises instead of hanging, and
        the caller (TimeSlicedObjective) then agrees on the torch.distributed fall-back."""
        import threading

        import torch.distributed as dist

        have = dist.is_available() and dist.is_initialized()
        world = dist.get_world_size(group) if have else 1
        rank = dist.get_rank(group) if have else 0
        if world == 1 and not force_rccl:
            check(self._lib.cmax_comm_init(self._h, None, 1, 0))
            return self
        ok, path = self.comm_available()
        err = None if ok else _lib.CmaxError(_lib.ENODEV, self._lib.cmax_last_error().decode("utf-8", "replace"))
        ids = [None]
        if rank == 0 and ok:
            buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
            try:
                check(self._lib.cmax_comm_unique_id(buf))
                ids = [buf.raw]
            except _lib.CmaxError as e:  # the other ranks wait in the exchange below: tell them before raising
                err, ok = e, False
        if world > 1:
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)  # (1) everybody, or nobody
            if int(flag.item()) == 0:
                raise err if err is not None else _lib.CmaxError(_lib.ENODEV, "RCCL is not available on another rank")
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(ids, src=src, group=group)
        elif err is not None:
            raise err
        if ids[0] is None:
            raise _lib.CmaxError(_lib.ECOMM, "rank 0 could not draw an RCCL rendezvous id")
        out = {}

        def init():  # (2) ncclCommInitRank blocks until all ranks have called it; ctypes releases the GIL meanwhile
            try:
                with torch.cuda.device(self.device):  # the current HIP device is per thread
                    out["rc"] = self._lib.cmax_comm_init(self._h, ids[0], world, rank)
                    out["msg"] = self._lib.cmax_last_error() if out["rc"] else b""
            except Exception as e:  # pragma: no cover
                out["exc"] = e

        t = threading.Thread(target=init, name="cmax_comm_init", daemon=True)
        t.start()
        t.join(timeout_s)
        if t.is_alive():
            # The thread is still inside ncclCommInitRank and holds the raw handle: the handle must outlive it (a
            # cmax_comm_destroy / cmax_destroy now, followed by the stuck call returning, is a use-after-free or a leaked
            # communicator).  comm_destroy / close wait for it when it has ended by then, and otherwise leave the native handle alone.
            self._comm_init_thread = t
            raise _lib.CmaxError(_lib.ECOMM, f"cmax_comm_init did not return within {timeout_s:.0f} s: a rank never reached ncclCommInitRank")
        if "exc" in out:
            raise out["exc"]
        if out.get("rc", 0):
            raise _lib.CmaxError(out["rc"], (out.get("msg") or b"").decode("utf-8", "replace"))
        self.rccl_path = path
        return self

    def comm_set_c2_bands(self, bands: int):
        """cmax_comm_set_c2_bands: dense objectives on owned groups all-reduce the gradient in `bands` row bands behind K3."""
        check(self._lib.cmax_comm_set_c2_bands(self._h, int(bands)))
        return self

    def comm_info(self) -> Tuple[int, int, int]:
        """(nranks, rank, RCCL version) of the handle's communicator; (1, 0, 0) without one."""
        n, r, v = ctypes.c_int(1), ctypes.c_int(0), ctypes.c_int(0)
        check(self._lib.cmax_comm_info(self._h, ctypes.byref(n), ctypes.byref(r), ctypes.byref(v)))
        return n.value, r.value, v.value

    def _comm_init_pending(self) -> bool:
        """True while a timed-out cmax_comm_init is still running on this handle (see comm_init)."""
        t = getattr(self, "_comm_init_thread", None)
        if t is None:
            return False
        t.join(0.0)
        if t.is_alive():
            return True
        self._comm_init_thread = None
        return False

    def comm_destroy(self):
        if self._comm_init_pending():
            raise _lib.CmaxError(-3, "a timed-out cmax_comm_init is still running on this handle")
        check(self._lib.cmax_comm_destroy(self._h))

    def comm_allreduce(self, t: torch.Tensor, op: str = "sum") -> torch.Tensor:
        """In-place all-reduce of a contiguous fp32 / fp64 device tensor on the handle's communicator."""
        if not (t.is_cuda and t.is_contiguous()):
            raise ValueError("comm_allreduce needs a contiguous device tensor")
        check(self._lib.cmax_comm_allreduce(self._h, t.data_ptr(), t.numel(), F._code(t), {"sum": 0, "min": 1, "max": 2}[op], F._stream()))
        return t

    def evaluate_dist(self, desc: CmaxObjective, motion, want_grad: bool = True):
        """One cmax_objective_dist call: the evaluation of the whole time-sliced batch, both all-reduces enqueued
        by the library between its kernels.  Same returns as `evaluate`, the same on every rank."""
        m, desc = self._motion_arg(desc, motion)
        result = torch.empty(8, dtype=torch.float64, device=self.device)
        grad = None
        if want_grad:
            if desc.model == _lib.MODEL_2DOF:
                grad = torch.empty(2, dtype=torch.float64, device=self.device)
            else:
                grad = torch.empty(tuple(m.shape), dtype=torch.float32, device=self.device)
        check(self._lib.cmax_objective_dist(self._h, ctypes.byref(desc), m.data_ptr(), result.data_ptr(),
                                            grad.data_ptr() if grad is not None else None, F._stream()))
        return result, grad

    # -- per-patch translation search (pyramid re-initialisation) ----------------------------------------
    def patch_search(self, boxes, patch_image_size: Tuple[int, int], candidates, sigma: float = 1.0):
        """Score translation candidates per patch with the reference's small-patch cost
        (calculate_cost_for_small_patch, src/solver/patch_contrast_pyramid.py:372-414) in one launch.
          boxes [n_patch, 4] = x_min, x_max, y_min, y_max (rows first, sensor coordinates; crop_event semantics)
          candidates [n_patch, n_cand, 2] pixel per unit of the raw timestamps
        -> (loss [n_patch, n_cand] = GM(un-warped) / GM(warped)  (NormalizedGradientMagnitude, "minimize"),
            gm [n_patch, n_cand + 1] raw gradient magnitudes (last column: un-warped), count [n_patch])."""
        b = torch.as_tensor(np.asarray(boxes), dtype=torch.int32).reshape(-1, 4).contiguous().to(self.device)
        c = torch.as_tensor(np.asarray(candidates) if not isinstance(candidates, torch.Tensor) else candidates)
        c = c.to(device=self.device, dtype=torch.float32).reshape(b.shape[0], -1, 2).contiguous()
        n_patch, n_cand = int(b.shape[0]), int(c.shape[1])
        gm = torch.empty((n_patch, n_cand + 1), dtype=torch.float32, device=self.device)
        count = torch.empty(n_patch, dtype=torch.int32, device=self.device)
        check(self._lib.cmax_patch_search(self._h, n_patch, b.data_ptr(), int(patch_image_size[0]), int(patch_image_size[1]),
                                          n_cand, c.data_ptr() if n_cand else None, float(sigma), gm.data_ptr(),
                                          count.data_ptr(), F._stream()))
        loss = gm[:, -1:].double() / gm[:, :-1].double()
        return loss, gm, count

    # -- per-kernel timing (bench.py roofline) ---------------------------------------------------------
    def set_profiling(self, enable, repeat: int = 1):
        """enable: bracket every hot launch with HIP events.  repeat > 1: issue each hot launch `repeat` times
        inside its bracket (amortises the bracket's dispatch latency; TIMING ONLY, results are meaningless)."""
        check(self._lib.cmax_set_profiling(self._h, (int(repeat) if repeat > 1 else 1) if enable else 0))

    def read_profile(self) -> Dict[str, Tuple[float, int]]:
        """{kernel class: (total ms, launches)} measured with HIP events on the launch stream: the four hot
        classes, "finish" (k_finish* / k_finalize) and "comm" (RCCL collectives)."""
        ms = (ctypes.c_double * _lib.PROF_CLASSES)()
        cnt = (ctypes.c_int64 * _lib.PROF_CLASSES)()
        check(self._lib.cmax_read_profile_all(self._h, ms, cnt))
        names = ("vote", "stats", "gimage", "grad", "finish", "comm")
        return {n: (ms[i], cnt[i]) for i, n in enumerate(names)}

    def last_iwe(self, k: int = 0) -> torch.Tensor:
        """Copy of the IWE of reference time k of the last evaluation (fp32 [Hp, Wp])."""
        out = torch.empty(self.padded_size, dtype=torch.float32, device=self.device)
        check(self._lib.cmax_copy_iwe(self._h, int(k), out.data_ptr(), F._stream()))
        return out


class _FusedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, motion, handle, desc):
        need = motion.requires_grad
        result, grad = handle.evaluate(desc, motion, want_grad=need)
        ctx.grad = grad
        ctx.mdtype = motion.dtype
        ctx.mdevice = motion.device
        return result[0].to(motion.dtype if motion.dtype.is_floating_point else torch.float64)

    @staticmethod
    def backward(ctx, gout):
        g = ctx.grad
        if g is None:
            return None, None, None
        g = g.to(ctx.mdtype) * gout.to(g.device).to(ctx.mdtype)
        return g.to(ctx.mdevice), None, None


class ContrastObjective:
    """loss = objective(motion[, coarse_flow]) with the reference's cost names.

    cost: any key of costs.functions, or "hybrid" with cost_with_weight={name: weight | "inv"}
    (src/costs/hybrid.py).  Event-based costs run as fused kernels on `handle`; total_variation runs
    on the patch flow handed in as `coarse_flow` (patch_contrast_base.py:349-350)."""

    def __init__(self, handle: CMaxHandle, motion_model: str, cost: str = "image_variance",
                 cost_with_weight: Optional[Dict[str, Union[float, str]]] = None, direction: str = "minimize",
                 sigma: float = 0.0, omit_boundary: bool = True, normalize_t: bool = True,
                 warp_direction: Union[str, float] = "first"):
        self.handle = handle
        self.motion_model = motion_model
        self.direction = direction
        self.omit_boundary = omit_boundary
        terms = cost_with_weight if cost == "hybrid" else {cost: 1.0}
        if cost == "hybrid" and not cost_with_weight:
            raise ValueError("hybrid cost needs cost_with_weight")
        self.terms = []
        for name, weight in terms.items():
            if name == "total_variation":
                self.terms.append((name, weight, None))
            else:
                desc = make_descriptor(name, motion_model, direction, sigma, omit_boundary, normalize_t,
                                       handle.time_bin if F.MODEL_CODES[motion_model] == _lib.MODEL_VOXEL else 0,
                                       warp_direction)
                self.terms.append((name, weight, desc))

    has_exact_hvp = True  # every fused term, numeric and "inv" weights (total_variation does not depend on `motion`)

    def hvp(self, motion: torch.Tensor, vector: torch.Tensor) -> torch.Tensor:
        """Exact Hessian-vector product w.r.t. `motion` (same shape/dtype as motion).  A term of weight w contributes
        w * H_c v (cmax_objective_hvp); an "inv" term phi(c) = 1 / c contributes phi' H_c v + phi'' <grad c, v> grad c
        with c and grad c from one cmax_objective call (src/costs/hybrid.py:51-53 differentiated twice)."""
        from .costs.hybrid import combine_derivatives

        out = None
        v64 = to_device_tensor(vector, "vector").detach().to(torch.float64)
        for name, weight, desc in self.terms:
            if desc is None:
                continue  # total_variation acts on the coarse flow, not on `motion`
            hv = self.handle.hvp(desc, motion, vector).to(torch.float64)
            if weight == "inv":
                res, grad = self.handle.evaluate(desc, motion, want_grad=True)
                p1, p2 = combine_derivatives("inv", res[0])
                g64 = grad.to(torch.float64).reshape(v64.shape)
                hv = p1 * hv + p2 * (g64 * v64).sum() * g64.reshape(hv.shape)
            else:
                hv = hv * float(weight)
            out = hv if out is None else out + hv
        if out is None:
            out = torch.zeros_like(vector, dtype=torch.float64)
        return out.reshape(vector.shape).to(vector.dtype if vector.dtype.is_floating_point else torch.float64)

    def evaluate_candidates(self, motions, coarse_flows=None) -> torch.Tensor:
        """Loss of K candidate motions `motions` [K, ...] -> float64 [K] on the device: one cmax_objective_batch call per fused term
        (the reference's gradient-free paths score batches of sampled motions, src/solver/base.py:738-758,
        src/solver/patch_contrast_base.py:126-187).  total_variation members act on `coarse_flows` [K, 2, ph, pw]; for a 2-DoF
        translation the reference hands them a [2, 1, 1] flow, whose total variation is 0: they may be omitted there."""
        mt = to_device_tensor(motions, "motions").detach()
        K = int(mt.shape[0])
        loss = torch.zeros(K, dtype=torch.float64, device=self.handle.device)
        for name, weight, desc in self.terms:
            if desc is None:
                if coarse_flows is None:
                    if F.MODEL_CODES[self.motion_model] == _lib.MODEL_2DOF:
                        continue
                    raise KeyError("flow")
                cf = to_device_tensor(coarse_flows, "flow")
                value = torch.stack([F.total_variation(cf[k], self.omit_boundary) for k in range(K)]).to(torch.float64)
                if self.direction != "minimize":
                    value = -value
            else:
                results, _ = self.handle.evaluate_batch(desc, mt)
                value = results[:, 0]
            loss = loss + combine(weight, value)
        return loss

    def __call__(self, motion: torch.Tensor, coarse_flow: Optional[torch.Tensor] = None) -> torch.Tensor:
        loss = 0.0
        for name, weight, desc in self.terms:
            if desc is None:
                if coarse_flow is None:
                    raise KeyError("flow")
                value = F.total_variation(to_device_tensor(coarse_flow, "flow"), self.omit_boundary)
                if self.direction != "minimize":
                    value = -value
            else:
                value = _FusedFn.apply(motion, self.handle, desc)
            value = value.to(motion.device) if isinstance(motion, torch.Tensor) else value
            loss = loss + combine(weight, value)
        return loss
