"""The optimiser's objective for patch-based flow: x[2 * n_patch] -> loss, entirely on the GPU.

Equivalent of `PyramidalPatchContrastMaximization.objective_scipy`
(src/solver/patch_contrast_pyramid.py:430-462) + `motion_to_dense_flow` (464-516) for one scale:
  patch motion -> dense flow (cmax_patch_to_dense, patch_contrast_base.py:462-506) -> x t_scale
  [-> Burgers / upwind voxel (cmax_voxel_construct)] -> fused contrast objective (cmax_objective)
  [+ weight * total_variation(patch motion)].
Every stage is a HIP kernel with a hand-written adjoint; torch only chains them on the autograd tape.
"""
import ctypes
from typing import Dict, Optional, Tuple, Union

import numpy as np
import torch

from .. import _lib
from .. import functional as F
from ..cmax import CMaxHandle, ContrastObjective


class _SlicedFusedFn(torch.autograd.Function):
    """One fused term on a TIME-SLICED batch with torch.distributed collectives -- the fall-back of cmax_patch_plan_* under a
    communicator (e.g. gloo, or RCCL refused): this rank's votes -> all-reduce of the images (C1) -> contrast and the gradient of THIS
    RANK'S events.  backward hands autograd that SHARE of dL/dflow; PatchFlowObjective reduces the shares after the adjoint of the
    patch interpolation -- 2 n_patch numbers cross the fabric instead of the flow gradient, like in the library's own plan."""

    @staticmethod
    def forward(ctx, motion, sliced, desc):
        local = sliced.local
        images = local.objective_vote(desc, motion)
        sliced._all_reduce(images)
        result, grad = local.objective_finish(desc, motion, images, want_grad=motion.requires_grad)
        sliced._rank0_value(result)  # the term's scalars as rank 0 summed them, on every rank (TimeSlicedObjective._rank0_value)
        ctx.grad, ctx.mdtype = grad, motion.dtype
        return result[0].to(motion.dtype)

    @staticmethod
    def backward(ctx, gout):
        if ctx.grad is None:
            return None, None, None
        return ctx.grad.to(ctx.mdtype) * gout.to(ctx.mdtype), None, None


def patch_pad(patch_size, sliding_window, patch_shift=(0, 0)) -> Tuple[int, int]:
    """pad_h, pad_w of the reference's interpolation (patch_contrast_base.py:470-479)."""
    return tuple(int(patch_size[k] / 2 // sliding_window[k]) + patch_shift[k] // sliding_window[k] + 1 for k in range(2))


class PatchFlowObjective:
    def __init__(self, handle: CMaxHandle, t_scale: float, patch_image_size, patch_size, sliding_window,
                 patch_shift=(0, 0), cost: str = "hybrid", cost_with_weight: Optional[Dict[str, Union[float, str]]] = None,
                 blur_sigma: float = 1.0, time_aware: bool = False, time_bin: int = 10,
                 flow_interpolation: str = "burgers", t0_flow_location: str = "middle", filter_type: str = "bilinear", sliced=None):
        """sliced: a distributed.TimeSlicedObjective around `handle` when the batch is time-sliced over ranks.  With the library's own
        communicator on the handle (RCCL) the native plan evaluates the whole batch (cmax_patch_plan_* exchange images + 2 n_patch
        numbers); otherwise -- torch.distributed collectives -- the autograd-chained path below does the same two exchanges.  That
        fall-back has no exact Hessian-vector product (TorchWrapper then takes a difference quotient of the gradient, which on this
        piecewise-smooth objective is dominated by the kinks at cell borders): prefer a first-order method there, or the RCCL plan."""
        if filter_type != "bilinear":
            raise NotImplementedError("only the bilinear patch filter (the shipped configs) is built")
        self.handle = handle
        self.t_scale = float(t_scale)
        self.patch_image_size = (int(patch_image_size[0]), int(patch_image_size[1]))
        self.sliding_window = (int(sliding_window[0]), int(sliding_window[1]))
        self.pad = patch_pad(patch_size, sliding_window, patch_shift)
        self.time_aware = bool(time_aware)
        self.time_bin = int(time_bin)
        self.flow_interpolation = flow_interpolation
        self.t0_flow_location = t0_flow_location
        if self.time_aware and handle.time_bin != self.time_bin:
            handle.set_time_bins(self.time_bin)
        model = "dense-flow-voxel" if self.time_aware else "dense-flow"
        self.contrast = ContrastObjective(handle, model, cost=cost, cost_with_weight=cost_with_weight, sigma=blur_sigma)
        self.device = handle.device  # read by scipy_autograd.TorchWrapper
        self.auto_slabs = True  # ensure_time_slabs: time-slab order of the batch follows the motion's size
        self._plan = None
        self.sliced = sliced if (sliced is not None and sliced.world_size > 1 and not sliced.collectives.startswith("in-library")) else None
        if self.sliced is None:
            self._build_native_plan()

    # -- one-call native path (cmax_patch_plan_*, csrc/cmax_solver.hip) ---------------------------------
    def _build_native_plan(self):
        """The same chain as `__call__` + autograd, as ONE library call per evaluation.  Available for numeric
        hybrid weights ("inv" weights keep the autograd path)."""
        fused = [(w, desc) for _, w, desc in self.contrast.terms if desc is not None]
        tv = [w for name, w, desc in self.contrast.terms if desc is None]
        if not fused or len(fused) > 4 or any(w == "inv" for w, _ in fused) or any(w == "inv" for w in tv):
            return
        d = _lib.CmaxPatchObjective()
        d.n_terms = len(fused)
        d.time_aware = int(self.time_aware)
        d.T = self.time_bin if self.time_aware else 0
        d.scheme = F.SCHEME_CODES.get(self.flow_interpolation, -1) if self.time_aware else 0
        if self.time_aware and (d.scheme < 0 or self.t0_flow_location not in ("first", "middle")):
            return
        d.t0 = (self.time_bin // 2 if self.t0_flow_location == "middle" else 0) if self.time_aware else 0
        d.H, d.W = self.handle.image_size
        d.ph, d.pw = self.patch_image_size
        d.sw_h, d.sw_w = self.sliding_window
        d.pad_h, d.pad_w = self.pad
        d.tv_omit_boundary = int(self.contrast.omit_boundary)
        d.t_scale = self.t_scale
        for i, (w, desc) in enumerate(fused):
            d.weight[i] = float(w)
            d.term[i] = desc
        sign = 1.0 if self.contrast.direction == "minimize" else -1.0
        d.tv_weight = sign * float(sum(tv)) if tv else 0.0
        plan = ctypes.c_void_p()
        with torch.cuda.device(self.handle.device):
            _lib.check(_lib.load().cmax_patch_plan_create(self.handle._h, ctypes.byref(d), ctypes.byref(plan)))
        self._plan, self._nx = plan, 2 * self.patch_image_size[0] * self.patch_image_size[1]

    def __del__(self):
        plan, self._plan = getattr(self, "_plan", None), None
        if plan:
            try:
                _lib.load().cmax_patch_plan_destroy(plan)
            except Exception:  # interpreter shutdown
                pass

    def set_t_scale(self, t_scale: float):
        """The next batch behind the same handle has another duration (a solver keeps its objectives across frames)."""
        self.t_scale = float(t_scale)
        if self._plan is not None:
            _lib.check(_lib.load().cmax_patch_plan_set_t_scale(self._plan, self.t_scale))

    def native_plan_info(self):
        """(number of captured hipGraphs, graph replay enabled) of the native plan."""
        n, ok = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(_lib.load().cmax_patch_plan_info(self._plan, ctypes.byref(n), ctypes.byref(ok)))
        return n.value, bool(ok.value)

    @property
    def has_native_plan(self) -> bool:
        """TorchWrapper calls `value_and_grad_numpy` / `hvp_numpy` when this is True."""
        return self._plan is not None

    def ensure_time_slabs(self, x) -> int:
        """Large motions (round 5): before an evaluation at patch motion `x` the batch is put into the time-slab order its displacement
        over the batch asks for (max |x| * t_scale pixels; CMaxHandle.auto_time_slabs: thresholds with hysteresis, so a converging
        optimiser re-sorts at most a few times per scale).  The shipped configs draw their coarsest start from +-150 px per unit time
        (configs/*.yaml optimizer.parameters, initialize_random); un-slabbed, a 150-px evaluation of a 1M-event batch costs 4x a
        slabbed one (profiles/r04_large_motion.txt).  Time-aware objectives keep their time bins."""
        if self.time_aware or not self.auto_slabs:
            return 0
        xm = float(x.detach().abs().max()) if isinstance(x, torch.Tensor) else float(np.abs(x).max())
        return self.handle.auto_time_slabs(xm * abs(self.t_scale))

    def value_and_grad_numpy(self, x: np.ndarray, with_tv: bool = True, want_grad: bool = True):
        """x [2*ph*pw] float64 (host) -> (loss, gradient [2*ph*pw] float64): one cmax_patch_plan_evaluate call."""
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        if x.size != self._nx:
            raise ValueError(f"x has {x.size} elements, the patch grid needs {self._nx}")
        self.ensure_time_slabs(x)
        loss = ctypes.c_double(0.0)
        grad = np.empty(self._nx, dtype=np.float64) if want_grad else None
        with torch.cuda.device(self.handle.device):
            _lib.check(_lib.load().cmax_patch_plan_evaluate(self._plan, x.ctypes.data, int(with_tv), ctypes.byref(loss),
                                                            grad.ctypes.data if want_grad else None, F._stream()))
        return loss.value, grad

    def hvp_numpy(self, x: np.ndarray, v: np.ndarray, disp_step: float = 0.05, exact: bool = True) -> np.ndarray:
        """Hessian-vector product on host arrays through cmax_patch_plan_hvp: exact, what the reference's
        torch.autograd.functional.vhp returns -- for time-aware objectives including the second-order adjoint of the
        Burgers / upwind voxel chain (cmax_voxel_construct_tan / _adj_tan).  exact=False: central difference of the
        analytic gradient of the smooth part, as in `hvp` (kept for cross-checks)."""
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(-1)
        self.ensure_time_slabs(x)
        if self.time_aware and not exact:
            vmax = float(np.abs(v).max())
            if vmax == 0.0:
                return np.zeros_like(v)
            h = disp_step / (self.t_scale * vmax)
            gp = self.value_and_grad_numpy(x + h * v, with_tv=False)[1]
            gm = self.value_and_grad_numpy(x - h * v, with_tv=False)[1]
            return (gp - gm) / (2.0 * h)
        hv = np.empty(self._nx, dtype=np.float64)
        with torch.cuda.device(self.handle.device):
            _lib.check(_lib.load().cmax_patch_plan_hvp(self._plan, x.ctypes.data, v.ctypes.data, hv.ctypes.data, F._stream()))
        return hv

    def dense_flow(self, x: torch.Tensor) -> torch.Tensor:
        """[2*ph*pw] patch motion -> [2,H,W] (or [T,2,H,W]) flow in pixel per NORMALISED time."""
        motion = x.reshape((2,) + self.patch_image_size)
        dense = F.patch_to_dense(motion, self.handle.image_size, self.sliding_window, self.pad) * self.t_scale
        if self.time_aware:
            # construct(dense_flow * t_scale / scale) * scale / t_scale with scale = 1, then * t_scale
            # (patch_contrast_pyramid.py:452, 499-515): the voxel is built on the displacement field
            dense = F.construct_dense_flow_voxel(dense, self.time_bin, self.flow_interpolation, self.t0_flow_location)
        return dense

    @property
    def has_exact_hvp(self) -> bool:
        """TorchWrapper calls `hvp_numpy` / `hvp` when this is True.  Time-ignorant objectives: patch -> dense is linear,
        so H_x = t^2 P^T H_flow P.  Time-aware objectives add the second-order adjoint of the Burgers / upwind voxel
        chain (cmax_voxel_construct_tan / _adj_tan).  The total-variation term is piecewise linear: zero Hessian
        almost everywhere (a difference quotient of the whole objective would push its kinks into the curvature); with an
        "inv" weight it still contributes its rank-one part phi'' <grad TV, v> grad TV."""
        return self.contrast.has_exact_hvp and self.sliced is None  # (fall-back across ranks: difference quotient of the gradient)

    def _smooth_grad(self, x: torch.Tensor) -> torch.Tensor:
        """Gradient of the contrast terms (everything except total_variation) w.r.t. x."""
        from ..cmax import _FusedFn
        from ..costs.hybrid import combine

        xt = x.detach().clone().requires_grad_()
        loss = 0.0
        for name, weight, desc in self.contrast.terms:
            if desc is None:
                continue
            loss = loss + combine(weight, _FusedFn.apply(self.dense_flow(xt), self.handle, desc))
        (g,) = torch.autograd.grad(loss, xt)
        return g

    def _tv_inverse_curvature(self, x: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
        """Second-order part of an "inv"-weighted total_variation term: phi'' <grad TV, v> grad TV (H_TV = 0 a.e.)."""
        from ..costs.hybrid import combine_derivatives

        out = torch.zeros_like(x)
        for name, weight, desc in self.contrast.terms:
            if desc is not None or weight != "inv":
                continue
            xt = x.detach().clone().requires_grad_()
            tv = F.total_variation(xt.reshape((2,) + self.patch_image_size), self.contrast.omit_boundary)
            if self.contrast.direction != "minimize":
                tv = -tv
            (g,) = torch.autograd.grad(tv, xt)
            _, p2 = combine_derivatives("inv", tv.detach())
            out = out + p2 * (g * v).sum() * g
        return out

    def hvp(self, x: torch.Tensor, v: torch.Tensor, disp_step: float = 0.05) -> torch.Tensor:
        """Hessian-vector product w.r.t. the patch motion x on tensors (the autograd-chained path; TorchWrapper prefers
        `hvp_numpy`, which is exact for both kinds).  Time-ignorant: exact, composed from the autograd-wrapped stages
        ("inv" weights included).  Time-aware: a central difference of the analytic gradient of the smooth part
        (disp_step in pixels of displacement over the batch)."""
        x = x.to(self.handle.device)
        v = v.to(self.handle.device).to(x.dtype)
        tv_part = self._tv_inverse_curvature(x, v)
        if self.time_aware:
            # step measured in PIXELS OF DISPLACEMENT over the batch (x is pixel per time unit): a fixed
            # 0.05 px keeps the quotient above the fp32 noise of the gradient and below the pixel scale
            h = disp_step / (self.t_scale * float(v.abs().max()))
            return (self._smooth_grad(x + h * v) - self._smooth_grad(x - h * v)) / (2.0 * h) + tv_part
        shape = (2,) + self.patch_image_size
        size, sw, pad = self.handle.image_size, self.sliding_window, self.pad
        dense = F.patch_to_dense(x.detach().reshape(shape), size, sw, pad) * self.t_scale
        vdense = F.patch_to_dense(v.detach().reshape(shape).to(x.dtype), size, sw, pad) * self.t_scale
        hflow = self.contrast.hvp(dense, vdense).to(x.dtype)  # [2,H,W]
        # adjoint of the (linear) interpolation: the backward of patch_to_dense applied to hflow
        probe = x.detach().reshape(shape).clone().requires_grad_()
        out = F.patch_to_dense(probe, size, sw, pad)
        (hx,) = torch.autograd.grad(out, probe, grad_outputs=hflow.contiguous())
        return (hx * self.t_scale).reshape(x.shape) + tv_part

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        x = x.to(self.handle.device)
        self.ensure_time_slabs(x)
        if self.sliced is not None:
            return self._call_sliced(x)
        return self.contrast(self.dense_flow(x), x.reshape((2,) + self.patch_image_size))

    def _call_sliced(self, x: torch.Tensor) -> torch.Tensor:
        """The objective on a time-sliced batch with torch.distributed collectives: identical loss and gradient on every rank."""
        from ..costs.hybrid import combine

        xs = x.clone()  # the tensor the EVENT terms depend on: its gradient is this rank's share -> summed over the ranks
        if xs.requires_grad:
            xs.register_hook(lambda g: self.sliced._all_reduce(g.contiguous().clone()))
        dense = self.dense_flow(xs)
        loss = 0.0
        for name, weight, desc in self.contrast.terms:
            if desc is None:  # total variation of the patch grid: replicated on every rank, not a share
                value = F.total_variation(x.reshape((2,) + self.patch_image_size), self.contrast.omit_boundary)
                if self.contrast.direction != "minimize":
                    value = -value
            else:
                value = _SlicedFusedFn.apply(dense, self.sliced, desc)
            loss = loss + combine(weight, value)
        return loss
