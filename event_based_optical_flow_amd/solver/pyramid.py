"""Coarse-to-fine patch solver on the GPU objective (SURVEY.md section 8f rank 4 -- a "next" row).

Same constructor signature, config keys, registry name and return value as the reference's
`PyramidalPatchContrastMaximization` (src/solver/patch_contrast_pyramid.py:31-67, 149-177), so
`solver.collections["pyramidal_patch_contrast_maximization"](image_shape, calib, solver_cfg,
optimizer_cfg, output_cfg, None).optimize(events)` works as in `main.py:141-179`.

What is the same: the patch grids per scale (prepare_patch / prepare_pyramidal_patch, lines 69-100),
the objective per scale (`PatchFlowObjective` == `objective_scipy`, pinned by golden fixtures), the
SciPy method and options, warm start from the previous frame, fine-to-coarse feedback.
The per-patch re-initialisation at finer scales (initialize_guess_from_optuna_sampling, lines 320-353) uses the
reference's small-patch cost (calculate_cost_for_small_patch, 372-414: pinned by tests/golden/patch_search.npz) and
the reference's search box (sampling_initial, 417-428) -- but the trial points come from a regular grid scored in
one GPU launch (CMaxHandle.patch_search / cmax_patch_search) instead of Optuna's sequential TPE sampler.
What differs, and is NOT pinned against the reference (its dependencies are absent from this image and
its trajectories depend on Optuna's TPE sampler):
  * WHICH points of the search box are tried (grid vs. TPE), hence the starting point of the finer scales;
  * skimage.transform.pyramid_expand / pyramid_reduce (lines 220-222, 265-267) are restated with
    scipy.ndimage (order-1 zoom + Gaussian sigma = 2*2/6, 'reflect') on the tiny [2,ph,pw] arrays.
These are host-side operations on a few hundred numbers between scales, not part of the hot path.
"""
import logging
import os
from typing import Dict, Optional

import numpy as np
import scipy.ndimage as ndi

from ..cmax import CMaxHandle
from . import scipy_autograd
from .patch_objective import PatchFlowObjective
from .translation_search import ONE_PATCH_FIELD, WHOLE_IMAGE_FIELD, grid_search_translation

logger = logging.getLogger(__name__)

SCIPY_OPTIMIZERS = ["Nelder-Mead", "Powell", "CG", "BFGS", "Newton-CG", "L-BFGS-B", "TNC", "COBYLA", "SLSQP",
                    "trust-constr", "dogleg", "trust-ncg", "trust-exact", "trust-krylov"]


def _check_key_and_bool(config: dict, key: str) -> bool:
    return key in config.keys() and bool(config[key])


def search_box(m0: np.ndarray, abs_range: float = 10.0):
    """sampling_initial (patch_contrast_pyramid.py:417-428): per component the interval spanned by
    {0.8 m, m - 10, 1.2 m, m + 10}.  m0 [...]: -> (lo, hi) of the same shape."""
    c = np.stack([0.8 * m0, m0 - abs_range, 1.2 * m0, m0 + abs_range])
    return c.min(axis=0), c.max(axis=0)


def pyramid_expand(motion: np.ndarray) -> np.ndarray:
    """[2,h,w] -> [2,2h,2w]: skimage.transform.pyramid_expand(channel_axis=0) restated on scipy.ndimage -- order-1 resize
    (skimage's mode "reflect" is ndimage's "mirror" for the resampling step), then Gaussian smoothing with
    sigma = 2 * upscale / 6, for which skimage hands its mode string to ndimage unchanged ("reflect").  skimage is not
    in this image: the mapping follows its source as recalled, unpinned."""
    out = np.stack([ndi.zoom(c, 2, order=1, mode="mirror", grid_mode=True) for c in motion])
    return np.stack([ndi.gaussian_filter(c, 2 * 2 / 6.0, mode="reflect") for c in out])


def pyramid_reduce(motion: np.ndarray) -> np.ndarray:
    """[2,h,w] -> [2,ceil(h/2),ceil(w/2)]: Gaussian smoothing (sigma = 2 * downscale / 6, ndimage "reflect") then order-1
    down-sampling (ndimage "mirror"), as skimage.transform.pyramid_reduce does."""
    sm = np.stack([ndi.gaussian_filter(c, 2 * 2 / 6.0, mode="reflect") for c in motion])
    shape = (int(np.ceil(motion.shape[1] / 2)), int(np.ceil(motion.shape[2] / 2)))
    return np.stack([ndi.zoom(c, (shape[0] / c.shape[0], shape[1] / c.shape[1]), order=1, mode="mirror", grid_mode=True)
                     for c in sm])


class PyramidalPatchContrastMaximization:
    def __init__(self, image_shape: tuple, calibration_parameter: dict, solver_config: dict = {},
                 optimizer_config: dict = {}, output_config: dict = {}, visualize_module=None):
        self.image_shape = tuple(image_shape)
        self.calib_param = calibration_parameter
        self.slv_config = solver_config
        self.opt_config = optimizer_config
        self.out_config = output_config
        self.visualizer = visualize_module
        self.opt_method = optimizer_config["method"]
        if self.opt_method not in SCIPY_OPTIMIZERS:
            raise NotImplementedError(f"Optimizer {self.opt_method} is not supported (SciPy methods only)")
        self.padding = solver_config["outer_padding"] if "outer_padding" in solver_config else 0
        self.iwe_config = solver_config["iwe"]
        if self.iwe_config["method"] != "bilinear_vote":
            raise NotImplementedError("the fused objective accumulates with bilinear_vote")
        self.motion_model = solver_config["motion_model"]
        self.motion_vector_size = 2
        self.cost_name = solver_config["cost"]
        self.cost_weight = solver_config.get("cost_with_weight") if self.cost_name == "hybrid" else None
        self.normalize_t_in_batch = True
        # time-aware set-up (src/solver/base.py:211-228)
        self.is_time_aware = _check_key_and_bool(solver_config, "time_aware")
        if self.is_time_aware:
            self.time_bin = solver_config["time_bin"]
            self.flow_interpolation = solver_config["flow_interpolation"]
            self.t0_flow_location = solver_config["t0_flow_location"]
            if _check_key_and_bool(solver_config, "scale_later"):
                raise NotImplementedError("scale_later is not built")
        else:
            self.time_bin, self.flow_interpolation, self.t0_flow_location = 0, "burgers", "middle"
        # pyramid (patch_contrast_pyramid.py:50-61)
        patch = solver_config["patch"]
        self.filter_type = patch["filter_type"]
        self.coarest_scale = 1
        self.patch_scales = patch["scale"]
        self.cropped_image_shape = (patch["crop_height"], patch["crop_width"])
        self.patch_shift = ((self.image_shape[0] - self.cropped_image_shape[0]) // 2,
                            (self.image_shape[1] - self.cropped_image_shape[1]) // 2)
        self.scaled_patch_size, self.scaled_patch_image_size, self.scaled_n_patch = {}, {}, {}
        for s in range(self.coarest_scale, self.patch_scales):
            size = (self.cropped_image_shape[0] // (2 ** s), self.cropped_image_shape[1] // (2 ** s))
            self.scaled_patch_size[s] = size
            # patch centres: np.arange(0, image - patch + slide, slide) + patch/2 with slide == patch (lines 87-91)
            self.scaled_patch_image_size[s] = (len(np.arange(0, self.cropped_image_shape[0], size[0])),
                                               len(np.arange(0, self.cropped_image_shape[1], size[1])))
            self.scaled_n_patch[s] = self.scaled_patch_image_size[s][0] * self.scaled_patch_image_size[s][1]
        self.total_n_patch = sum(self.scaled_n_patch.values())
        self.previous_frame_best_estimation: Optional[Dict[int, np.ndarray]] = None
        self.history = []  # (scale, OptimizeResult)
        self.search_history = []  # (scale, candidates, loss, picked index) of the per-patch re-initialisation
        self._handle: Optional[CMaxHandle] = None
        self._objectives: Dict[int, PatchFlowObjective] = {}
        self._warper = self._imager = None  # built on first use by the metric / picture methods below
        self.iwe_visualize_max_scale = solver_config.get("max_scale", 50)

    # -- reference API ---------------------------------------------------------------------------
    def set_previous_frame_best_estimation(self, previous_best):
        self.previous_frame_best_estimation = previous_best.copy() if isinstance(previous_best, dict) else np.copy(previous_best)

    def initialize_random(self, n_patch: int) -> np.ndarray:
        x0 = np.random.rand(self.motion_vector_size, n_patch).astype(np.float64)
        p = self.opt_config["parameters"]
        x0[0] = x0[0] * (p["trans_x"]["max"] - p["trans_x"]["min"]) + p["trans_x"]["min"]
        x0[1] = x0[1] * (p["trans_y"]["max"] - p["trans_y"]["min"]) + p["trans_y"]["min"]
        return x0

    def optimize(self, events: np.ndarray) -> Dict[int, np.ndarray]:
        """events [n,4] (x row, y col, t, p) -> {scale: motion [2, ph, pw]} in pixel per time unit."""
        logger.info(f"DoF is {self.motion_vector_size * self.total_n_patch}")
        # one handle and one objective per scale for the life of the solver: the device workspaces, the sorted-event
        # buffers and the native plans are reused from frame to frame (main.py runs one solver over a whole sequence)
        if self._handle is None:
            # the per-scale objective warps with a DENSE flow: an event whose source pixel is off the sensor has no flow value (the
            # reference's gather indexes out of bounds there, src/warp.py:303-307) -- asked to be dropped while packing, and ...
            self._handle = CMaxHandle(self.image_shape, self.padding).set_keep_outside(False)
        # ... on_dropped="raise" (VERDICT r3 #8): a solver must not diverge from the reference silently
        handle = self._handle.set_events(events, time_bin=self.time_bin, on_dropped="raise")
        t = events[:, 2]
        t_scale = float(t.max() - t.min()) if self.normalize_t_in_batch else 1.0
        self.slab_history = []  # (scale, slab count in force when the scale's optimisation ended): large motions, DESIGN section 2
        best: Dict[int, np.ndarray] = {}
        self.history = []
        self.search_history = []
        for s in range(self.coarest_scale, self.patch_scales):
            pis = self.scaled_patch_image_size[s]
            objective = self._objectives.get(s)
            if objective is None:
                objective = self._objectives[s] = PatchFlowObjective(
                    handle, t_scale, pis, self.scaled_patch_size[s], self.scaled_patch_size[s], self.patch_shift,
                    cost=self.cost_name, cost_with_weight=self.cost_weight, blur_sigma=self.iwe_config["blur_sigma"],
                    time_aware=self.is_time_aware, time_bin=self.time_bin, flow_interpolation=self.flow_interpolation,
                    t0_flow_location=self.t0_flow_location, filter_type=self.filter_type)
            else:
                objective.set_t_scale(t_scale)
            if self.previous_frame_best_estimation is not None and s == self.coarest_scale:
                x0 = np.copy(self.previous_frame_best_estimation[s]).reshape(-1)
            elif s > self.coarest_scale:
                x0 = pyramid_expand(best[s - 1])[:, : pis[0], : pis[1]].reshape(-1)
                if self.previous_frame_best_estimation is not None:
                    x0 = (x0 + self.previous_frame_best_estimation[s].reshape(-1)) / 2
                x0 = self.initialize_guess_from_patch_search(handle, s, x0)
            elif self.slv_config["patch"]["initialize"] == "zero":
                x0 = np.zeros(2 * self.scaled_n_patch[s])
            elif self.slv_config["patch"]["initialize"] == "global-best":  # patch_contrast_pyramid.py:292-297
                best_guess = self.initialize_guess_from_whole_image(handle, t_scale)
                x0 = np.tile(best_guess[None], (self.scaled_n_patch[s], 1)).T.reshape(-1)
            elif self.slv_config["patch"]["initialize"] == "grid-best":  # patch_contrast_pyramid.py:298-305
                best_guess = self.initialize_guess_from_patch(events, s, self.scaled_n_patch[s] // 2 - 1)
                x0 = np.tile(best_guess[None], (self.scaled_n_patch[s], 1)).T.reshape(-1)
            else:
                x0 = self.initialize_random(self.scaled_n_patch[s]).reshape(-1)
            res = scipy_autograd.minimize(objective, x0, method=self.opt_method, precision="float64",
                                          torch_device=str(handle.device),
                                          options={"gtol": 1e-5, "disp": False, "maxiter": self.opt_config["max_iter"]}
                                          if self.opt_method in ("Newton-CG", "BFGS", "CG", "L-BFGS-B", "trust-ncg", "trust-krylov")
                                          else {"maxiter": self.opt_config["max_iter"]})
            logger.info(f"Scale {s}: loss {res.fun}")
            self.history.append((s, res))
            self.slab_history.append((s, handle.time_slabs))
            best[s] = np.asarray(res.x).reshape((2,) + pis)
        self._last_handle, self._last_t_scale = handle, t_scale
        return self.update_coarse_from_fine(best)

    # -- grid initialisers of the coarsest scale: K candidates per library call ------------------------------------------
    def _search_cost(self) -> dict:
        return {"cost": self.cost_name, "cost_with_weight": self.cost_weight, "sigma": self.iwe_config["blur_sigma"]}

    def initialize_guess_from_whole_image(self, handle: CMaxHandle, t_scale: float) -> np.ndarray:
        """patch.initialize: "global-best" (src/solver/patch_contrast_base.py:164-187): the best of the 30 x 30 translations
        np.arange(-150, 150, 10)^2 for the WHOLE batch under the solver's own cost, warped as one 2-DoF motion (`motion * t_scale`,
        objective_scipy_for_patch 244-271).  900 objective evaluations in the reference, 900 / 32 cmax_objective_batch calls here,
        the batch in time slabs while the candidates are large (CMaxHandle.auto_time_slabs)."""
        if self.is_time_aware:
            raise NotImplementedError("global-best initialisation on a time-binned handle is not built (the 2-DoF search wants slabs)")
        # a 2-DoF warp would keep events from off the sensor; this handle dropped them on request (and raised if there were any)
        guess, loss = grid_search_translation(handle, t_scale, WHOLE_IMAGE_FIELD, **self._search_cost())
        self.search_history.append((self.coarest_scale, "global-best", loss, guess))
        logger.info(f"Initial value: best_guess = {guess}")
        return guess

    def initialize_guess_from_patch(self, events: np.ndarray, s: int, patch_index: int) -> np.ndarray:
        """patch.initialize: "grid-best" (src/solver/patch_contrast_base.py:126-162): np.arange(-150, 150, 30)^2 on the events of ONE
        patch (utils.crop_event with the patch's bounds), its own time span as t_scale; warper and imager stay full-size."""
        if self.is_time_aware:
            raise NotImplementedError("grid-best initialisation on a time-binned handle is not built")
        x0, x1, y0, y1 = self.patch_boxes(s)[patch_index]
        m = (x0 <= events[:, 0]) & (events[:, 0] < x1) & (y0 <= events[:, 1]) & (events[:, 1] < y1)
        cropped = events[m]
        if len(cropped) == 0:  # "No events in the patch": every loss is 0, the first candidate wins
            return np.array([float(ONE_PATCH_FIELD[0]), float(ONE_PATCH_FIELD[0])])
        t_scale = float(cropped[:, 2].max() - cropped[:, 2].min()) if self.normalize_t_in_batch else 1.0
        sub = CMaxHandle(self.image_shape, self.padding).set_events(cropped)
        try:
            guess, loss = grid_search_translation(sub, t_scale, ONE_PATCH_FIELD, **self._search_cost())
        finally:
            sub.close()
        self.search_history.append((s, "grid-best", loss, guess))
        logger.info(f"Initial value: best_guess = {guess}")
        return guess

    def patch_boxes(self, s: int) -> np.ndarray:
        """[n_patch, 4] = x_min, x_max, y_min, y_max of scale s: FlowPatch bounds (src/types/flow_patch.py:29-42) of the
        centres prepare_patch lays out (patch_contrast_base.py:86-105).  As in the reference, these boxes live on the
        CROPPED grid without the (image - crop) / 2 shift the dense interpolation applies (patch_contrast_pyramid.py:
        325-335 crops the raw events with them)."""
        h, w = self.scaled_patch_size[s]
        cx = np.arange(0, self.cropped_image_shape[0], h) + h / 2
        cy = np.arange(0, self.cropped_image_shape[1], w) + w / 2
        xx, yy = np.meshgrid(cx, cy, indexing="ij")
        xx, yy = xx.reshape(-1), yy.reshape(-1)
        return np.stack([(xx - np.ceil(h / 2)).astype(int), (xx + np.floor(h / 2)).astype(int),
                         (yy - np.ceil(w / 2)).astype(int), (yy + np.floor(w / 2)).astype(int)], axis=1)

    def initialize_guess_from_patch_search(self, handle: CMaxHandle, s: int, motion0: np.ndarray) -> np.ndarray:
        """Role of initialize_guess_from_optuna_sampling (patch_contrast_pyramid.py:320-353): per patch, the best
        translation inside sampling_initial's box around motion0 under the small-patch cost; patches with <= 10 events
        keep motion0.  n_iter / (s - coarsest) trials per patch like the reference, placed on a g x g grid
        (g = ceil(sqrt(trials)), or patch["search_grid"]) plus motion0 itself; all patches and trials in one launch."""
        m0 = np.asarray(motion0, dtype=np.float64).reshape(2, -1)  # [2, n_patch]
        n_patch = m0.shape[1]
        size = self.scaled_patch_size[s]
        if self.padding:
            # the reference's small-patch imager is padded by outer_padding (EventImageConverter(scaled_size, outer_padding));
            # cmax_patch_search votes into the un-padded patch image.  No shipped YAML uses a padding: refuse rather than
            # score the candidates on a different image
            raise NotImplementedError("per-patch re-initialisation with outer_padding != 0 is not built")
        if 2 * size[0] * size[1] * 4 > 64 * 1024 - 256:  # patch image beyond the workgroup's LDS (scale 1 of a large crop)
            logger.info(f"Scale {s}: patch {size} too large for the batched search, keeping the expanded motion")
            return m0.reshape(-1)
        trials = max(1, int(np.ceil(self.opt_config["n_iter"] / (s - self.coarest_scale))))
        g = int(self.slv_config["patch"].get("search_grid", int(np.ceil(np.sqrt(trials)))))
        if g <= 0:  # search_grid: 0 switches the re-initialisation off (A/B runs)
            return m0.reshape(-1)
        lo, hi = search_box(m0)  # [2, n_patch]
        u = np.linspace(0.0, 1.0, g) if g > 1 else np.array([0.5])
        gx = lo[0][:, None] + (hi[0] - lo[0])[:, None] * u[None, :]  # [n_patch, g]
        gy = lo[1][:, None] + (hi[1] - lo[1])[:, None] * u[None, :]
        cand = np.stack([np.repeat(gx, g, axis=1), np.tile(gy, (1, g))], axis=-1)  # [n_patch, g*g, 2]
        cand = np.concatenate([m0.T[:, None, :], cand], axis=1)
        loss, _, count = handle.patch_search(self.patch_boxes(s), size, cand, self.iwe_config["blur_sigma"])
        loss = loss.cpu().numpy()
        # the reference turns a NaN loss into 0.0 (calculate_cost_for_small_patch / objective_initial,
        # patch_contrast_pyramid.py:376-378, 411-414) -- under "minimize" that trial then wins; mirrored as it is
        loss[np.isnan(loss)] = 0.0
        pick = loss.argmin(axis=1)
        m1 = cand[np.arange(n_patch), pick].T.copy()  # [2, n_patch]
        keep = count.cpu().numpy() <= 10
        m1[:, keep] = m0[:, keep]
        self.search_history.append((s, cand, loss, pick))
        return m1.reshape(-1)

    def update_coarse_from_fine(self, motion_per_scale: dict) -> dict:
        """patch_contrast_pyramid.py:205-222: the finest motion, and below it the reduced OPTIMISED motion of the next finer
        scale -- keys finest ... coarsest - 1, exactly the reference's (the extra key coarsest - 1 is never read)."""
        finest, coarsest = max(motion_per_scale), min(motion_per_scale)
        refined = {finest: motion_per_scale[finest]}
        for i in range(finest, coarsest - 1, -1):
            refined[i - 1] = pyramid_reduce(motion_per_scale[i])
        return refined

    def motion_to_dense_flow(self, motion_per_scale: dict, t_scale: float = 1.0) -> np.ndarray:
        """Finest-scale motion -> dense flow [2,H,W] in pixel per time unit (patch_contrast_pyramid.py:464-516); time-aware solvers:
        the flow voxel [time_bin,2,H,W] propagated from it at displacement scale (flow * t_scale through the Burgers / upwind chain,
        divided by t_scale again)."""
        import torch

        from .. import functional as F
        from ..utils.flow_utils import construct_dense_flow_voxel_torch
        from .patch_objective import patch_pad

        s = max(motion_per_scale)
        ps = self.scaled_patch_size[s]
        m = torch.as_tensor(np.asarray(motion_per_scale[s]), dtype=torch.float64, device="cuda")
        dense = F.patch_to_dense(m, self.image_shape, ps, patch_pad(ps, ps, self.patch_shift))
        if not self.is_time_aware:
            return dense.cpu().numpy()
        voxel = construct_dense_flow_voxel_torch(dense * t_scale, self.time_bin, self.flow_interpolation,
                                                 t0_location=self.t0_flow_location) / t_scale
        return voxel.cpu().numpy()

    # -- what main.py calls on a solver besides optimize(): metrics and (optional) pictures ---------------------------------------
    # (src/solver/base.py:230-250, 272-420, 543-660 and their overrides in patch_contrast_pyramid.py:518-660.)  The arithmetic of the
    # metrics is the library's (Warp / EventImageConverter / costs over the HIP kernels) plus host NumPy for the end-point errors;
    # the pictures are drawn by whatever `visualize_module` the caller passed -- without one every visualize_* returns at once.
    @property
    def motion_model_for_dense_warp(self) -> str:
        return "dense-flow-voxel" if self.is_time_aware else "dense-flow"

    @property
    def warper(self):
        if self._warper is None:
            from ..warp import Warp

            self._warper = Warp(self.image_shape, calculate_feature=True, normalize_t=self.normalize_t_in_batch, calib_param=self.calib_param)
        return self._warper

    @property
    def imager(self):
        if self._imager is None:
            from ..event_image_converter import EventImageConverter

            self._imager = EventImageConverter(self.image_shape, outer_padding=self.padding)
        return self._imager

    def get_original_flow_from_time_aware_flow_voxel(self, flow_voxel: np.ndarray) -> np.ndarray:
        """[(b,) time_bin, 2, H, W] -> the slice the flow was given at (src/solver/base.py:230-250)."""
        if flow_voxel.ndim == 4:
            flow_voxel = flow_voxel[None]
        if self.t0_flow_location == "first":
            orig_ind = 0
        elif self.t0_flow_location == "middle":
            orig_ind = flow_voxel.shape[1] // 2
        else:
            raise NotImplementedError(f"t0_flow_location {self.t0_flow_location}")
        return np.squeeze(flow_voxel[:, orig_ind])

    def create_clipped_iwe_for_visualization(self, events: np.ndarray, max_scale=50) -> np.ndarray:
        """src/solver/base.py:272-291: un-blurred IWE as an inverted 8-bit picture, padding cut off."""
        assert events.shape[-1] <= 4, "this function is for events"
        if hasattr(events, "detach"):
            events = events.clone().detach().cpu().numpy()
        im = self.imager.create_image_from_events_numpy(events, method=self.iwe_config["method"], sigma=0)
        clipped_iwe = 255 - np.clip(max_scale * im, 0, 255).astype(np.uint8)
        if self.padding > 0:
            clipped_iwe = clipped_iwe[self.padding: -self.padding, self.padding: -self.padding]
        return clipped_iwe

    @staticmethod
    def _batch_t_scale(events: np.ndarray) -> float:
        return float(np.max(events[:, 2]) - np.min(events[:, 2]))

    def visualize_one_batch_warp(self, events: np.ndarray, warp: Optional[dict] = None):
        """patch_contrast_pyramid.py:518-536."""
        if self.visualizer is None:
            return
        flow = None
        if warp is not None:
            flow = self.motion_to_dense_flow(warp)
            if self.normalize_t_in_batch:
                flow = flow * self._batch_t_scale(events)
            events, _ = self.warper.warp_event(events, flow, self.motion_model_for_dense_warp)
            if self.is_time_aware:
                flow = self.get_original_flow_from_time_aware_flow_voxel(flow)
        clipped_iwe = self.create_clipped_iwe_for_visualization(events, max_scale=self.iwe_visualize_max_scale)
        self.visualizer.visualize_image(clipped_iwe)
        if warp is not None:
            self.visualizer.visualize_optical_flow_on_event_mask(flow, events)
            self.visualizer.visualize_overlay_optical_flow_on_event(flow, clipped_iwe)

    def visualize_one_batch_warp_gt(self, events: np.ndarray, gt_warp: np.ndarray, motion_model: str = "dense-flow"):
        """src/solver/base.py:313-330; gt_warp [H, W, 2] for "dense-flow"."""
        if self.visualizer is None:
            return
        if motion_model == "dense-flow":
            gt_warp = np.transpose(gt_warp, (2, 0, 1))
        events, _ = self.warper.warp_event(events, gt_warp, motion_model=motion_model)
        clipped_iwe = self.create_clipped_iwe_for_visualization(events, max_scale=self.iwe_visualize_max_scale)
        self.visualizer.visualize_image(clipped_iwe)
        if motion_model == "dense-flow":
            self.visualizer.visualize_overlay_optical_flow_on_event(gt_warp, clipped_iwe)

    def visualize_original_sequential(self, events: np.ndarray):
        """src/solver/base.py:332-341."""
        if self.visualizer is None:
            return
        clipped_iwe = self.create_clipped_iwe_for_visualization(events, max_scale=self.iwe_visualize_max_scale)
        self.visualizer.visualize_image(clipped_iwe, file_prefix="original")

    def visualize_pred_sequential(self, events: np.ndarray, warp: dict):
        """patch_contrast_pyramid.py:538-558 (warped to the MIDDLE of the batch, as the reference's override does)."""
        if self.visualizer is None:
            return
        t_scale = self._batch_t_scale(events) if self.normalize_t_in_batch else 1.0
        flow = self.motion_to_dense_flow(warp, t_scale) * t_scale
        events, _ = self.warper.warp_event(events, flow, self.motion_model_for_dense_warp, direction="middle")
        clipped_iwe = self.create_clipped_iwe_for_visualization(events, max_scale=self.iwe_visualize_max_scale)
        if self.is_time_aware:
            flow = self.get_original_flow_from_time_aware_flow_voxel(flow)
        self.visualizer.visualize_image(clipped_iwe, file_prefix="pred_warp")
        self.visualizer.visualize_optical_flow_on_event_mask(flow, events, file_prefix="pred_masked")

    def visualize_gt_sequential(self, events: np.ndarray, gt_warp: np.ndarray, gt_type: str = "flow"):
        """src/solver/base.py:385-420; gt_warp [H, W, 2] displacement."""
        if self.visualizer is None:
            return
        if gt_type != "flow":
            raise NotImplementedError("the patch solver's ground truth is a flow")
        gt_flow = np.transpose(gt_warp, (2, 0, 1))
        events, _ = self.warper.warp_event(events, gt_flow, "dense-flow", direction="first")
        clipped_iwe = self.create_clipped_iwe_for_visualization(events, max_scale=self.iwe_visualize_max_scale)
        self.visualizer.visualize_image(clipped_iwe, file_prefix="gt_warp")
        self.visualizer.visualize_optical_flow(gt_flow[0], gt_flow[1], visualize_color_wheel=False, file_prefix="gt_flow")

    def calculate_flow_error(self, motion: dict, gt_flow: np.ndarray, timescale: float = 1.0,
                             events: Optional[np.ndarray] = None) -> dict:
        """patch_contrast_pyramid.py:560-599: EPE / nPE / AE of the predicted displacement against gt_flow [H, W, 2] (masked by
        the pixels that saw an event when `events` is given), plus the flow-warp-loss ratios of calculate_fwl."""
        gt_flow = np.transpose(gt_flow, (2, 0, 1))  # [2, H, W]
        pred_flow = self.motion_to_dense_flow(motion, timescale) * timescale
        if self.is_time_aware:
            pred_flow = self.get_original_flow_from_time_aware_flow_voxel(pred_flow)
        pred_flow = pred_flow[None]
        if events is not None:
            event_mask = self.imager.create_eventmask(events)
            if self.padding:
                event_mask = event_mask[..., self.padding: -self.padding, self.padding: -self.padding]
            fwl = self.calculate_fwl(motion, gt_flow, timescale, events)
        else:
            event_mask, fwl = None, {}
        flow_error = calculate_flow_error_numpy(gt_flow[None], pred_flow, event_mask=event_mask)
        flow_error.update(fwl)
        logger.info(f"{flow_error = } for time period {timescale} sec.")
        return flow_error

    def calculate_fwl(self, motion: dict, gt_flow: np.ndarray, timescale: float, events: np.ndarray) -> dict:
        """FWL (Stoffregen 2020) as the reference reports it, Var(IWE_orig) / Var(IWE): below 1 = sharper than the un-warped image
        (patch_contrast_pyramid.py:601-629).  gt_flow [2, H, W] displacement over the batch."""
        from .. import costs
        from ..warp import Warp

        orig_iwe = self.imager.create_iwe(events)
        gt_warper = Warp(self.image_shape, normalize_t=True)
        gt_warp, _ = gt_warper.warp_event(events, gt_flow, "dense-flow")
        gt_iwe = self.imager.create_iwe(gt_warp)
        gt_fwl = costs.NormalizedImageVariance().calculate({"orig_iwe": orig_iwe, "iwe": gt_iwe, "omit_boundary": False})
        fwl = {"GT_FWL": gt_fwl}
        fwl.update(self.calculate_fwl_pred(motion, events, timescale))
        return fwl

    def calculate_fwl_pred(self, motion: dict, events: np.ndarray, timescale: float = 1.0) -> dict:
        """patch_contrast_pyramid.py:631-660."""
        from .. import costs

        orig_iwe = self.imager.create_iwe(events)
        pred_flow = self.motion_to_dense_flow(motion, timescale) * timescale
        pred_warp, _ = self.warper.warp_event(events, pred_flow, self.motion_model_for_dense_warp)
        pred_iwe = self.imager.create_iwe(pred_warp)
        pred_fwl = costs.NormalizedImageVariance().calculate({"orig_iwe": orig_iwe, "iwe": pred_iwe, "omit_boundary": False})
        return {"PRED_FWL": pred_fwl}

    def save_flow_error_as_text(self, nth_frame: int, flow_error_dict: dict, fname: str = "flow_error_per_frame.txt"):
        """src/solver/base.py:651-660: one line per frame appended to <visualizer.save_dir>/<fname> (the working directory
        without a visualizer)."""
        save_file_name = os.path.join(self.visualizer.save_dir, fname) if self.visualizer is not None else fname
        with open(save_file_name, "a") as f:
            f.write(f"frame {nth_frame}::" + str(flow_error_dict) + "\n")


def calculate_flow_error_numpy(flow_gt: np.ndarray, flow_pred: np.ndarray, event_mask: Optional[np.ndarray] = None) -> dict:
    """End-point and angular errors of src/utils/flow_utils.py:705-758.  flow_gt / flow_pred [B, 2, H, W], event_mask [B, 1, H, W].
    Only pixels whose ground truth is finite and non-zero in both components count (and, with a mask, saw an event); n_points carries
    the reference's + 1e-5."""
    assert flow_gt.ndim == flow_pred.ndim == 4
    finite = ~np.isinf(flow_gt[:, [0]]) & ~np.isinf(flow_gt[:, [1]])
    moving = (np.abs(flow_gt[:, [0]]) > 0) & (np.abs(flow_gt[:, [1]]) > 0)
    total_mask = finite & moving
    if event_mask is not None:
        total_mask = np.logical_and(event_mask, total_mask)
    with np.errstate(invalid="ignore"):  # inf * False
        gt_masked = flow_gt * total_mask
    pred_masked = flow_pred * total_mask
    n_points = np.sum(total_mask, axis=(1, 2, 3)) + 1e-5
    errors = {}
    epe = np.linalg.norm(gt_masked - pred_masked, axis=1)
    errors["EPE"] = np.mean(np.sum(epe, axis=(1, 2)) / n_points)
    for k in (1, 2, 3, 5, 10, 20):
        errors[f"{k}PE"] = np.mean(np.sum(epe > k, axis=(1, 2)) / n_points)
    u, v = pred_masked[:, 0], pred_masked[:, 1]
    u_gt, v_gt = gt_masked[:, 0], gt_masked[:, 1]
    cosine = (1.0 + u * u_gt + v * v_gt) / (np.sqrt(1 + u * u + v * v) * np.sqrt(1 + u_gt * u_gt + v_gt * v_gt))
    errors["AE"] = np.mean(np.sum(np.arccos(cosine), axis=(1, 2)) / n_points)
    return errors
