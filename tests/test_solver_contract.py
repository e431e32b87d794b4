"""The drop-in solver class answers every call the reference's driver makes (main.py:85-107 evaluation loop, 170-189 single frame):
optimize, set_previous_frame_best_estimation, calculate_flow_error, save_flow_error_as_text, visualize_* (no-ops without a visualizer).

CPU part: the host arithmetic (end-point / angular errors, the FWL call sequence) against tests/golden/flow_error.npz = the REFERENCE's own
calculate_flow_error output (gen_golden.py flow_error), with the device work of the class stood in for by the oracle.  The GPU part
(tests/test_gpu_solver.py) runs the same fixture through the real class."""
import os

import numpy as np
import pytest

import event_based_optical_flow_amd as E
from event_based_optical_flow_amd.solver import pyramid as P
from oracle import oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "flow_error.npz")
METRICS = ["EPE", "1PE", "2PE", "3PE", "5PE", "10PE", "20PE", "AE"]


def solver_config(time_aware: bool, scale: int = 3, crop=(64, 80)):
    cfg = {"method": "pyramidal_patch_contrast_maximization", "time_aware": time_aware,
           "patch": {"initialize": "random", "scale": scale, "crop_height": crop[0], "crop_width": crop[1], "filter_type": "bilinear"},
           "motion_model": "2d-translation", "warp_direction": "first", "parameters": ["trans_x", "trans_y"], "cost": "hybrid",
           "outer_padding": 0, "cost_with_weight": {"multi_focal_normalized_gradient_magnitude": 1.0, "total_variation": 0.01},
           "iwe": {"method": "bilinear_vote", "blur_sigma": 1}}
    if time_aware:
        cfg.update({"time_bin": 10, "flow_interpolation": "burgers", "t0_flow_location": "middle"})
    return cfg


OPT_CFG = {"n_iter": 40, "method": "Newton-CG", "max_iter": 25,
           "parameters": {"trans_x": {"min": -150, "max": 150}, "trans_y": {"min": -150, "max": 150}}}


def test_flow_error_metrics_match_the_reference():
    g = np.load(GOLD)
    period = float(g["timescale"])
    gt = np.transpose(g["gt_flow"], (2, 0, 1))[None]
    pred = (g["plain__dense"] * period)[None]
    got = P.calculate_flow_error_numpy(gt, pred)
    for k in METRICS:
        assert got[k] == pytest.approx(float(g[f"plain__nomask__{k}"]), rel=1e-12, abs=1e-15), k
    # an infinite ground-truth pixel: the reference's `flow_gt * mask` makes it NaN (inf * False) -- mirrored, not repaired
    with np.errstate(invalid="ignore"):
        bad = P.calculate_flow_error_numpy(np.transpose(g["gt_flow_inf"], (2, 0, 1))[None], pred)
    assert np.isnan(bad["EPE"]) and np.isnan(float(g["plain__inf__EPE"]))
    assert bad["1PE"] == pytest.approx(float(g["plain__inf__1PE"]), rel=1e-12)


class OracleImager:
    """EventImageConverter's numpy branch on the oracle (eps 1e-8 vote, scipy-style Gaussian)."""

    def __init__(self, image_size):
        self.image_size = image_size

    def create_iwe(self, events, method="bilinear_vote", sigma=1):
        img = orc.vote(events, self.image_size, 0, 1.0, 1e-8, method)
        return orc.gaussian_filter(img, sigma) if sigma > 0 else img

    def create_image_from_events_numpy(self, events, method="bilinear_vote", weight=1.0, sigma=1):
        return self.create_iwe(events, method, sigma)

    def create_eventmask(self, events):
        return (0 != self.create_iwe(events, sigma=0))[None]


class OracleWarper:
    def __init__(self, image_size, normalize_t=True, **_kw):
        self.image_size, self.normalize_t = image_size, normalize_t

    def warp_event(self, events, motion, motion_model, direction="first"):
        return orc.warp_event(events, motion, motion_model, direction, self.image_size, self.normalize_t)[0], {}


class OracleNormalizedVariance:
    def calculate(self, arg):  # numpy branch: biased variances, only `iwe` may be cropped (normalized_image_variance.py:40-64)
        return orc.variance(arg["orig_iwe"], False, 0, False)[0] / orc.variance(arg["iwe"], arg["omit_boundary"], 0, False)[0]


class Recorder:
    """Stands for visualizer.Visualizer: records what was drawn."""

    def __init__(self, save_dir):
        self.save_dir = save_dir
        self.calls = []

    def __getattr__(self, name):
        if not name.startswith("visualize_"):
            raise AttributeError(name)
        return lambda *a, **k: self.calls.append((name, k.get("file_prefix")))


@pytest.mark.parametrize("tag,time_aware", [("plain", False), ("burgers", True)])
def test_driver_call_sequence_on_the_solver_class(tag, time_aware, tmp_path, monkeypatch):
    g = np.load(GOLD)
    H, W = (int(v) for v in g["image_size"])
    period = float(g["timescale"])
    events, gt_flow = g["events"], g["gt_flow"]
    s = int(g[tag + "__scale"])
    best = {s: g[tag + "__motion"]}
    monkeypatch.setattr(E.costs, "NormalizedImageVariance", OracleNormalizedVariance)
    monkeypatch.setattr(E.warp, "Warp", OracleWarper)
    for viz in (None, Recorder(str(tmp_path))):
        solv = E.solver.collections["pyramidal_patch_contrast_maximization"]((H, W), {}, solver_config(time_aware), OPT_CFG, {}, viz)
        assert solv.motion_model_for_dense_warp == ("dense-flow-voxel" if time_aware else "dense-flow")
        # the device work, stood in for by the oracle / the fixture's dense flow (pixel per second; voxel for the time-aware solver)
        solv._imager, solv._warper = OracleImager((H, W)), OracleWarper((H, W))
        monkeypatch.setattr(solv, "optimize", lambda ev: best)
        monkeypatch.setattr(solv, "motion_to_dense_flow", lambda m, t_scale=1.0: g[tag + "__dense"].copy())
        # main.py:170-189 (single frame) ...
        solv.visualize_one_batch_warp(events)
        best_motion = solv.optimize(events)
        solv.visualize_one_batch_warp(events, best_motion)
        solv.visualize_one_batch_warp_gt(events, gt_flow)
        err = solv.calculate_flow_error(best_motion, gt_flow, period, events)
        # ... and main.py:98-107 (evaluation loop)
        solv.set_previous_frame_best_estimation(best_motion)
        assert solv.previous_frame_best_estimation[s] is best_motion[s]
        solv.save_flow_error_as_text(7, err, str(tmp_path / "flow_error_per_frame_with_mask.txt") if viz is None else "flow_error_per_frame_with_mask.txt")
        solv.visualize_original_sequential(events)
        solv.visualize_pred_sequential(events, best_motion)
        solv.visualize_gt_sequential(events, gt_flow)
        for k in METRICS + ["GT_FWL", "PRED_FWL"]:
            assert err[k] == pytest.approx(float(g[f"{tag}__mask__{k}"]), rel=1e-9, abs=1e-12), k
        assert solv.calculate_fwl_pred(best_motion, events, period)["PRED_FWL"] == pytest.approx(float(g[tag + "__fwl_pred_only"]), rel=1e-9)
        line = open(tmp_path / "flow_error_per_frame_with_mask.txt").read().splitlines()[-1]
        assert line.startswith("frame 7::{") and "'EPE'" in line and "'PRED_FWL'" in line
        if viz is not None:
            names = [c[0] for c in viz.calls]
            assert names.count("visualize_image") == 6  # before / after / gt warp, original, pred, gt
            assert ("visualize_image", "original") in viz.calls and ("visualize_image", "pred_warp") in viz.calls
            assert ("visualize_image", "gt_warp") in viz.calls and ("visualize_optical_flow", "gt_flow") in viz.calls
            assert ("visualize_optical_flow_on_event_mask", "pred_masked") in viz.calls
            assert names.count("visualize_overlay_optical_flow_on_event") == 2
        err_nomask = solv.calculate_flow_error(best_motion, gt_flow, period)
        for k in METRICS:
            assert err_nomask[k] == pytest.approx(float(g[f"{tag}__nomask__{k}"]), rel=1e-9, abs=1e-12), k
        assert "GT_FWL" not in err_nomask
