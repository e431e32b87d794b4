"""CPU-only checks of the host layer: the C-ABI library loads and exports every symbol
include/cmax_hip.h declares, the ctypes binding matches the header, the reference's registries
and error conventions are mirrored, and the product path refuses to run without a GPU (no CPU
fallback).  No kernel is launched here."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

import event_based_optical_flow_amd as E
from event_based_optical_flow_amd import _lib, build
from event_based_optical_flow_amd import functional as F
from event_based_optical_flow_amd.cmax import FUSED_COSTS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cmax_hip.h")
NO_GPU = not torch.cuda.is_available()


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cmax_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_bound_and_exported():
    decl = declared_symbols()
    assert len(decl) >= 25
    assert sorted(_lib.SIGNATURES) == decl, "ctypes binding and include/cmax_hip.h disagree"
    lib = _lib.load()  # builds with hipcc if the .so is missing; raises if a symbol is absent
    for name in decl:
        assert hasattr(lib, name)
    out = subprocess.check_output(["nm", "-D", "--defined-only", build.LIB_PATH], text=True)
    exported = set(re.findall(r" T (cmax_[a-z0-9_]+)", out))
    assert set(decl) <= exported
    assert lib.cmax_abi_version() == _lib.ABI_VERSION
    assert lib.cmax_sizeof_objective() == ctypes.sizeof(_lib.CmaxObjective)


def test_header_constants_match_binding():
    text = open(HEADER).read()
    consts = dict(re.findall(r"#define\s+(CMAX_[A-Z0-9_]+)\s+(-?\d+)", text))
    assert int(consts["CMAX_F32"]) == _lib.F32 and int(consts["CMAX_F64"]) == _lib.F64
    assert (int(consts["CMAX_MODEL_2DOF"]), int(consts["CMAX_MODEL_DENSE"]), int(consts["CMAX_MODEL_VOXEL"])) == (
        _lib.MODEL_2DOF, _lib.MODEL_DENSE, _lib.MODEL_VOXEL)
    assert (int(consts["CMAX_REF_FIRST"]), int(consts["CMAX_REF_LAST"]), int(consts["CMAX_REF_FRAC"])) == (
        _lib.REF_FIRST, _lib.REF_LAST, _lib.REF_FRAC)
    assert (int(consts["CMAX_COST_VARIANCE"]), int(consts["CMAX_COST_GRADMAG"])) == (_lib.COST_VARIANCE, _lib.COST_GRADMAG)
    assert int(consts["CMAX_ABI_VERSION"]) == _lib.ABI_VERSION


def test_bad_arguments_are_rejected_before_any_launch():
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.cmax_create(0, 10, 0, 0, ctypes.byref(h)) == -1  # CMAX_EINVAL
    assert b"image size" in lib.cmax_last_error()
    assert lib.cmax_create(5000, 10, 0, 0, ctypes.byref(h)) == -1
    assert lib.cmax_vote(None, _lib.F32, 4, 10, None, 1.0, 0, 5, 0, 0, 1e-6, 0, None, None) == -1
    assert lib.cmax_warp_events(None, _lib.F32, 5, 7, None, 0, 4, 4, None, 0, 0.0, 1, None, None, None, None) == -1
    assert lib.cmax_destroy(None) == 0


@pytest.mark.skipif(not NO_GPU, reason="checks the no-GPU behaviour")
def test_no_cpu_fallback_without_gpu():
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.cmax_create(10, 10, 0, 0, ctypes.byref(h)) == -4  # CMAX_ENODEV
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        E.CMaxHandle((10, 10))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        E.EventImageConverter((10, 10)).create_iwe(np.zeros((3, 4)), sigma=0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        E.Warp((10, 10)).warp_event(np.zeros((3, 4)), np.zeros(2), "2d-translation")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        E.costs.ImageVariance().calculate({"iwe": np.zeros((5, 5)), "omit_boundary": True})


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "event_based_optical_flow_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "cmax_oracle" not in src, f


def test_cost_registry_and_protocol():
    # reference: src/costs/__init__.py:23-38
    assert sorted(E.costs.functions) == sorted([
        "image_variance", "gradient_magnitude", "total_variation", "normalized_image_variance",
        "normalized_gradient_magnitude", "multi_focal_normalized_image_variance",
        "multi_focal_normalized_gradient_magnitude"])
    for name, cls in E.costs.functions.items():
        assert issubclass(cls, E.costs.CostBase) and cls.name == name and cls.required_keys
    with pytest.raises(ValueError):
        E.costs.ImageVariance(direction="up")
    c = E.costs.ImageVariance(store_history=True)
    assert c.get_history() == {"loss": []}
    c.disable_history_register()
    assert c.store_history is False
    with pytest.raises(KeyError):
        c.calculate({"iwe": np.zeros((4, 4))})
    h = E.costs.HybridCost("minimize", {"multi_focal_normalized_gradient_magnitude": 1.0, "total_variation": 0.01})
    assert {"forward_iwe", "backward_iwe", "middle_iwe", "orig_iwe", "flow", "omit_boundary"} <= set(h.required_keys)
    assert set(h.get_history()) == {"loss", "multi_focal_normalized_gradient_magnitude", "total_variation"}
    with pytest.raises(KeyError):
        E.costs.HybridCost("minimize", {"no_such_cost": 1.0})
    assert set(FUSED_COSTS) == set(E.costs.functions) - {"total_variation"}


def test_warp_host_helpers_and_errors():
    # reference tests/test_warp.py:8-93 (CPU parts)
    w = E.Warp((100, 200), normalize_t=True)
    assert w.get_motion_vector_size("2d-translation") == 2
    assert w.get_key_names("rigid-optical-flow") == ["trans_x", "trans_y"]
    with pytest.raises(E.MotionModelKeyError):
        w.get_key_names("affine")
    with pytest.raises(E.MotionModelKeyError):
        w.warp_event(np.zeros((2, 4)), np.zeros(3), "affine")
    with pytest.raises(ValueError):  # a7: the time-bin count is the model's parameter (src/warp.py:176)
        w.warp_event(np.zeros((2, 4)), np.zeros((2, 100, 200)), "dense-flow-voxel-optimized")
    for lo, hi, ref, exp in ((1, 2, 1.0, (0, 1)), (0, 0.5, 0.0, (0, 1)), (-1, 1, 0.0, (-0.5, 0.5)), (-1, 1, -1.0, (0, 1))):
        ev = E.utils.generate_events(300, 100, 200, tmin=lo, tmax=hi, seed=1)
        dt = w.calculate_dt(ev, ref)
        np.testing.assert_allclose([dt.min(), dt.max()], exp, atol=0.1)
        dtt = w.calculate_dt(torch.from_numpy(ev), ref).numpy()
        np.testing.assert_allclose(dtt, dt)
    raw = E.Warp((10, 20), normalize_t=False)
    ev = E.utils.generate_events(300, 10, 20, tmin=-1, tmax=1, seed=2)
    np.testing.assert_allclose(raw.calculate_dt(ev, -1).max(), 2.0, atol=0.1)
    batch = np.stack([E.utils.generate_events(300, 10, 20, tmin=1, tmax=i + 2, seed=i) for i in range(4)])
    dt = E.Warp((10, 20), normalize_t=True).calculate_dt(batch, 1.0)
    assert dt.shape == (4, 300)
    np.testing.assert_allclose(dt.max(axis=-1), 1.0, atol=0.1)
    assert w.calculate_reftime(ev, "first") == ev[:, 2].min() and w.calculate_reftime(ev, "last") == ev[:, 2].max()
    np.testing.assert_allclose(w.calculate_reftime(ev, "middle"), 0.5 * (ev[:, 2].min() + ev[:, 2].max()))
    with pytest.raises(ValueError):
        w.calculate_reftime(ev, "sideways")


def test_direction_and_descriptor_host_logic():
    assert F.direction_to_ref("first") == (_lib.REF_FIRST, 0.0)
    assert F.direction_to_ref("last") == (_lib.REF_LAST, 1.0)
    assert F.direction_to_ref("middle") == (_lib.REF_FRAC, 0.5)
    assert F.direction_to_ref("before") == (_lib.REF_FRAC, -1.0)
    assert F.direction_to_ref(0.25) == (_lib.REF_FRAC, 0.25)
    with pytest.raises(ValueError):
        F.direction_to_ref("sideways")
    d = E.make_descriptor("multi_focal_normalized_gradient_magnitude", "dense-flow", sigma=1)
    assert (d.cost, d.normalized, d.n_ref, d.minimize, d.negate) == (_lib.COST_GRADMAG, 1, 3, 1, 0)
    assert list(d.ref_mode[:3]) == [_lib.REF_LAST, _lib.REF_FIRST, _lib.REF_FRAC] and list(d.mult[:3]) == [1.0, 1.0, 2.0]
    d = E.make_descriptor("image_variance", "2d-translation", direction="natural", warp_direction="middle")
    assert (d.n_ref, d.minimize, d.ref_mode[0], d.ref_frac[0]) == (1, 0, _lib.REF_FRAC, 0.5)
    assert E.make_descriptor("multi_focal_normalized_image_variance", "dense-flow", direction="maximize").negate == 1
    with pytest.raises(KeyError):
        E.make_descriptor("total_variation", "dense-flow")


def test_event_image_converter_host_logic():
    im = E.EventImageConverter((10, 20), outer_padding=3)
    assert im.image_size == (16, 26) and im.outer_padding == (3, 3)  # reference line 28
    im.update_property(outer_padding=2)  # the reference adds the padding once here (line 42)
    assert im.image_size == (18, 28)
    with pytest.raises(RuntimeError):
        im.create_iwe("not an array")


def test_synthetic_generators():
    ev = E.utils.generate_events(1000, 26, 34, 0.0, 0.05, seed=46)
    assert ev.shape == (1000, 4) and np.all(np.diff(ev[:, 2]) >= 0)
    assert np.all(ev[:, 0] == np.floor(ev[:, 0])) and ev[:, 0].max() < 26 and ev[:, 1].max() < 34
    np.testing.assert_array_equal(ev, E.utils.generate_events(1000, 26, 34, 0.0, 0.05, seed=46))
    st = E.utils.generate_structured_events(2000, 26, 34, (5.0, -3.0), n_dots=20, seed=1)
    assert st[:, 0].min() >= 0 and st[:, 0].max() <= 25
    f = E.utils.generate_smooth_flow((26, 34), 20, seed=2)
    assert f.shape == (2, 26, 34) and np.abs(f).max() <= 20


def test_minimize_adapter_on_a_plain_torch_function():
    """scipy_autograd.minimize (reference src/solver/scipy_autograd/scipy_minimize.py:6-19): value+grad
    through autograd, Hessian-vector product by central differences of the gradient for Newton-CG."""
    from event_based_optical_flow_amd.solver.scipy_autograd import TorchWrapper, minimize

    target = torch.tensor([[1.0, -2.0], [3.0, 0.5]], dtype=torch.float64)
    f = lambda x: ((x - target) ** 2).sum() + 0.1 * (x ** 4).sum()  # noqa: E731
    ref = None
    for method in ("BFGS", "L-BFGS-B", "Newton-CG", "trust-ncg", "CG"):
        res = minimize(f, np.zeros((2, 2)), method=method, precision="float64")
        assert res.x.shape == (2, 2)  # reshaped like x0 (line 117)
        ref = res.x if ref is None else ref
        np.testing.assert_allclose(res.x, ref, atol=2e-4)
    w = TorchWrapper(f, precision="float64")
    x = w.get_input(np.array([[0.3, -0.2], [0.1, 0.4]]))
    v = np.array([1.0, -0.5, 0.25, 2.0])
    hv = w.get_hvp(x, v)
    exact = (2.0 + 1.2 * x ** 2) * v  # diagonal Hessian of f
    np.testing.assert_allclose(hv, exact, rtol=1e-5)
    with pytest.raises(ValueError):
        TorchWrapper(f, precision="float16")
    with pytest.raises(NotImplementedError):
        minimize(f, np.zeros(2), method="dogleg")


def _pyramid_solver(n_iter=40):
    from event_based_optical_flow_amd import solver

    slv_cfg = {"method": "pyramidal_patch_contrast_maximization", "time_aware": False,
               "patch": {"initialize": "random", "scale": 4, "crop_height": 64, "crop_width": 80, "filter_type": "bilinear"},
               "motion_model": "2d-translation", "warp_direction": "first", "parameters": ["trans_x", "trans_y"],
               "cost": "hybrid", "outer_padding": 0,
               "cost_with_weight": {"multi_focal_normalized_gradient_magnitude": 1.0, "total_variation": 0.01},
               "iwe": {"method": "bilinear_vote", "blur_sigma": 1}}
    opt_cfg = {"n_iter": n_iter, "method": "Newton-CG", "max_iter": 25,
               "parameters": {"trans_x": {"min": -150, "max": 150}, "trans_y": {"min": -150, "max": 150}}}
    return solver.collections["pyramidal_patch_contrast_maximization"]((68, 90), {}, slv_cfg, opt_cfg, {}, None)


def test_patch_boxes_and_search_box_follow_the_reference(golden):
    """Patch bounds per scale against the reference's FlowPatch objects (fixture), and the re-initialisation box of
    sampling_initial (src/solver/patch_contrast_pyramid.py:417-428)."""
    from event_based_optical_flow_amd.solver.pyramid import search_box

    g = golden("patch_search")
    slv = _pyramid_solver()
    for s in (2, 3):
        np.testing.assert_array_equal(slv.patch_boxes(s), g[f"s{s}__boxes"])
        assert tuple(slv.scaled_patch_size[s]) == tuple(g[f"s{s}__patch_size"])
    lo, hi = search_box(np.array([100.0, -100.0, 2.0, 0.0]))
    np.testing.assert_allclose(lo, [80.0, -120.0, -8.0, -10.0])
    np.testing.assert_allclose(hi, [120.0, -80.0, 12.0, 10.0])


def test_pyramid_ops_and_feedback_keys():
    """pyramid_expand / pyramid_reduce (skimage restated on scipy.ndimage) keep constants and shapes; the coarse-from-fine
    feedback returns the reference's key set finest ... coarsest - 1 (src/solver/patch_contrast_pyramid.py:205-222) and
    reduces the OPTIMISED motion of the next finer scale."""
    from event_based_optical_flow_amd.solver import pyramid as P

    m = np.full((2, 4, 6), 3.5)
    up, down = P.pyramid_expand(m), P.pyramid_reduce(m)
    assert up.shape == (2, 8, 12) and down.shape == (2, 2, 3)
    np.testing.assert_allclose(up, 3.5, rtol=1e-12)
    np.testing.assert_allclose(down, 3.5, rtol=1e-12)
    ramp = np.stack([np.tile(np.arange(6.0), (4, 1)), np.tile(np.arange(4.0)[:, None], (1, 6))])
    up = P.pyramid_expand(ramp)
    assert np.all(np.diff(up[0], axis=1) >= -1e-12) and np.all(np.diff(up[1], axis=0) >= -1e-12)  # monotone stays monotone
    motions = {1: np.random.default_rng(0).normal(size=(2, 2, 2)), 2: np.random.default_rng(1).normal(size=(2, 4, 4)),
               3: np.random.default_rng(2).normal(size=(2, 8, 8))}
    fb = P.PyramidalPatchContrastMaximization.update_coarse_from_fine(None, motions)
    assert sorted(fb) == [0, 1, 2, 3]
    np.testing.assert_array_equal(fb[3], motions[3])
    np.testing.assert_allclose(fb[2], P.pyramid_reduce(motions[3]))
    np.testing.assert_allclose(fb[1], P.pyramid_reduce(motions[2]))  # the optimised scale-2 motion, not the reduced feedback
    np.testing.assert_allclose(fb[0], P.pyramid_reduce(motions[1]))


def test_grid_search_skips_nan_candidates(monkeypatch):
    """initialize_guess_from_whole_image keeps `best_guess` unless `loss < best_loss` (patch_contrast_base.py:164-187): a NaN candidate
    never wins, the FIRST minimum does, and a grid of NaNs leaves the reference's initial zeros(2)."""
    from event_based_optical_flow_amd.solver import translation_search as ts

    field = [-1.0, 0.0, 1.0]
    losses = np.array([3.0, np.nan, 2.0, 2.0, np.nan, 5.0, np.nan, np.nan, 2.5])
    monkeypatch.setattr(ts, "candidate_losses", lambda handle, cand, t_scale, **kw: losses.copy())
    best, table = ts.grid_search_translation(None, 1.0, field)
    assert np.array_equal(best, [-1.0, 1.0]) and table.shape == (3, 3) and np.isnan(table[0, 1])  # index 2: the first of the two 2.0
    monkeypatch.setattr(ts, "candidate_losses", lambda handle, cand, t_scale, **kw: np.full(9, np.nan))
    best, _ = ts.grid_search_translation(None, 1.0, field)
    assert np.array_equal(best, np.zeros(2))
    # a positive grid with one NaN: mapping NaN to 0.0 would have picked the NaN candidate
    monkeypatch.setattr(ts, "candidate_losses", lambda handle, cand, t_scale, **kw: np.where(np.arange(9) == 4, np.nan, 1.0 + np.arange(9)))
    best, _ = ts.grid_search_translation(None, 1.0, field)
    assert np.array_equal(best, [-1.0, -1.0])
