"""ctypes front-end of the CPU parity oracle (oracle/cmax_oracle.c).

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.  All arithmetic is fp64 and lives
in the C file (each C function cites the reference file:line it restates); this module only
marshals numpy arrays and composes the per-stage functions into one objective evaluation the
way the reference's `get_arg_for_cost` + `CostBase.calculate` do
(src/solver/patch_contrast_base.py:289-352, src/costs/*.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcmax_oracle.so")

_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_int_p = ctypes.POINTER(ctypes.c_int)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "cmax_oracle.c")
    if force or (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libcmax_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_reftime.restype = ctypes.c_double
        _lib.orc_variance.restype = ctypes.c_double
        _lib.orc_gradmag.restype = ctypes.c_double
        _lib.orc_total_variation.restype = ctypes.c_double
    return _lib


def _p(a):
    return a.ctypes.data_as(_c_double_p) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ev4(events):
    """Accept [n,2..4] arrays; pad to [n,4] (t=0, p=0) like the reference tests do implicitly."""
    ev = np.asarray(events, dtype=np.float64)
    assert ev.ndim == 2
    if ev.shape[1] < 4:
        ev = np.concatenate([ev, np.zeros((ev.shape[0], 4 - ev.shape[1]))], axis=1)
    return np.ascontiguousarray(ev)


# --------------------------------------------------------------------------------------------
# a1 / a2
# --------------------------------------------------------------------------------------------
_DIRECTION_FRAC = {"middle": 0.5, "before": -1.0, "after": 2.0}


def reftime(events, direction="first"):
    ev = _ev4(events)
    n = ev.shape[0]
    L = lib()
    if isinstance(direction, float):
        return L.orc_reftime(_p(ev), ctypes.c_longlong(n), 2, ctypes.c_double(direction))
    if direction == "first":
        return L.orc_reftime(_p(ev), ctypes.c_longlong(n), 0, ctypes.c_double(0.0))
    if direction == "last":
        return L.orc_reftime(_p(ev), ctypes.c_longlong(n), 1, ctypes.c_double(0.0))
    if direction in _DIRECTION_FRAC:
        return L.orc_reftime(_p(ev), ctypes.c_longlong(n), 2, ctypes.c_double(_DIRECTION_FRAC[direction]))
    raise ValueError(direction)


def calculate_dt(events, tref, normalize_t=True, period=None):
    ev = _ev4(events)
    n = ev.shape[0]
    dt = np.empty(n)
    lib().orc_calculate_dt(_p(ev), ctypes.c_longlong(n), ctypes.c_double(tref), int(bool(normalize_t)),
                           int(period is not None), ctypes.c_double(0.0 if period is None else period), _p(dt))
    return dt


# --------------------------------------------------------------------------------------------
# a3-a6 warps.  Returns (warped[n,4], aux) ; aux holds dt and (voxel) the per-event bin.
# --------------------------------------------------------------------------------------------
def voxel_from_sequential_burgers(flow, n_time_bin):
    """The flow voxel warp_event_from_optical_flow_voxel_optimized builds on the fly (src/warp.py:438-442): bin k uses the
    flow propagated k + 1 times by inviscid_burger_flow_to_voxel with delta_t = 1 / n_time_bin (the step comes BEFORE the use)."""
    f = _f64(flow)
    out = []
    for _ in range(int(n_time_bin)):
        f = burgers_step(f, 1.0 / n_time_bin)
        out.append(f)
    return np.stack(out)


def warp_event(events, motion, motion_model, direction="first", image_size=None, normalize_t=True, flow_propagate_bin=None):
    if motion_model == "dense-flow-voxel-optimized":  # a7: the same warp as "dense-flow-voxel" on the voxel built bin by bin
        return warp_event(events, voxel_from_sequential_burgers(motion, flow_propagate_bin), "dense-flow-voxel", direction, image_size,
                          normalize_t)
    ev = _ev4(events)
    n = ev.shape[0]
    tref = reftime(ev, direction)
    dt = calculate_dt(ev, tref, normalize_t)
    out = np.empty((n, 4))
    aux = {"dt": dt, "tref": tref}
    L = lib()
    m = _f64(motion)
    if motion_model in ("2d-translation", "rigid-optical-flow"):
        assert m.shape[-1] == 2
        L.orc_warp_2dof(_p(ev), ctypes.c_longlong(n), _p(m), _p(dt), _p(out))
    elif motion_model == "dense-flow":
        H, W = m.shape[-2:]
        L.orc_warp_dense(_p(ev), ctypes.c_longlong(n), _p(m), int(H), int(W), _p(dt), _p(out))
    elif motion_model == "dense-flow-voxel":
        T, _, H, W = m.shape
        bins = np.empty(n, dtype=np.int32)
        L.orc_warp_voxel(_p(ev), ctypes.c_longlong(n), _p(m), int(T), int(H), int(W), _p(dt), _p(out),
                         bins.ctypes.data_as(_c_int_p))
        aux["bin"] = bins
    else:
        raise KeyError(motion_model)
    return out, aux


# --------------------------------------------------------------------------------------------
# a10-a12 image formation
# --------------------------------------------------------------------------------------------
def _pad2(outer_padding):
    if isinstance(outer_padding, (int, float)):
        return int(outer_padding), int(outer_padding)
    return int(outer_padding[0]), int(outer_padding[1])


def vote(events, image_size, outer_padding=0, weight=1.0, eps=1e-6, method="bilinear_vote"):
    """image_size is the UN-padded (H, W); the returned image is (H+2ph, W+2pw)
    (src/event_image_converter.py:28)."""
    ev = _f64(events)
    n, stride = ev.shape
    ph, pw = _pad2(outer_padding)
    Hp, Wp = int(image_size[0]) + 2 * ph, int(image_size[1]) + 2 * pw
    img = np.empty((Hp, Wp))
    warr = _f64(weight) if isinstance(weight, np.ndarray) else None
    wsc = 1.0 if warr is not None else float(weight)
    lib().orc_vote(_p(ev), ctypes.c_longlong(stride), ctypes.c_longlong(n), _p(warr), ctypes.c_double(wsc),
                   Hp, Wp, ph, pw, ctypes.c_double(eps), int(method == "count"), _p(img))
    return img


def vote_bwd(events, image_size, G, outer_padding=0, weight=1.0, eps=1e-6, want_gw=False):
    ev = _f64(events)
    n, stride = ev.shape
    ph, pw = _pad2(outer_padding)
    Hp, Wp = int(image_size[0]) + 2 * ph, int(image_size[1]) + 2 * pw
    G = _f64(G)
    assert G.shape == (Hp, Wp)
    gx, gy = np.empty(n), np.empty(n)
    gw = np.empty(n) if want_gw else None
    warr = _f64(weight) if isinstance(weight, np.ndarray) else None
    wsc = 1.0 if warr is not None else float(weight)
    lib().orc_vote_bwd(_p(ev), ctypes.c_longlong(stride), ctypes.c_longlong(n), _p(warr), ctypes.c_double(wsc),
                       Hp, Wp, ph, pw, ctypes.c_double(eps), _p(G), _p(gx), _p(gy), _p(gw))
    return (gx, gy, gw) if want_gw else (gx, gy)


def blur3(img, sigma):
    img = _f64(img)
    out = np.empty_like(img)
    lib().orc_blur3(_p(img), img.shape[0], img.shape[1], ctypes.c_double(sigma), _p(out))
    return out


def blur3_adj(g, sigma):
    g = _f64(g)
    out = np.empty_like(g)
    lib().orc_blur3_adj(_p(g), g.shape[0], g.shape[1], ctypes.c_double(sigma), _p(out))
    return out


def create_iwe(events, image_size, outer_padding=0, method="bilinear_vote", sigma=1, weight=1.0, eps=1e-6):
    """EventImageConverter.create_iwe, torch branch (src/event_image_converter.py:45-67,126-159)."""
    img = vote(events, image_size, outer_padding, weight, eps, method)
    if sigma > 0:
        img = blur3(img, sigma)
    return img


# --------------------------------------------------------------------------------------------
# a14-a16 costs: raw value + d value / d image
# --------------------------------------------------------------------------------------------
def variance(img, omit_boundary=True, ddof=1, want_grad=True):
    img = _f64(img)
    G = np.empty_like(img) if want_grad else None
    v = lib().orc_variance(_p(img), img.shape[0], img.shape[1], int(bool(omit_boundary)), int(ddof), _p(G))
    return v, G


def gradmag(img, omit_boundary=True, want_grad=True):
    img = _f64(img)
    G = np.empty_like(img) if want_grad else None
    v = lib().orc_gradmag(_p(img), img.shape[0], img.shape[1], int(bool(omit_boundary)), _p(G))
    return v, G


def total_variation(flow, omit_boundary=True, want_grad=True):
    flow = _f64(flow)
    assert flow.ndim == 3 and flow.shape[0] == 2
    G = np.empty_like(flow) if want_grad else None
    v = lib().orc_total_variation(_p(flow), flow.shape[1], flow.shape[2], int(bool(omit_boundary)), _p(G))
    return v, G


# --------------------------------------------------------------------------------------------
# a8 / a9 time-aware flow
# --------------------------------------------------------------------------------------------
def burgers_step(flow, dt):
    f = _f64(flow)
    out = np.empty_like(f)
    lib().orc_burgers_step(_p(f), f.shape[1], f.shape[2], ctypes.c_double(dt), _p(out))
    return out


def burgers_step_adj(flow, dt, gout):
    f, g = _f64(flow), _f64(gout)
    gF = np.zeros_like(f)
    lib().orc_burgers_step_adj(_p(f), f.shape[1], f.shape[2], ctypes.c_double(dt), _p(g), _p(gF))
    return gF


def upwind_step(flow, dt):
    f = _f64(flow)
    out = np.empty_like(f)
    lib().orc_upwind_step(_p(f), f.shape[1], f.shape[2], ctypes.c_double(dt), _p(out))
    return out


def upwind_step_adj(flow, dt, gout):
    f, g = _f64(flow), _f64(gout)
    gF = np.zeros_like(f)
    lib().orc_upwind_step_adj(_p(f), f.shape[1], f.shape[2], ctypes.c_double(dt), _p(g), _p(gF))
    return gF


_SCHEME = {"burgers": 0, "upwind": 1}


def _t0_index(time_bin, t0_location):
    if t0_location == "first":
        return 0
    if t0_location == "middle":
        return time_bin // 2
    raise NotImplementedError(t0_location)


def construct_dense_flow_voxel(flow, time_bin, scheme="upwind", t0_location="middle"):
    f = _f64(flow)
    _, H, W = f.shape
    V = np.zeros((time_bin, 2, H, W))
    lib().orc_voxel_construct(_p(f), int(time_bin), _t0_index(time_bin, t0_location), H, W, _SCHEME[scheme], _p(V))
    return V


def construct_dense_flow_voxel_adj(voxel, gvoxel, scheme="upwind", t0_location="middle"):
    V = _f64(voxel)
    gV = _f64(gvoxel).copy()
    T, _, H, W = V.shape
    gF = np.empty((2, H, W))
    lib().orc_voxel_construct_adj(_p(V), int(T), _t0_index(T, t0_location), H, W, _SCHEME[scheme], _p(gV), _p(gF))
    return gF


# --------------------------------------------------------------------------------------------
# a17 chain rule to motion parameters
# --------------------------------------------------------------------------------------------
def motion_grad(events, motion, motion_model, aux, gx, gy):
    ev = _ev4(events)
    n = ev.shape[0]
    m = np.asarray(motion)
    dt = aux["dt"]
    L = lib()
    if motion_model in ("2d-translation", "rigid-optical-flow"):
        g = np.empty(2)
        L.orc_grad_2dof(_p(dt), _p(gx), _p(gy), ctypes.c_longlong(n), _p(g))
        return g
    if motion_model == "dense-flow":
        H, W = m.shape[-2:]
        g = np.empty((2, H, W))
        L.orc_grad_dense(_p(ev), _p(dt), _p(gx), _p(gy), ctypes.c_longlong(n), int(H), int(W), _p(g))
        return g
    if motion_model == "dense-flow-voxel":
        T, _, H, W = m.shape
        g = np.empty((T, 2, H, W))
        L.orc_grad_voxel(_p(ev), _p(dt), aux["bin"].ctypes.data_as(_c_int_p), _p(gx), _p(gy),
                         ctypes.c_longlong(n), int(T), int(H), int(W), _p(g))
        return g
    raise KeyError(motion_model)


# --------------------------------------------------------------------------------------------
# One objective evaluation = what get_arg_for_cost + cost.calculate + autograd.grad compute
# (src/solver/patch_contrast_base.py:273-352; src/solver/scipy_autograd/torch_wrapper.py:30-49)
# --------------------------------------------------------------------------------------------
_BASE = {"image_variance": variance, "gradient_magnitude": gradmag}


def _base_cost(kind, img, omit):
    """raw contrast + gradient image for kind in {'var','gm'} (torch branch: unbiased var)."""
    v, G = variance(img, omit, 1) if kind == "var" else gradmag(img, omit)
    return np.float64(v), G  # IEEE division like the reference's tensors: x/0 = inf, 0/0 = nan (no ZeroDivisionError)


def cost_and_image_grads(cost, iwes, omit_boundary=True, direction="minimize", cost_with_weight=None, flow=None):
    with np.errstate(divide="ignore", invalid="ignore"):
        return _cost_and_image_grads(cost, iwes, omit_boundary, direction, cost_with_weight, flow)


def _cost_and_image_grads(cost, iwes, omit_boundary=True, direction="minimize", cost_with_weight=None, flow=None):
    """Evaluate a named cost on a dict of IWEs.  Returns (loss, {key: dL/d iwe_key}, dL/d flow or None).

    cost: one of image_variance, gradient_magnitude, normalized_image_variance,
    normalized_gradient_magnitude, multi_focal_normalized_image_variance,
    multi_focal_normalized_gradient_magnitude, total_variation, hybrid.
    Semantics: src/costs/*.py (torch branches), direction handling per class.
    """
    grads = {}
    gflow = None

    def add(key, g):
        grads[key] = grads.get(key, 0) + g

    if cost == "hybrid":
        loss = 0.0
        for name, wgt in cost_with_weight.items():
            l, g, gf = cost_and_image_grads(name, iwes, omit_boundary, direction, None, flow)
            if wgt == "inv":  # src/costs/hybrid.py:51-53
                loss += 1.0 / l
                scale = -1.0 / (l * l)
            else:
                loss += wgt * l
                scale = wgt
            for k, v in g.items():
                add(k, scale * v)
            if gf is not None:
                gflow = (gflow if gflow is not None else 0) + scale * gf
        return loss, grads, gflow

    if cost in ("image_variance", "gradient_magnitude"):
        kind = "var" if cost == "image_variance" else "gm"
        v, G = _base_cost(kind, iwes["iwe"], omit_boundary)
        s = -1.0 if direction == "minimize" else 1.0  # image_variance.py:56-58
        add("iwe", s * G)
        return s * v, grads, None

    if cost in ("normalized_image_variance", "normalized_gradient_magnitude"):
        kind = "var" if cost == "normalized_image_variance" else "gm"
        # normalized_image_variance.py:40-41 crops ONLY iwe; orig_iwe stays un-cropped for the variance;
        # normalized_gradient_magnitude.py:63-79 applies omit_boundary to both.
        v1, G1 = _base_cost(kind, iwes["iwe"], omit_boundary)
        v2, _ = _base_cost(kind, iwes["orig_iwe"], omit_boundary if kind == "gm" else False)
        if direction == "minimize":
            add("iwe", -v2 / (v1 * v1) * G1)
            return v2 / v1, grads, None
        add("iwe", G1 / v2)
        return v1 / v2, grads, None

    if cost in ("multi_focal_normalized_image_variance", "multi_focal_normalized_gradient_magnitude"):
        kind = "var" if cost.endswith("image_variance") else "gm"
        v2, _ = _base_cost(kind, iwes["orig_iwe"], omit_boundary if kind == "gm" else False)
        loss = 0.0
        for key, mult in (("forward_iwe", 1.0), ("backward_iwe", 1.0), ("middle_iwe", 2.0)):
            if key not in iwes:
                continue
            v1, G1 = _base_cost(kind, iwes[key], omit_boundary)
            if direction == "minimize":
                loss += mult * v2 / v1
                add(key, mult * (-v2 / (v1 * v1)) * G1)
            else:  # inner normalised cost returns iwe/orig for natural/maximize
                loss += mult * v1 / v2
                add(key, mult * G1 / v2)
        if direction == "maximize":  # multi_focal_*.py: "-loss" for maximize
            loss = -loss
            grads = {k: -v for k, v in grads.items()}
        return loss, grads, None

    if cost == "total_variation":
        v, G = total_variation(flow, omit_boundary)
        if direction == "minimize":
            return v, grads, G
        return -v, grads, -G

    raise KeyError(cost)


_REQUIRED = {
    "image_variance": ["iwe"],
    "gradient_magnitude": ["iwe"],
    "normalized_image_variance": ["orig_iwe", "iwe"],
    "normalized_gradient_magnitude": ["orig_iwe", "iwe"],
    "multi_focal_normalized_image_variance": ["forward_iwe", "backward_iwe", "middle_iwe", "orig_iwe"],
    "multi_focal_normalized_gradient_magnitude": ["forward_iwe", "backward_iwe", "middle_iwe", "orig_iwe"],
    "total_variation": ["flow"],
}
_KEY_DIRECTION = {"iwe": "first", "backward_iwe": "first", "forward_iwe": "last", "middle_iwe": "middle"}


def required_keys(cost, cost_with_weight=None):
    if cost == "hybrid":
        keys = []
        for name in cost_with_weight:
            keys.extend(_REQUIRED[name])
        return keys
    return _REQUIRED[cost]


def objective(events, motion, motion_model, image_size, cost="image_variance", sigma=0, outer_padding=0,
              iwe_method="bilinear_vote", omit_boundary=True, direction="minimize", cost_with_weight=None,
              coarse_flow=None, normalize_t=True, want_grad=True, warp_direction="first"):
    """One evaluation of the reference objective; returns dict(loss, grad, iwes, image_grads, grad_flow).

    warp_direction: reference time of the "iwe" key -- "first" in get_arg_for_cost (patch_contrast_base.py:310-312); any direction
    Warp.calculate_reftime takes (src/warp.py:201-233: names or a float) for callers of the leaf API that warp elsewhere.

    `motion` is theta[2] / flow[2,H,W] / voxel[T,2,H,W] (already in pixel per normalised time, i.e.
    what `calculate_cost` receives).  `coarse_flow` is the patch-flow array handed to
    total_variation (patch_contrast_base.py:349-350).
    """
    ev = _ev4(events)
    keys = required_keys(cost, cost_with_weight)
    iwes, ctx = {}, {}
    if "orig_iwe" in keys:
        iwes["orig_iwe"] = create_iwe(ev, image_size, outer_padding, iwe_method, sigma)
    need = [k for k in ("iwe", "backward_iwe", "forward_iwe", "middle_iwe") if k in keys]
    if "iwe" in need or "backward_iwe" in need:
        need = [k for k in need if k not in ("iwe", "backward_iwe")] + ["iwe"]
    for key in need:
        warped, aux = warp_event(ev, motion, motion_model, warp_direction if key == "iwe" else _KEY_DIRECTION[key], image_size, normalize_t)
        img = create_iwe(warped, image_size, outer_padding, iwe_method, sigma)
        ctx[key] = (warped, aux)
        iwes[key] = img
        if key == "iwe":
            iwes["backward_iwe"] = img
    loss, image_grads, grad_flow = cost_and_image_grads(cost, iwes, omit_boundary, direction, cost_with_weight, coarse_flow)
    out = {"loss": loss, "iwes": iwes, "image_grads": image_grads, "grad_flow": grad_flow, "grad": None}
    if not want_grad:
        return out
    total = None
    # iwe and backward_iwe alias the same tensor in the reference: gradients add
    merged = {}
    for k, g in image_grads.items():
        kk = "iwe" if k == "backward_iwe" else k
        merged[kk] = merged.get(kk, 0) + g
    for key, G in merged.items():
        if key not in ctx:
            continue
        warped, aux = ctx[key]
        if sigma > 0:
            G = blur3_adj(G, sigma)
        if iwe_method == "count":
            continue
        gx, gy = vote_bwd(warped, image_size, G, outer_padding)
        g = motion_grad(ev, motion, motion_model, aux, gx, gy)
        total = g if total is None else total + g
    if total is None:
        total = np.zeros_like(np.asarray(motion, dtype=np.float64))
    out["grad"] = total
    return out


# --------------------------------------------------------------------------------------------
# next-row 1: patch grid -> dense flow (src/solver/patch_contrast_base.py:462-506)
# --------------------------------------------------------------------------------------------
def patch_pad(patch_size, sliding_window, patch_shift=(0, 0)):
    """pad_h, pad_w of interpolate_dense_flow_from_patch_tensor (lines 470-479)."""
    return tuple(int(patch_size[k] / 2 // sliding_window[k]) + patch_shift[k] // sliding_window[k] + 1 for k in range(2))


def patch_to_dense(motion, image_size, sliding_window, pad):
    m = _f64(motion)
    _, ph, pw = m.shape
    H, W = int(image_size[0]), int(image_size[1])
    flow = np.empty((2, H, W))
    lib().orc_patch_to_dense(_p(m), ph, pw, int(pad[0]), int(pad[1]), int(sliding_window[0]), int(sliding_window[1]), H, W, _p(flow))
    return flow


def patch_to_dense_adj(gflow, patch_image_size, sliding_window, pad):
    g = _f64(gflow)
    _, H, W = g.shape
    ph, pw = int(patch_image_size[0]), int(patch_image_size[1])
    gm = np.empty((2, ph, pw))
    lib().orc_patch_to_dense_adj(_p(g), ph, pw, int(pad[0]), int(pad[1]), int(sliding_window[0]), int(sliding_window[1]), H, W, _p(gm))
    return gm


def solver_objective(events, x, image_size, patch_image_size, patch_size, sliding_window, patch_shift,
                     cost="hybrid", cost_with_weight=None, sigma=1, time_aware=False, time_bin=10,
                     flow_interpolation="burgers", t0_flow_location="middle"):
    """PyramidalPatchContrastMaximization.objective_scipy (src/solver/patch_contrast_pyramid.py:430-462,
    464-516) for one scale: x[2*ph*pw] (pixel / time unit) -> (loss, dloss/dx)."""
    ev = _ev4(events)
    ph, pw = int(patch_image_size[0]), int(patch_image_size[1])
    t_scale = ev[:, 2].max() - ev[:, 2].min()  # line 444
    pad = patch_pad(patch_size, sliding_window, patch_shift)
    motion = np.asarray(x, dtype=np.float64).reshape(2, ph, pw)
    dense = patch_to_dense(motion, image_size, sliding_window, pad)  # pixel / time unit
    if time_aware:
        # construct(dense * t_scale / scale) * scale / t_scale, scale = 1 (lines 499-515); then * t_scale (452)
        voxel = construct_dense_flow_voxel(dense * t_scale, time_bin, flow_interpolation, t0_flow_location)
        res = objective(ev, voxel, "dense-flow-voxel", image_size, cost=cost, sigma=sigma,
                        cost_with_weight=cost_with_weight, coarse_flow=motion)
        g_dense = construct_dense_flow_voxel_adj(voxel, res["grad"], flow_interpolation, t0_flow_location) * t_scale
    else:
        res = objective(ev, dense * t_scale, "dense-flow", image_size, cost=cost, sigma=sigma,
                        cost_with_weight=cost_with_weight, coarse_flow=motion)
        g_dense = res["grad"] * t_scale
    g = patch_to_dense_adj(g_dense, (ph, pw), sliding_window, pad)
    if res["grad_flow"] is not None:
        g = g + res["grad_flow"]
    return res["loss"], g.reshape(-1)


def gaussian_filter(img, sigma):
    """numpy-branch blur: scipy.ndimage.gaussian_filter restated (src/event_image_converter.py:122-124)."""
    img = _f64(img)
    out = np.empty_like(img)
    lib().orc_gaussian_filter(_p(img), img.shape[0], img.shape[1], ctypes.c_double(sigma), _p(out))
    return out


# --------------------------------------------------------------------------------------------
# per-patch translation search (pyramid re-initialisation at scales above the coarsest)
# --------------------------------------------------------------------------------------------
def gradmag_reflect101(img):
    """GradientMagnitude.calculate_numpy with omit_boundary False (src/costs/gradient_magnitude.py:78-95), un-signed:
    mean(gx^2 + gy^2), gx = cv2.Sobel(img, CV_64F, 1, 0, ksize=3) / 8 (derivative along columns), gy = Sobel(0, 1) / 8,
    OpenCV's default border BORDER_REFLECT_101 (numpy's pad mode 'reflect')."""
    img = _f64(img)
    p = np.pad(img, 1, mode="reflect") if min(img.shape) > 1 else np.pad(img, 1, mode="edge")
    gx = ((p[:-2, 2:] - p[:-2, :-2]) + 2.0 * (p[1:-1, 2:] - p[1:-1, :-2]) + (p[2:, 2:] - p[2:, :-2])) / 8.0
    gy = ((p[2:, :-2] - p[:-2, :-2]) + 2.0 * (p[2:, 1:-1] - p[:-2, 1:-1]) + (p[2:, 2:] - p[:-2, 2:])) / 8.0
    return float(np.mean(gx * gx + gy * gy))


def crop_event(events, x0, x1, y0, y1):
    """utils.crop_event (src/utils/event_utils.py:50-70)."""
    ev = np.asarray(events)
    m = (x0 <= ev[:, 0]) & (ev[:, 0] < x1) & (y0 <= ev[:, 1]) & (ev[:, 1] < y1)
    return ev[m]


def small_patch_gm(events, box, patch_image_size, theta, sigma):
    """GM of one patch image for one translation: calculate_cost_for_small_patch's `iwe` leg
    (src/solver/patch_contrast_pyramid.py:372-414) with objective_initial's scaling (355-370): the candidate is
    multiplied by the patch's time span and the warper divides dt by the same span, so the displacement is
    theta * (t - t_mid).  theta None: the un-warped image (`orig_iwe`)."""
    x0, x1, y0, y1 = (int(v) for v in box)
    ev = np.array(crop_event(_ev4(events), x0, x1, y0, y1), dtype=np.float64)
    ev[:, 0] -= x0  # set_event_origin_to_zero (event_utils.py:73-90)
    ev[:, 1] -= y0
    if theta is not None and len(ev):
        t_scale = ev[:, 2].max() - ev[:, 2].min()
        ev, _ = warp_event(ev, np.asarray(theta, dtype=np.float64) * t_scale, "2d-translation", direction="middle", normalize_t=True)
    img = vote(ev, patch_image_size, 0, 1.0, eps=1e-8)  # bilinear_vote_numpy floors x + 1e-8
    if sigma > 0:
        img = gaussian_filter(img, sigma)
    return gradmag_reflect101(img), len(ev)


def patch_search(events, boxes, patch_image_size, candidates, sigma):
    """-> (loss [n_patch, n_cand] = GM(orig) / GM(warped), gm [n_patch, n_cand + 1], count [n_patch])."""
    boxes = np.asarray(boxes).reshape(-1, 4)
    cands = np.asarray(candidates, dtype=np.float64).reshape(len(boxes), -1, 2)
    gm = np.zeros((len(boxes), cands.shape[1] + 1))
    count = np.zeros(len(boxes), dtype=np.int64)
    for p, box in enumerate(boxes):
        for c in range(cands.shape[1]):
            gm[p, c], count[p] = small_patch_gm(events, box, patch_image_size, cands[p, c], sigma)
        gm[p, -1], count[p] = small_patch_gm(events, box, patch_image_size, None, sigma)
    with np.errstate(divide="ignore", invalid="ignore"):
        loss = gm[:, -1:] / gm[:, :-1]
    return loss, gm, count
