"""GPU parity of the FUSED objective (cmax_set_events / cmax_objective: fp32 per event, fp64
reductions) against the reference's fp64 values.

Tolerance (BASELINE.json north_star, SURVEY.md section 8d "parity gate"), judged against fp64:
    IWE      max|I - I_ref| / max|I_ref|        <= 1e-4
    loss     |L - L_ref| / |L_ref|              <= 1e-4
    gradient max|g - g_ref| / max|g_ref|        <= 1e-4
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import event_based_optical_flow_amd as E  # noqa: E402
from oracle import oracle as orc  # noqa: E402

TOL = 1e-4
DEV = "cuda"


def rel_max(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def T(a, dtype=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=DEV)


def fused_eval(size, events, motion, model, cost, sigma, pad=0, time_bin=0, cost_with_weight=None, coarse=None,
               direction="minimize"):
    h = E.CMaxHandle(size, pad)
    h.set_events(events, time_bin=time_bin)
    obj = E.ContrastObjective(h, model, cost=cost, cost_with_weight=cost_with_weight, sigma=sigma, direction=direction)
    m = T(motion).requires_grad_()
    c = T(coarse).requires_grad_() if coarse is not None else None
    loss = obj(m, c)
    ins = [m] + ([c] if c is not None and cost_with_weight and "total_variation" in cost_with_weight else [])
    grads = torch.autograd.grad(loss, ins)
    return loss.item(), [g.cpu().numpy() for g in grads], h


YAML_HYBRID = {"multi_focal_normalized_gradient_magnitude": 1.0, "total_variation": 0.01}
OBJ_CASES = [(c, s) for c in ("image_variance", "gradient_magnitude") for s in (0, 1)] + [
    (c, 1) for c in ("normalized_image_variance", "normalized_gradient_magnitude",
                     "multi_focal_normalized_image_variance", "multi_focal_normalized_gradient_magnitude", "hybrid")]
MOTIONS = {"2dof": ("2d-translation", "theta"), "dense_rand": ("dense-flow", "flow_rand"),
           "dense_smooth": ("dense-flow", "flow_smooth"), "voxel": ("dense-flow-voxel", "voxel")}


@pytest.mark.parametrize("mname", list(MOTIONS))
@pytest.mark.parametrize("cost,sigma", OBJ_CASES)
def test_fused_objective_golden(golden, mname, cost, sigma):
    """Fixtures hold the REFERENCE's loss / gradient / IWEs for the same inputs."""
    g = golden("objective")
    model, mkey = MOTIONS[mname]
    size = tuple(int(v) for v in g["image_size"])
    tb = g[mkey].shape[0] if model == "dense-flow-voxel" else 0
    loss, grads, h = fused_eval(size, g["events"], g[mkey], model, cost, sigma, time_bin=tb,
                                cost_with_weight=YAML_HYBRID if cost == "hybrid" else None, coarse=g["coarse"])
    tag = f"{mname}__{cost}__s{sigma}"
    assert abs(loss - g[tag + "__loss"]) <= TOL * abs(g[tag + "__loss"])
    assert rel_max(grads[0], g[tag + "__grad"]) <= TOL
    if len(grads) > 1:
        assert rel_max(grads[1], g[tag + "__grad_coarse"]) <= TOL
    if tag + "__iwe" in g and cost == "image_variance":
        assert rel_max(h.last_iwe(0).cpu().numpy(), g[tag + "__iwe"]) <= TOL
    if tag + "__forward_iwe" in g:  # multi-focal: slot 0 = forward ("last"), 1 = backward ("first"), 2 = middle
        assert rel_max(h.last_iwe(0).cpu().numpy(), g[tag + "__forward_iwe"]) <= TOL
        assert rel_max(h.last_iwe(2).cpu().numpy(), g[tag + "__middle_iwe"]) <= TOL


@pytest.mark.parametrize("pad", [0, 4])
def test_fused_fractional_sources_and_padding(golden, pad):
    g = golden("objective")
    size = tuple(int(v) for v in g["image_size"])
    loss, grads, h = fused_eval(size, g["events_frac"], g["theta"], "2d-translation", "image_variance", 1, pad=pad)
    assert abs(loss - g[f"frac_pad{pad}__loss"]) <= TOL * abs(g[f"frac_pad{pad}__loss"])
    assert rel_max(grads[0], g[f"frac_pad{pad}__grad"]) <= TOL
    assert rel_max(h.last_iwe(0).cpu().numpy(), g[f"frac_pad{pad}__iwe"]) <= TOL


@pytest.mark.parametrize("direction", ["natural", "maximize"])
@pytest.mark.parametrize("cost", ["image_variance", "normalized_gradient_magnitude", "multi_focal_normalized_image_variance"])
def test_fused_directions_vs_oracle(golden, direction, cost):
    g = golden("objective")
    size = tuple(int(v) for v in g["image_size"])
    ref = orc.objective(g["events"], g["flow_smooth"], "dense-flow", size, cost=cost, sigma=1, direction=direction)
    loss, grads, _ = fused_eval(size, g["events"], g["flow_smooth"], "dense-flow", cost, 1, direction=direction)
    assert abs(loss - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_max(grads[0], ref["grad"]) <= TOL


# ---------------------------------------------------------------------------------------------
# BASELINE configs at sizes the oracle finishes in seconds (seeded inputs, fp64 oracle)
# ---------------------------------------------------------------------------------------------
def _structured(n, size, vel, seed):
    return E.utils.generate_structured_events(n, size[0], size[1], vel, n_dots=max(50, n // 400), seed=seed)


def test_cfg2_2dof_variance_structured():
    """cfg2 shape: 260x346, 2-DoF, variance; structured (moving-dot) events, theta 20 % off the optimum."""
    size, n = (260, 346), 300_000
    vel = (12.3, -7.7)
    ev = _structured(n, size, vel, 46)
    theta = np.array(vel) * 0.8
    ref = orc.objective(ev, theta, "2d-translation", size, cost="image_variance", sigma=0)
    loss, grads, h = fused_eval(size, ev, theta, "2d-translation", "image_variance", 0)
    assert rel_max(h.last_iwe(0).cpu().numpy(), ref["iwes"]["iwe"]) <= TOL
    assert abs(loss - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_max(grads[0], ref["grad"]) <= TOL


def test_cfg2_2dof_variance_uniform_random():
    """Documented worst case (SURVEY.md section 7 hard part 2): uniform-random events, the gradient
    is a heavily cancelling sum.  The integer-base/displacement split keeps it inside 1e-4 (the bench workload
    itself, 1M events, is tests/test_gpu_fullsize.py::test_cfg2_full_size_bench_workload)."""
    size, n = (260, 346), 300_000
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=46)
    theta = np.array([12.3, -7.7])
    ref = orc.objective(ev, theta, "2d-translation", size, cost="image_variance", sigma=0)
    loss, grads, h = fused_eval(size, ev, theta, "2d-translation", "image_variance", 0)
    assert rel_max(h.last_iwe(0).cpu().numpy(), ref["iwes"]["iwe"]) <= TOL
    assert abs(loss - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_max(grads[0], ref["grad"]) <= TOL, (grads[0], ref["grad"])


def test_cfg3_dense_gradmag():
    size, n = (120, 160), 400_000  # DSEC aspect, reduced so the oracle finishes in seconds
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=47)
    flow = E.utils.generate_smooth_flow(size, 20, seed=48)
    ref = orc.objective(ev, flow, "dense-flow", size, cost="gradient_magnitude", sigma=0)
    loss, grads, h = fused_eval(size, ev, flow, "dense-flow", "gradient_magnitude", 0)
    assert rel_max(h.last_iwe(0).cpu().numpy(), ref["iwes"]["iwe"]) <= TOL
    assert abs(loss - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_max(grads[0], ref["grad"]) <= TOL


@pytest.mark.parametrize("n,cost,sigma", [(150_000, "image_variance", 0), (150_000, "image_variance", 1), (9_000, "gradient_magnitude", 0)])
def test_dense_gradient_run_reduction_regimes(n, cost, sigma):
    """The dense flow gradient sums runs of equal source pixel: serially per thread when a pixel holds >= 8
    events on average (150k events on 60x80 = 31 per pixel), with a per-slot segmented scan otherwise (9k = 1.9
    per pixel).  Hot pixels (every 7th event lands on one of 5 pixels) give runs that span many threads and waves."""
    size = (60, 80)
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=61)
    hot = np.arange(0, n, 7)
    ev[hot, 0] = 10 + (hot % 5)
    ev[hot, 1] = 20
    flow = E.utils.generate_smooth_flow(size, 12, seed=62)
    ref = orc.objective(ev, flow, "dense-flow", size, cost=cost, sigma=sigma)
    loss, grads, h = fused_eval(size, ev, flow, "dense-flow", cost, sigma)
    assert abs(loss - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_max(grads[0], ref["grad"]) <= TOL


@pytest.mark.parametrize("model,cost,sigma", [("2d-translation", "image_variance", 0), ("2d-translation", "multi_focal_normalized_image_variance", 0),
                                              ("dense-flow", "image_variance", 0), ("dense-flow", "gradient_magnitude", 1),
                                              ("dense-flow", "multi_focal_normalized_gradient_magnitude", 1),
                                              ("dense-flow", "image_variance", 1)])
def test_odd_image_size_and_repeated_evaluations(model, cost, sigma):
    """37 x 45 = 1665 pixels: not a multiple of 4, so the 16-byte write-through clearing of the next vote image / of the
    flow gradient falls back to 4-byte stores and the per-reference-time images are not 16-byte aligned.  Five
    evaluations on one handle (alternating vote buffers, cached un-warped image) must all match the oracle."""
    size, n = (37, 45), 12_000
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=81)
    h = E.CMaxHandle(size).set_events(ev)
    obj = E.ContrastObjective(h, model, cost=cost, sigma=sigma)
    rng = np.random.default_rng(82)
    for it in range(5):
        motion = rng.uniform(-8, 8, 2) if model == "2d-translation" else E.utils.generate_smooth_flow(size, 6, seed=90 + it)
        ref = orc.objective(ev, motion, model, size, cost=cost, sigma=sigma)
        m = T(motion).requires_grad_()
        loss = obj(m)
        (grad,) = torch.autograd.grad(loss, m)
        assert abs(loss.item() - ref["loss"]) <= TOL * abs(ref["loss"]), it
        assert rel_max(grad.cpu().numpy(), ref["grad"]) <= TOL, it


def test_cfg4_burgers_voxel_variance():
    size, n, Tn = (130, 173), 300_000, 10
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=49)
    f0 = E.utils.generate_smooth_flow(size, 20, seed=50)
    voxel = orc.construct_dense_flow_voxel(f0 / 20.0, Tn, "burgers", "middle") * 20.0
    ref = orc.objective(ev, voxel, "dense-flow-voxel", size, cost="image_variance", sigma=1)
    loss, grads, h = fused_eval(size, ev, voxel, "dense-flow-voxel", "image_variance", 1, time_bin=Tn)
    assert rel_max(h.last_iwe(0).cpu().numpy(), ref["iwes"]["iwe"]) <= TOL
    assert abs(loss - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_max(grads[0], ref["grad"]) <= TOL


def test_voxel_chain_to_t0_flow():
    """flow_t0 -> Burgers voxel (HIP, autograd) -> fused voxel objective: gradient w.r.t. flow_t0."""
    size, n, Tn = (64, 80), 50_000, 10
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=51)
    f0 = E.utils.generate_smooth_flow(size, 10, seed=52)
    V = orc.construct_dense_flow_voxel(f0, Tn, "burgers", "middle")
    ref = orc.objective(ev, V, "dense-flow-voxel", size, cost="image_variance", sigma=1)
    g_f0 = orc.construct_dense_flow_voxel_adj(V, ref["grad"], "burgers", "middle")
    h = E.CMaxHandle(size).set_events(ev, time_bin=Tn)
    obj = E.ContrastObjective(h, "dense-flow-voxel", cost="image_variance", sigma=1)
    tf = T(f0).requires_grad_()
    loss = obj(E.utils.construct_dense_flow_voxel_torch(tf, Tn, "burgers", "middle"))
    (g,) = torch.autograd.grad(loss, tf)
    assert abs(loss.item() - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_max(g.cpu().numpy(), g_f0) <= TOL


# ---------------------------------------------------------------------------------------------
# size-independent properties at BASELINE's full sizes
# ---------------------------------------------------------------------------------------------
def test_full_size_mass_and_linearity():
    """cfg2 full size (1M events, 260x346): (i) with zero motion every event votes weight 1 in
    bounds -> sum(IWE) == N exactly representable; (ii) IWE(A u B) == IWE(A) + IWE(B)."""
    size, n = (260, 346), 1_000_000
    ev = E.utils.generate_events(n, size[0] - 1, size[1] - 1, 0.0, 0.05, seed=46)
    h = E.CMaxHandle(size).set_events(ev)
    assert h.n_events == n
    iwe0 = h.iwe(np.zeros(2), "2d-translation")
    assert abs(float(iwe0.double().sum()) - n) < 1e-6 * n
    theta = np.array([12.3, -7.7])
    full = h.iwe(theta, "2d-translation").double()
    tmin, tmax = ev[:, 2].min(), ev[:, 2].max()
    a = E.CMaxHandle(size).set_events(ev[: n // 3], tmin, tmax).iwe(theta, "2d-translation").double()
    b = E.CMaxHandle(size).set_events(ev[n // 3:], tmin, tmax).iwe(theta, "2d-translation").double()
    assert float((a + b - full).abs().max()) <= 1e-5 * float(full.abs().max())


def test_full_size_gradient_vs_finite_difference():
    """cfg2 full size: analytic 2-DoF gradient against a central difference of the loss itself."""
    size, n = (260, 346), 1_000_000
    ev = _structured(n, size, (12.3, -7.7), 46)
    h = E.CMaxHandle(size).set_events(ev)
    desc = E.make_descriptor("image_variance", "2d-translation", sigma=1.0)
    theta = np.array([10.0, -6.0])
    res, grad = h.evaluate(desc, theta)
    g = grad.cpu().numpy()
    eps = 0.05
    for c in range(2):
        d = np.zeros(2)
        d[c] = eps
        lp = h.evaluate(desc, theta + d, want_grad=False)[0][0].item()
        lm = h.evaluate(desc, theta - d, want_grad=False)[0][0].item()
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - g[c]) <= 2e-2 * max(abs(g).max(), 1e-12), (c, fd, g)


def test_cfg5_shape_time_slices_sum_to_whole():
    """cfg5 shape (1280x720, dense flow, variance): per-time-slice IWEs and gradients (what each GPU
    of the time-sliced run owns) add up to the single-handle result."""
    size, n, parts = (720, 1280), 2_000_000, 4
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=53)
    flow = T(E.utils.generate_smooth_flow(size, 20, seed=54), torch.float32)
    tmin, tmax = ev[:, 2].min(), ev[:, 2].max()
    whole = E.CMaxHandle(size).set_events(ev)
    iwe = whole.iwe(flow, "dense-flow").double()
    acc = torch.zeros_like(iwe)
    for sl in np.array_split(np.arange(n), parts):
        acc += E.CMaxHandle(size).set_events(ev[sl], tmin, tmax).iwe(flow, "dense-flow").double()
    assert float((acc - iwe).abs().max()) <= 1e-5 * float(iwe.abs().max())


# ---------------------------------------------------------------------------------------------
# edge cases (failure handling)
# ---------------------------------------------------------------------------------------------
def test_empty_batch_gives_zero_loss_and_grad():
    h = E.CMaxHandle((32, 40)).set_events(np.zeros((0, 4)))
    desc = E.make_descriptor("image_variance", "dense-flow")
    res, grad = h.evaluate(desc, np.zeros((2, 32, 40)))
    assert res[0].item() == 0.0 and float(grad.abs().sum()) == 0.0


def test_out_of_sensor_sources_are_dropped_not_crashing():
    ev = E.utils.generate_events(1000, 32, 40, seed=1)
    ev[::10, 0] = -5.0
    ev[5::10, 1] = 1e9
    ev[7, 0] = np.nan
    h = E.CMaxHandle((32, 40)).set_keep_outside(False).set_events(ev)  # (asked to drop: the default keeps finite off-sensor events)
    keep = np.isfinite(ev[:, 0]) & (ev[:, 0] >= 0) & (ev[:, 0] < 32) & (ev[:, 1] >= 0) & (ev[:, 1] < 40)
    assert h.n_events == int(keep.sum())
    iwe = h.iwe(np.zeros(2), "2d-translation")
    assert abs(float(iwe.sum()) - keep.sum()) < 1e-3


@pytest.mark.parametrize("pad", [0, 6])
@pytest.mark.parametrize("cost,sigma", [("image_variance", 0), ("gradient_magnitude", 1), ("normalized_image_variance", 0)])
def test_events_off_the_sensor_vote_like_the_reference(pad, cost, sigma):
    """cmax_set_keep_outside: the reference's 2-DoF warp has no bounds test on the source (src/warp.py:506-515) -- an event from outside
    the sensor votes wherever it warps into the padded image.  A third of this batch starts up to 25 px outside; theta brings part of
    it in.  Against the oracle (which follows the reference): IWE, loss, gradient; NaN / absurd coordinates are still dropped; dense
    objectives refuse such a batch."""
    size, n = (64, 80), 60_000
    rng = np.random.default_rng(91)
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=90)
    out = rng.random(n) < 0.33
    ev[out, 0] = rng.uniform(-25.0, size[0] + 25.0, int(out.sum()))
    ev[out, 1] = rng.uniform(-25.0, size[1] + 25.0, int(out.sum()))
    ev[::7, 0] = np.floor(ev[::7, 0])  # integer and fractional coordinates mixed
    theta = np.array([17.0, -21.0])
    ref = orc.objective(ev, theta, "2d-translation", size, cost=cost, sigma=sigma, outer_padding=pad)
    bad = ev.copy()
    bad = np.concatenate([bad, np.array([[np.nan, 3.0, 0.01, 1.0], [1e12, 3.0, 0.01, 1.0]])])  # dropped either way
    h = E.CMaxHandle(size, outer_padding=pad).set_keep_outside(True).set_events(bad, on_dropped="ignore")
    info = h.batch_info()
    off = (np.floor(ev[:, 0]) < 0) | (np.floor(ev[:, 0]) >= size[0]) | (np.floor(ev[:, 1]) < 0) | (np.floor(ev[:, 1]) >= size[1])
    assert info["packed"] == n and info["dropped"] == 2 and info["outside"] == int(off.sum()) and info["fractional"]
    desc = E.make_descriptor(cost, "2d-translation", sigma=float(sigma))
    res, grad = h.evaluate(desc, theta)
    assert rel_max(h.last_iwe(0).cpu().numpy(), ref["iwes"]["iwe"]) <= TOL
    assert abs(res[0].item() - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_max(grad.cpu().numpy(), ref["grad"]) <= TOL
    with pytest.raises(E._lib.CmaxError):
        h.evaluate(E.make_descriptor(cost, "dense-flow", sigma=float(sigma)), np.zeros((2,) + size, np.float32))
    # default handles drop them, as before
    h2 = E.CMaxHandle(size, outer_padding=pad).set_keep_outside(False).set_events(ev, on_dropped="ignore")
    assert h2.batch_info()["dropped"] == int(off.sum()) and h2.batch_info()["outside"] == 0


@pytest.mark.parametrize("pad", [0, 6])
@pytest.mark.parametrize("cost,sigma", [("image_variance", 0), ("gradient_magnitude", 1), ("normalized_image_variance", 1)])
def test_events_off_the_sensor_golden(golden, pad, cost, sigma):
    """... and against the reference itself (tests/golden/outside_sensor.npz)."""
    g = golden("outside_sensor")
    size = tuple(int(v) for v in g["image_size"])
    tag = f"pad{pad}__{cost}__s{sigma}"
    h = E.CMaxHandle(size, outer_padding=pad).set_keep_outside(True).set_events(g["events"])
    assert h.batch_info()["dropped"] == 0 and h.batch_info()["outside"] > 1000
    res, grad = h.evaluate(E.make_descriptor(cost, "2d-translation", sigma=float(sigma)), g["theta"])
    assert rel_max(h.last_iwe(0).cpu().numpy(), g[tag + "__iwe"]) <= TOL
    assert abs(res[0].item() - float(g[tag + "__loss"])) <= TOL * abs(float(g[tag + "__loss"]))
    assert rel_max(grad.cpu().numpy(), g[tag + "__grad"]) <= TOL


def test_everything_warps_out_of_the_image():
    ev = E.utils.generate_events(2000, 32, 40, seed=2)
    h = E.CMaxHandle((32, 40)).set_events(ev)
    iwe = h.iwe(np.array([1e5, 1e5]), "2d-translation", direction="middle")
    assert float(iwe.abs().sum()) < 40.0  # only events with dt ~ 0 stay


def test_bad_arguments_raise():
    h = E.CMaxHandle((32, 40)).set_events(E.utils.generate_events(100, 32, 40, seed=3))
    with pytest.raises(KeyError):
        E.make_descriptor("zhu_average_timestamp", "dense-flow")
    with pytest.raises(KeyError):
        E.make_descriptor("image_variance", "affine")
    with pytest.raises(ValueError):
        E.make_descriptor("image_variance", "dense-flow", direction="sideways")
    with pytest.raises(E._lib.CmaxError):  # voxel objective without time bins on the handle
        h.evaluate(E.make_descriptor("image_variance", "dense-flow-voxel", time_bin=10), np.zeros((10, 2, 32, 40)))
    with pytest.raises(E._lib.CmaxError):
        E.CMaxHandle((5000, 40))
    with pytest.raises(ValueError):
        h.set_events(np.zeros((10, 3)))


@pytest.mark.parametrize("model", ["2d-translation", "dense-flow"])
def test_large_displacements_clip_the_lds_window(model):
    """Displacements far beyond the LDS window (theta = 150 px over the batch, the optimiser's search
    range in configs/*.yaml): votes / gradient reads outside the window take the global-memory path."""
    size, n = (130, 173), 120_000
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=61)
    if model == "2d-translation":
        motion = np.array([150.0, -140.0])
    else:
        motion = E.utils.generate_smooth_flow(size, 150, grid=3, seed=62)
    ref = orc.objective(ev, motion, model, size, cost="image_variance", sigma=0)
    loss, grads, h = fused_eval(size, ev, motion, model, "image_variance", 0)
    assert rel_max(h.last_iwe(0).cpu().numpy(), ref["iwes"]["iwe"]) <= TOL
    assert abs(loss - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_max(grads[0], ref["grad"]) <= TOL


@pytest.mark.parametrize("model", ["2d-translation", "dense-flow"])
@pytest.mark.parametrize("slabs", [2, 4, 7])
def test_time_slabs_large_displacement_parity(model, slabs):
    """cmax_set_time_slabs: the batch in time slabs, slab-major inside a tile row -- a segment's window spans 1 / slabs of the
    displacement range.  150 px over the batch (the search range of configs/*.yaml) against the oracle, every cost family that the
    2-DoF / dense models take; and back to the un-binned order."""
    size, n = (130, 173), 300_000
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=63)
    if model == "2d-translation":
        motion = np.array([150.0, -100.0])
    else:
        motion = E.utils.generate_smooth_flow(size, 150, grid=3, seed=64)
    h = E.CMaxHandle(size).set_events(ev)
    segs0 = h.work_list_info()["segments"]
    h.set_time_slabs(slabs)
    assert h.work_list_info()["segments"] >= segs0
    for cost, sigma in (("image_variance", 0), ("gradient_magnitude", 1)):
        ref = orc.objective(ev, motion, model, size, cost=cost, sigma=sigma)
        desc = E.make_descriptor(cost, model, sigma=float(sigma))
        res, grad = h.evaluate(desc, motion)
        assert rel_max(h.last_iwe(0).cpu().numpy(), ref["iwes"]["iwe"]) <= TOL
        assert abs(res[0].item() - ref["loss"]) <= TOL * abs(ref["loss"])
        assert rel_max(grad.cpu().numpy(), ref["grad"]) <= TOL
    with pytest.raises(E._lib.CmaxError):  # voxel motions need time BINS
        h.evaluate(E.make_descriptor("image_variance", "dense-flow-voxel", time_bin=slabs), np.zeros((slabs, 2) + size, np.float32))
    assert h.suggest_time_slabs(150.0) == 4 and h.suggest_time_slabs(30.0) == 2 and h.suggest_time_slabs(10.0) == 1
    h.set_time_slabs(0)
    assert h.work_list_info()["segments"] == segs0
    res0, grad0 = h.evaluate(desc, motion)
    assert abs(res0[0].item() - res[0].item()) <= 1e-6 * abs(res[0].item())


def test_repeated_evaluations_are_consistent():
    """The handle double-buffers its vote images (K2 of evaluation e zeroes the images of e+1): many
    evaluations with changing costs / motions must keep giving the single-shot answer."""
    size = (64, 80)
    ev = E.utils.generate_events(30_000, size[0], size[1], 0.0, 0.05, seed=63)
    h = E.CMaxHandle(size).set_events(ev)
    rng = np.random.default_rng(0)
    descs = [E.make_descriptor("image_variance", "2d-translation"),
             E.make_descriptor("multi_focal_normalized_gradient_magnitude", "2d-translation", sigma=1.0),
             E.make_descriptor("normalized_image_variance", "2d-translation")]
    for it in range(12):
        d = descs[it % 3]
        theta = rng.uniform(-20, 20, 2)
        res, grad = h.evaluate(d, theta)
        fresh = E.CMaxHandle(size).set_events(ev)
        res2, grad2 = fresh.evaluate(d, theta)
        assert abs(res[0].item() - res2[0].item()) <= 1e-6 * abs(res2[0].item())
        assert rel_max(grad.cpu().numpy(), grad2.cpu().numpy()) <= 1e-5


@pytest.mark.parametrize("model", ["2d-translation", "dense-flow", "dense-flow-voxel"])
def test_wide_workgroup_path_equals_time_sliced_sum(model):
    """Above ~2.1M events per handle the event kernels switch to 512-thread workgroups (4 events per
    thread; the voxel K3 to 1024 x 2).  One 2.6M-event handle (wide path) must equal two 1.3M-event time slices
    (256-thread path) combined through the phase-split API exactly like the multi-GPU run does."""
    size, n = (260, 346), 2_600_000
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=71)
    Tn = 6 if model == "dense-flow-voxel" else 0
    if model == "2d-translation":
        motion = np.array([12.3, -7.7])
    elif model == "dense-flow":
        motion = E.utils.generate_smooth_flow(size, 15, seed=72)
    else:
        motion = np.stack([E.utils.generate_smooth_flow(size, 15, seed=72 + b) for b in range(Tn)])
    desc = E.make_descriptor("image_variance", model, sigma=1.0, time_bin=Tn)
    whole = E.CMaxHandle(size).set_events(ev, time_bin=Tn)
    res, grad = whole.evaluate(desc, motion)
    tmin, tmax = ev[:, 2].min(), ev[:, 2].max()
    halves = [E.CMaxHandle(size).set_events(ev[: n // 2], tmin, tmax, time_bin=Tn),
              E.CMaxHandle(size).set_events(ev[n // 2:], tmin, tmax, time_bin=Tn)]
    images = sum(h.objective_vote(desc, motion) for h in halves)
    outs = [h.objective_finish(desc, motion, images) for h in halves]
    assert abs(outs[0][0][0].item() - res[0].item()) <= 1e-6 * abs(res[0].item())
    gsum = (outs[0][1].double() + outs[1][1].double()).cpu().numpy()
    assert rel_max(gsum, grad.double().cpu().numpy()) <= 2e-5


def test_voxel_many_time_bins_and_rebinning():
    """T = 20 bins: one 16x16 tile needs 5120 accumulator cells > the 3072 held in LDS, so part of the
    flow-gradient runs take the direct global-atomic path; and cmax_set_time_bins re-bins a packed batch."""
    size, n, Tn = (64, 80), 60_000, 20
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=81)
    voxel = orc.construct_dense_flow_voxel(E.utils.generate_smooth_flow(size, 8, seed=82), Tn, "burgers", "middle")
    ref = orc.objective(ev, voxel, "dense-flow-voxel", size, cost="image_variance", sigma=1)
    h = E.CMaxHandle(size).set_events(ev)  # packed without bins ...
    h.set_time_bins(Tn)  # ... re-binned afterwards
    desc = E.make_descriptor("image_variance", "dense-flow-voxel", sigma=1.0, time_bin=Tn)
    res, grad = h.evaluate(desc, voxel)
    assert abs(res[0].item() - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_max(grad.cpu().numpy(), ref["grad"]) <= TOL
    h2 = E.CMaxHandle(size).set_events(ev, time_bin=Tn)
    res2, grad2 = h2.evaluate(desc, voxel)
    assert rel_max(grad2.cpu().numpy(), grad.cpu().numpy()) <= 1e-5


@pytest.mark.parametrize("direction", ["before", "after", 0.3, "last"])
def test_reference_time_variants(direction):
    """Warp.calculate_reftime's other directions (src/warp.py:216-233) through the fused path."""
    size, n = (48, 64), 40_000
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=83)
    flow = E.utils.generate_smooth_flow(size, 6, seed=84)
    warped, _ = orc.warp_event(ev, flow, "dense-flow", direction, size)
    iwe_ref = orc.create_iwe(warped, size, sigma=0)
    h = E.CMaxHandle(size).set_events(ev)
    iwe = h.iwe(flow, "dense-flow", direction=direction).cpu().numpy()
    assert rel_max(iwe, iwe_ref) <= TOL


def test_unnormalised_time():
    """normalize_t = False: dt in the events' own time unit (src/warp.py:254-259)."""
    size, n = (48, 64), 40_000
    ev = E.utils.generate_events(n, size[0], size[1], 1.0, 3.5, seed=85)  # period 2.5 time units
    theta = np.array([3.0, -2.0])
    warped, _ = orc.warp_event(ev, theta, "2d-translation", "middle", size, normalize_t=False)
    iwe_ref = orc.create_iwe(warped, size, sigma=1)
    h = E.CMaxHandle(size).set_events(ev)
    iwe = h.iwe(theta, "2d-translation", direction="middle", normalize_t=False, sigma=1.0).cpu().numpy()
    assert rel_max(iwe, iwe_ref) <= TOL


def test_dense_with_padding_and_fractional_sources():
    size, pad, n = (40, 56), 5, 30_000
    rng = np.random.default_rng(86)
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=86)
    ev[:, 0] = np.clip(ev[:, 0] + rng.uniform(0, 0.99, n), 0, size[0] - 1e-3)
    ev[:, 1] = np.clip(ev[:, 1] + rng.uniform(0, 0.99, n), 0, size[1] - 1e-3)
    flow = E.utils.generate_smooth_flow(size, 12, seed=87)
    ref = orc.objective(ev, flow, "dense-flow", size, cost="gradient_magnitude", sigma=1, outer_padding=pad)
    loss, grads, h = fused_eval(size, ev, flow, "dense-flow", "gradient_magnitude", 1, pad=pad)
    assert rel_max(h.last_iwe(0).cpu().numpy(), ref["iwes"]["iwe"]) <= TOL
    assert abs(loss - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_max(grads[0], ref["grad"]) <= TOL


def test_dropped_events_are_counted_and_reported(caplog):
    """cmax_set_events drops events whose source pixel is off the sensor or NaN; the count is exposed
    (cmax_batch_info) and logged.  The surviving events give the oracle's result for the same survivors."""
    import logging

    size, n = (48, 64), 20_000
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=91)
    ev[5, 0] = -3.0
    ev[77, 1] = size[1] + 2.0
    ev[1234, 0] = np.nan
    with caplog.at_level(logging.WARNING):
        h = E.CMaxHandle(size).set_keep_outside(False).set_events(ev)
    info = h.batch_info()
    assert info["dropped"] == 3 and info["packed"] == n - 3 and not info["fractional"]
    assert any("dropped 3" in r.message for r in caplog.records)
    keep = np.ones(n, dtype=bool)
    keep[[5, 77, 1234]] = False
    theta = np.array([4.0, -6.0])
    res, grad = h.evaluate(E.make_descriptor("image_variance", "2d-translation"), theta)
    ref = orc.objective(ev[keep], theta, "2d-translation", size, cost="image_variance", sigma=0)
    # t_min / t_max of the batch are taken over ALL events (dropped ones included), like the reference's warp would
    assert abs(res[0].item() - ref["loss"]) <= 1e-3 * abs(ref["loss"])
    h.set_events(ev[keep])
    assert h.batch_info()["dropped"] == 0
    # a solver that must not diverge from the reference silently asks for an error instead
    with pytest.raises(ValueError, match="dropped 3"):
        E.CMaxHandle(size).set_keep_outside(False).set_events(ev, on_dropped="raise")
    with pytest.raises(ValueError):
        E.CMaxHandle(size).set_events(ev, on_dropped="sometimes")
    # THE DEFAULT (round 5) follows the reference's 2-DoF warp, which has no bounds test on the source (src/warp.py:506-515): the two
    # finite events are kept (they vote if theta brings them in), only the NaN is dropped; a dense objective refuses such a batch
    hk = E.CMaxHandle(size).set_events(ev, on_dropped="ignore")
    assert hk.batch_info()["dropped"] == 1 and hk.batch_info()["outside"] == 2 and hk.batch_info()["packed"] == n - 1
    resk, gradk = hk.evaluate(E.make_descriptor("image_variance", "2d-translation"), theta)
    fin = np.isfinite(ev[:, 0])
    refk = orc.objective(ev[fin], theta, "2d-translation", size, cost="image_variance", sigma=0)
    assert abs(resk[0].item() - refk["loss"]) <= 1e-4 * abs(refk["loss"])
    assert np.abs(gradk.cpu().numpy() - refk["grad"]).max() <= 1e-4 * np.abs(refk["grad"]).max()
    with pytest.raises(E._lib.CmaxError, match="off the sensor"):
        hk.evaluate(E.make_descriptor("image_variance", "dense-flow"), np.zeros((2,) + size))


def test_prepared_call_equals_evaluate_and_follows_the_motion_buffer():
    """CMaxHandle.prepare: one allocation, many evaluations of a motion buffer that is updated in place."""
    size = (60, 80)
    ev = E.utils.generate_events(20000, size[0], size[1], 0.0, 0.05, seed=12)
    h = E.CMaxHandle(size).set_events(ev)
    desc = E.make_descriptor("image_variance", "2d-translation")
    theta = torch.tensor([4.0, -3.0], dtype=torch.float32, device="cuda")
    call, res, grad = h.prepare(desc, theta)
    for t in ([4.0, -3.0], [-7.5, 2.25]):
        theta.copy_(torch.tensor(t, dtype=torch.float32))
        call()
        r2, g2 = h.evaluate(desc, theta)
        ref = orc.objective(ev, np.array(t), "2d-translation", size, cost="image_variance", sigma=0)
        assert abs(res[0].item() - ref["loss"]) <= TOL * abs(ref["loss"])
        assert rel_max(grad.cpu().numpy(), ref["grad"]) <= TOL
        assert abs(res[0].item() - r2[0].item()) <= 1e-6 * abs(r2[0].item()) and rel_max(grad.cpu().numpy(), g2.cpu().numpy()) <= 1e-5


# ---------------------------------------------------------------------------------------------
# round 3: results on the host (cmax_objective_host) and the raw form of the 2-DoF variance objective
# (cmax_objective_raw + cmax_finalize_raw_host: no finishing kernel, the consumer folds 32 x 6 partial sums)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_ref_cost", ["image_variance", "multi_focal_normalized_image_variance"])
@pytest.mark.parametrize("pad,frac", [(0, False), (3, True)])
def test_two_dof_raw_and_host_forms(n_ref_cost, pad, frac):
    size, n = (120, 160), 200_000
    ev = E.utils.generate_structured_events(n, size[0], size[1], (9.0, -6.0), n_dots=300, seed=21)
    if frac:
        rng = np.random.default_rng(21)
        ev[:, 0] = np.clip(ev[:, 0] + rng.uniform(0, 0.99, n), 0, size[0] - 1e-3)
        ev[:, 1] = np.clip(ev[:, 1] + rng.uniform(0, 0.99, n), 0, size[1] - 1e-3)
    theta = np.array([8.0, -5.0])
    h = E.CMaxHandle(size, pad).set_events(ev)
    desc = E.make_descriptor(n_ref_cost, "2d-translation")
    ref = orc.objective(ev, theta, "2d-translation", size, cost=n_ref_cost, sigma=0, outer_padding=pad)
    res_d, grad_d = h.evaluate(desc, theta)  # device results (k_finish_raw behind K3)
    assert abs(res_d[0].item() - ref["loss"]) <= TOL * abs(ref["loss"]) and rel_max(grad_d.cpu().numpy(), ref["grad"]) <= TOL
    for rep in range(3):  # host results, repeated: the raw sums are cleared inside every evaluation
        res_h, grad_h = h.evaluate_host(desc, theta)
        assert abs(res_h[0] - ref["loss"]) <= TOL * abs(ref["loss"]), (rep, res_h[0], ref["loss"])
        assert rel_max(grad_h, ref["grad"]) <= TOL
    if n_ref_cost == "image_variance":
        assert h.has_raw(desc)
        call, raw, finalize = h.prepare_raw(desc, theta)
        for rep in range(3):
            call()
        res_r, grad_r = finalize()
        assert abs(res_r[0] - ref["loss"]) <= TOL * abs(ref["loss"]) and rel_max(grad_r, ref["grad"]) <= TOL
        assert abs(res_r[1] - (-ref["loss"])) <= TOL * abs(ref["loss"])  # result[1] = the raw contrast
        # interleaved with the device-result form on the same handle (they share the vote images, not the sums)
        call()
        res_d2, grad_d2 = h.evaluate(desc, theta)
        res_r2, grad_r2 = finalize()
        assert abs(res_r2[0] - res_d2[0].item()) <= 1e-6 * abs(res_r2[0]) and rel_max(grad_r2, grad_d2.cpu().numpy()) <= 1e-5
    else:
        assert not h.has_raw(desc)  # normalised: needs the un-warped image's statistics on the device
        with pytest.raises(E._lib.CmaxError):
            h.prepare_raw(desc, theta)


@pytest.mark.parametrize("cost,sigma", [("gradient_magnitude", 0.0), ("image_variance", 1.0), ("multi_focal_normalized_gradient_magnitude", 1.0),
                                        ("normalized_image_variance", 1.0)])
def test_two_dof_raw_form_of_objectives_with_statistics(cost, sigma):
    """Every 2-DoF objective of the default mode has a raw form: K3 adds sum dt g into 32 lines and its first workgroup writes the
    result into the spare doubles of the first line; the consumer folds (no k_finish launch).  Device, host and raw forms agree with
    each other and with the oracle; repeated, and interleaved on one handle."""
    size, n = (120, 160), 200_000
    ev = E.utils.generate_structured_events(n, size[0], size[1], (9.0, -6.0), n_dots=300, seed=22)
    theta = np.array([8.0, -5.0])
    h = E.CMaxHandle(size).set_events(ev)
    desc = E.make_descriptor(cost, "2d-translation", sigma=sigma)
    ref = orc.objective(ev, theta, "2d-translation", size, cost=cost, sigma=int(sigma))
    assert h.has_raw(desc)
    call, raw, finalize = h.prepare_raw(desc, theta)
    for rep in range(3):
        call()
        res_r, grad_r = finalize()
        assert abs(res_r[0] - ref["loss"]) <= TOL * abs(ref["loss"]), (rep, res_r[0], ref["loss"])
        assert rel_max(grad_r, ref["grad"]) <= TOL
    res_d, grad_d = h.evaluate(desc, theta)
    res_h, grad_h = h.evaluate_host(desc, theta)
    call()
    res_r, grad_r = finalize()
    for res, grad in ((res_d.cpu().numpy(), grad_d.cpu().numpy()), (res_h, grad_h)):
        assert abs(res[0] - res_r[0]) <= 1e-6 * abs(res_r[0]) and rel_max(grad, grad_r) <= 1e-5
    for k in range(1, 1 + desc.n_ref):  # the per-reference-time contrasts ride along
        assert abs(res_r[k] - res_d[k].item()) <= 1e-6 * abs(res_r[k])


@pytest.mark.parametrize("model,cost,sigma", [("dense-flow", "gradient_magnitude", 1.0), ("dense-flow-voxel", "image_variance", 0.0),
                                               ("2d-translation", "gradient_magnitude", 0.0)])
def test_objective_host_other_models(model, cost, sigma):
    """cmax_objective_host for objectives without a raw form: the same numbers as cmax_objective, on the host."""
    size, n, Tn = (96, 128), 120_000, 4
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=23)
    if model == "2d-translation":
        motion, tb = np.array([7.0, -4.0]), 0
    elif model == "dense-flow":
        motion, tb = E.utils.generate_smooth_flow(size, 10, seed=24), 0
    else:
        motion, tb = np.stack([E.utils.generate_smooth_flow(size, 10, seed=24 + b) for b in range(Tn)]), Tn
    h = E.CMaxHandle(size).set_events(ev, time_bin=tb)
    desc = E.make_descriptor(cost, model, sigma=sigma, time_bin=tb)
    ref = orc.objective(ev, motion, model, size, cost=cost, sigma=int(sigma))
    for want_grad in (True, False, True):
        res, grad = h.evaluate_host(desc, motion, want_grad=want_grad)
        assert abs(res[0] - ref["loss"]) <= TOL * abs(ref["loss"])
        if want_grad:
            assert rel_max(grad, ref["grad"]) <= TOL
    if model == "2d-translation":
        assert h.has_raw(desc)
    else:
        with pytest.raises(E._lib.CmaxError):
            h.prepare_raw(desc, motion)


def test_exact_cells_fp64_theta_beats_fp32_rounding():
    """warp_one's exact-cell branch: with theta handed over in fp64 the 2-DoF gradient on UNIFORM events (every one of
    the 600k a potential cell-border event) meets the plain gate against the fp64 oracle, for all three reference times."""
    size, n = (260, 346), 600_000
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=31)
    theta = np.array([17.123456789, -11.987654321])
    h = E.CMaxHandle(size).set_events(ev)
    for direction in ("first", "middle", "last"):
        desc = E.make_descriptor("image_variance", "2d-translation", warp_direction=direction)
        res, grad = h.evaluate(desc, theta)
        warped, _ = orc.warp_event(ev, theta, "2d-translation", direction, size)
        iwe_ref = orc.create_iwe(warped, size, sigma=0)
        assert rel_max(h.last_iwe(0).cpu().numpy(), iwe_ref) <= TOL
        if direction == "first":
            ref = orc.objective(ev, theta, "2d-translation", size, cost="image_variance", sigma=0)
            e = rel_max(grad.cpu().numpy(), ref["grad"])
            print(f"[exact cells] {direction}: gradient rel err {e:.2e}")
            assert e <= TOL and abs(res[0].item() - ref["loss"]) <= TOL * abs(ref["loss"])


# ---- round 4: K candidate motions per launch pair (cmax_objective_batch) ------------------------------------------------------
def test_objective_batch_eight_thetas_in_one_launch_pair():
    """VERDICT r3 #6.  Eight DISTINCT 2-DoF motions through cmax_objective_batch -- blockIdx.z of K1 / K3 is the candidate, each with
    its own vote image, raw-sum lines and windows, one finishing wave per candidate -- against the oracle and against eight single
    cmax_objective calls; twice (the handle double-buffers the batch's vote images), then with another K, and with an fp32 batch."""
    size, n = (180, 240), 300_000
    ev = E.utils.generate_structured_events(n, size[0], size[1], (11.0, -6.0), n_dots=900, seed=8)
    h = E.CMaxHandle(size).set_events(ev)
    desc = E.make_descriptor("image_variance", "2d-translation")
    rng = np.random.default_rng(5)
    thetas = np.concatenate([[[11.0, -6.0], [0.0, 0.0], [80.0, 55.0]], rng.uniform(-25, 25, (5, 2))])
    refs = [orc.objective(ev, t, "2d-translation", size, cost="image_variance", sigma=0) for t in thetas]
    for rep in range(2):
        res, grad = h.evaluate_batch(desc, thetas)
        res, grad = res.cpu().numpy(), grad.cpu().numpy()
        for k, ref in enumerate(refs):
            assert abs(res[k, 0] - ref["loss"]) <= TOL * abs(ref["loss"]), (rep, k)
            assert np.abs(grad[k] - ref["grad"]).max() <= TOL * np.abs(ref["grad"]).max(), (rep, k, grad[k], ref["grad"])
    for k, t in enumerate(thetas):  # the same numbers one candidate at a time
        r1, g1 = h.evaluate(desc, t)
        assert abs(r1[0].item() - res[k, 0]) <= 1e-6 * abs(res[k, 0])
        assert np.abs(g1.cpu().numpy() - grad[k]).max() <= 2e-5 * np.abs(grad[k]).max()
    res3, grad3 = h.evaluate_batch(desc, thetas[:3])  # fewer candidates ...
    res12, grad12 = h.evaluate_batch(desc, np.concatenate([thetas, thetas[:4]]))  # ... and more than before: buffers grow, stale images are cleared
    assert np.abs(res3.cpu().numpy()[:, 0] - res[:3, 0]).max() <= 1e-6 * np.abs(res[:3, 0]).max()
    assert np.abs(res12.cpu().numpy()[:8, 0] - res[:, 0]).max() <= 1e-6 * np.abs(res[:, 0]).max()
    assert np.abs(res12.cpu().numpy()[8:, 0] - res[:4, 0]).max() <= 1e-6 * np.abs(res[:4, 0]).max()
    assert np.abs(grad12.cpu().numpy()[8:] - grad[:4]).max() <= 2e-5 * np.abs(grad[:4]).max()
    res32, grad32 = h.evaluate_batch(desc, thetas.astype(np.float32))  # fp32 thetas
    assert np.abs(res32.cpu().numpy()[:, 0] - res[:, 0]).max() <= 2e-5 * np.abs(res[:, 0]).max()


def test_objective_batch_falls_back_for_other_objectives():
    """Objectives outside the fast path (here: dense flow with blur, gradient magnitude) are evaluated candidate by candidate inside the
    call: same results as single calls."""
    size, n = (96, 128), 60_000
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=9)
    h = E.CMaxHandle(size).set_events(ev)
    desc = E.make_descriptor("gradient_magnitude", "dense-flow", sigma=1.0)
    flows = np.stack([E.utils.generate_smooth_flow(size, 10, seed=20 + k) for k in range(3)])
    res, grad = h.evaluate_batch(desc, flows)
    for k in range(3):
        r1, g1 = h.evaluate(desc, flows[k])
        assert abs(r1[0].item() - res[k, 0].item()) <= 1e-6 * abs(r1[0].item())
        assert rel_max(grad[k].cpu().numpy(), g1.cpu().numpy()) <= 2e-5


def test_rebinning_keeps_the_batch_counts_and_set_events_leaves_slab_order():
    """ADVICE r4.  (1) cmax_set_time_bins / cmax_set_time_slabs re-sort the PACKED events: the counts cmax_set_events took from the raw
    input (events dropped, events kept from off the sensor) must survive, or the guards of the dense / voxel objectives are bypassed and
    such events are warped as if they sat on the nearest sensor pixel.  (2) the slab order belongs to the batch that was re-ordered: a
    later cmax_set_events on the same handle starts un-slabbed (voxel objectives and the patch search work again)."""
    size, n = (64, 96), 40_000
    rng = np.random.default_rng(5)
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=5)
    ev[:300, 0] = -rng.uniform(1.0, 9.0, 300)  # 300 events above the sensor
    ev[300:303, 1] = np.nan                    # three unusable ones
    h = E.CMaxHandle(size).set_keep_outside(True).set_events(ev, on_dropped="ignore")
    info0 = h.batch_info()
    assert info0["outside"] == 300 and info0["dropped"] == 3
    flow = E.utils.generate_smooth_flow(size, 3, seed=6)
    dense = E.make_descriptor("image_variance", "dense-flow")
    for rebin in (lambda: h.set_time_bins(4), lambda: h.set_time_slabs(2), lambda: h.set_time_bins(0)):
        rebin()
        info = h.batch_info()
        assert info["outside"] == 300 and info["dropped"] == 3 and info["packed"] == n - 3, info
        with pytest.raises(E._lib.CmaxError):  # a flow has no value off the sensor: refused, not clamped
            h.evaluate(dense, flow)
    # (2) slabs, then a new batch with time bins on the same handle: voxel objectives must work
    h2 = E.CMaxHandle(size).set_events(ev[303:])
    h2.set_time_slabs(2)
    ev2 = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=7)
    h2.set_events(ev2, time_bin=4)
    voxel = np.stack([flow * (1.0 + 0.1 * k) for k in range(4)])
    res, grad = h2.evaluate(E.make_descriptor("image_variance", "dense-flow-voxel", time_bin=4), voxel)
    ref = orc.objective(ev2, np.asarray(voxel, np.float32).astype(np.float64), "dense-flow-voxel", size, cost="image_variance", sigma=0)
    assert abs(res[0].item() - ref["loss"]) <= 1e-4 * abs(ref["loss"])
    assert np.abs(grad.double().cpu().numpy() - ref["grad"]).max() <= 1e-4 * np.abs(ref["grad"]).max()
    h2.set_time_slabs(2)
    h2.set_events(ev2)  # un-binned batch after a slabbed one: the patch search walks tile-major groups again
    loss, gm, count = h2.patch_search(np.array([[0, 32, 0, 48]]), (32, 48), np.zeros((1, 1, 2)), 1.0)
    assert int(count[0].item()) > 0


@pytest.mark.parametrize("size,n,bins", [((64, 96), 50_000, 0), ((260, 346), 400_000, 0), ((48, 64), 600_000, 0), ((64, 96), 60_000, 4),
                                         ((720, 1280), 9_000_000, 0), ((260, 346), 8_500_000, 10)])
def test_packed_order(size, n, bins):
    """The order cmax_set_events leaves: source tile (16 x 16) major; un-binned handles by pixel inside a tile and BY
    TIME inside a pixel (k_run_time_sort: runs of up to 192 events -- the (48, 64) case holds ~195 per pixel, so both branches run);
    binned handles by (tile, bin).  Every input event appears exactly once with its pixel and its normalised time.  The two cases of
    8.5M / 9M events take the STABLE RADIX SORT (cmax_radix_sort.h, batches >= 8M events): every run by time whatever its length.
    (CMAX_SORT=radix python -m pytest tests -m gpu runs the whole suite on that pipeline.)"""
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=3)
    h = E.CMaxHandle(size).set_events(ev, time_bin=bins)
    packed, gs = h.packed_events()
    assert packed.shape[0] == n and gs[0] == 0 and gs[-1] == n and (np.diff(gs) >= 0).all()
    row, col = packed[:, 0] & 0xFFF, (packed[:, 0] >> 12) & 0xFFF
    tau = packed[:, 1].astype(np.uint32).view(np.float32)
    # the same multiset of (pixel, time) as the input
    tn = ((ev[:, 2] - ev[:, 2].min()) / (ev[:, 2].max() - ev[:, 2].min())).astype(np.float32)
    key_in = np.lexsort((tn, ev[:, 1].astype(np.int64), ev[:, 0].astype(np.int64)))
    key_out = np.lexsort((tau, col, row))
    np.testing.assert_array_equal(row[key_out], ev[key_in, 0].astype(np.int64))
    np.testing.assert_array_equal(col[key_out], ev[key_in, 1].astype(np.int64))
    np.testing.assert_array_equal(tau[key_out], tn[key_in])
    # tile-major groups
    ntc = (size[1] + 15) // 16
    tile = (row >> 4) * ntc + (col >> 4)
    group = np.searchsorted(gs, np.arange(n), side="right") - 1
    if bins == 0:
        np.testing.assert_array_equal(group, tile)
        pix = ((row & 15) << 4) | (col & 15)
        same = (tile[1:] == tile[:-1])
        assert (pix[1:][same] >= pix[:-1][same]).all()  # by pixel inside a tile
        run_id = np.concatenate([[0], np.cumsum((pix[1:] != pix[:-1]) | ~same)])
        run_len = np.bincount(run_id)
        same_run = run_id[1:] == run_id[:-1]
        radix = n >= 8_000_000 or os.environ.get("CMAX_SORT") == "radix"
        short = (run_len[run_id[1:]] <= 192) | radix
        assert (tau[1:][same_run & short] >= tau[:-1][same_run & short]).all()  # by time inside a pixel run
        assert (run_len <= 192).any() and ((run_len > 192).any() == (n // (size[0] * size[1]) > 150))
    else:
        np.testing.assert_array_equal(group // bins, tile)
        np.testing.assert_array_equal(group % bins, packed[:, 0] >> 24)
