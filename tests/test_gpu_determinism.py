"""Deterministic mode (cmax_set_deterministic): bit-identical IWE, loss and gradient from run to run.

By default the vote flush, the flow gradient and the statistics use floating-point atomics and the order of events inside
a sorted group depends on atomics of the sort, so two evaluations of the same inputs differ in their last bits.  In
deterministic mode every such sum is an integer sum.  Each case below packs the same batch into FRESH handles (the
sort's own atomics reorder events between handles) and evaluates it several times; the outputs must be identical byte
for byte, and still within 1e-4 of the fp64 oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import event_based_optical_flow_amd as E  # noqa: E402
from oracle import oracle as orc  # noqa: E402

TOL = 1e-4
CASES = [
    ("2d-translation", "image_variance", 0.0, 0),
    ("2d-translation", "multi_focal_normalized_gradient_magnitude", 1.0, 0),
    ("dense-flow", "gradient_magnitude", 0.0, 0),
    ("dense-flow", "normalized_image_variance", 1.0, 0),
    ("dense-flow-voxel", "image_variance", 1.0, 6),
    ("dense-flow-voxel", "multi_focal_normalized_image_variance", 0.0, 6),
]


def rel_max(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _motion(model, size, Tn):
    if model == "2d-translation":
        return np.array([11.0, -7.0])
    if model == "dense-flow":
        return E.utils.generate_smooth_flow(size, 12, seed=21)
    return np.stack([E.utils.generate_smooth_flow(size, 12, seed=21 + b) for b in range(Tn)])


def _run(size, ev, desc, motion, Tn, deterministic):
    h = E.CMaxHandle(size)
    h.set_deterministic(deterministic)
    assert h.deterministic == deterministic
    h.set_events(ev, time_bin=Tn)
    outs = []
    defined = [0] + list(range(1, 1 + desc.n_ref)) + ([5] if desc.normalized else [])  # result[]: loss, v_k, v_orig (the rest is never written)
    for _ in range(2):  # the second evaluation runs on the other vote buffer, with the cached un-warped image
        res, grad = h.evaluate(desc, motion)
        outs.append((res.cpu().numpy()[defined].tobytes(), grad.cpu().numpy().tobytes(), h.last_iwe(0).cpu().numpy().tobytes()))
    return outs, res.cpu().numpy(), grad.cpu().numpy()


@pytest.mark.parametrize("model,cost,sigma,Tn", CASES, ids=[f"{c[0]}-{c[1]}-s{int(c[2])}" for c in CASES])
def test_bit_repeatable(model, cost, sigma, Tn):
    size, n = (180, 240), 400_000
    if model == "2d-translation":
        # moving dots: a 2-DoF gradient is ONE sum over all events, and on uniform-random events it cancels so far that the
        # handful of cell-border events (tests/_border.py) shows at 1e-3 in either mode
        ev = E.utils.generate_structured_events(n, size[0], size[1], (13.0, -8.0), n_dots=1200, seed=31)
    else:
        ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=31)
    motion = _motion(model, size, Tn)
    desc = E.make_descriptor(cost, model, sigma=sigma, time_bin=Tn)
    runs = [_run(size, ev, desc, motion, Tn, True) for _ in range(3)]
    first = runs[0][0][0]
    for outs, _, _ in runs:
        for o in outs:
            assert o == first, "deterministic mode produced different bits"
    res, grad = runs[0][1], runs[0][2]
    ref = orc.objective(ev, motion, model, size, cost=cost, sigma=int(sigma))
    assert abs(res[0] - ref["loss"]) <= TOL * abs(ref["loss"])
    assert rel_max(grad, ref["grad"]) <= TOL
    # the default mode agrees with it to fp32 accumulation noise (and is allowed to differ in the last bits)
    _, res_d, grad_d = _run(size, ev, desc, motion, Tn, False)
    assert abs(res_d[0] - res[0]) <= 1e-6 * abs(res[0])
    assert rel_max(grad_d, grad) <= 1e-5


def test_default_mode_is_not_bit_repeatable_on_a_large_batch():
    """Documents WHY the mode exists: fp32 atomics + the sort's ordering show in the last bits (if this ever starts to
    pass bit-for-bit the default path became deterministic by itself -- fine, the assertion is one-sided)."""
    size, n = (260, 346), 1_000_000
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=46)
    desc = E.make_descriptor("gradient_magnitude", "dense-flow")
    flow = E.utils.generate_smooth_flow(size, 15, seed=5)
    outs = [_run(size, ev, desc, flow, 0, False)[2] for _ in range(3)]
    worst = max(rel_max(o, outs[0]) for o in outs[1:])
    print(f"[determinism] default mode, three fresh handles: max relative gradient difference {worst:.2e}")
    assert worst <= 1e-5


def test_switching_modes_on_one_handle():
    size, n = (96, 128), 80_000
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=33)
    h = E.CMaxHandle(size).set_events(ev)
    desc = E.make_descriptor("image_variance", "2d-translation")
    theta = np.array([6.0, -9.0])
    ref = orc.objective(ev, theta, "2d-translation", size, cost="image_variance", sigma=0)
    for det in (False, True, False, True):
        h.set_deterministic(det)
        res, grad = h.evaluate(desc, theta)
        assert abs(res[0].item() - ref["loss"]) <= TOL * abs(ref["loss"]), det
        assert rel_max(grad.cpu().numpy(), ref["grad"]) <= TOL, det
    iwe = h.iwe(theta, "2d-translation").cpu().numpy()  # cmax_iwe in deterministic mode
    assert rel_max(iwe, ref["iwes"]["iwe"]) <= TOL


@pytest.mark.parametrize("model,cost,sigma,Tn", CASES, ids=[f"{c[0]}-{c[1]}-s{int(c[2])}" for c in CASES])
def test_hvp_bit_repeatable(model, cost, sigma, Tn):
    """Round 3: cmax_objective_hvp in deterministic mode -- integer tangent-vote images, one-writer tangent statistics, integer
    accumulation of the second-order gather.  Three FRESH handles (the sort reorders events between them), two products each:
    identical bytes; equal to the default mode's product to fp32 accumulation noise; symmetric against a second tangent."""
    size, n = (120, 160), 150_000
    if model == "2d-translation":
        ev = E.utils.generate_structured_events(n, size[0], size[1], (13.0, -8.0), n_dots=600, seed=37)
    else:
        ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=37)
    motion = _motion(model, size, Tn)
    desc = E.make_descriptor(cost, model, sigma=sigma, time_bin=Tn)
    rng = np.random.default_rng(5)
    u = rng.standard_normal(np.asarray(motion).shape)
    v = rng.standard_normal(np.asarray(motion).shape)

    def products(deterministic):
        h = E.CMaxHandle(size)
        h.set_deterministic(deterministic)
        h.set_events(ev, time_bin=Tn)
        outs = [h.hvp(desc, motion, u).cpu().numpy() for _ in range(2)]
        return outs, h.hvp(desc, motion, v).cpu().numpy()

    runs = [products(True) for _ in range(3)]
    first = runs[0][0][0].tobytes()
    for outs, _ in runs:
        for o in outs:
            assert o.tobytes() == first, "deterministic mode produced different bits in the Hessian-vector product"
    Hu, Hv = runs[0][0][0].astype(np.float64), runs[0][1].astype(np.float64)
    Hu_default = products(False)[0][0]
    assert rel_max(Hu_default, Hu) <= 2e-4  # fp32 atomics / another summation order on the default side
    lhs, rhs = float((v * Hu).sum()), float((u * Hv).sum())
    assert abs(lhs - rhs) <= 2e-3 * max(abs(lhs), abs(rhs), 1e-300), (lhs, rhs)  # <v, H u> = <u, H v>


@pytest.mark.parametrize("tag", ["plain", "burgers"])
def test_patch_plan_bit_repeatable(golden, tag):
    """The patch plan (cmax_patch_plan_evaluate / _hvp: interpolation, Burgers voxel chain and their adjoints around the fused
    objective) on a deterministic handle: three FRESH handles + plans, two calls each, identical bytes in loss, gradient and
    Hessian-vector product -- the adjoint sweeps of the voxel chain then run their order-free step kernels (no LDS atomics) --
    and the same numbers as the default mode up to its summation order."""
    from event_based_optical_flow_amd.solver import PatchFlowObjective

    g = golden("solver_objective")
    k = f"{tag}_s3"
    size = tuple(int(v) for v in g["image_size"])
    ev = g["events"]
    x = np.asarray(g[k + "__x"], dtype=np.float64).reshape(-1)
    v = np.random.default_rng(3).normal(size=x.shape)

    def run(deterministic):
        h = E.CMaxHandle(size)
        h.set_deterministic(deterministic)
        h.set_events(ev, time_bin=10 if tag == "burgers" else 0)
        obj = PatchFlowObjective(h, ev[:, 2].max() - ev[:, 2].min(), g[k + "__patch_image_size"], g[k + "__patch_size"],
                                 g[k + "__sliding_window"], g[tag + "__patch_shift"], cost="hybrid",
                                 cost_with_weight={"multi_focal_normalized_gradient_magnitude": 1.0, "total_variation": 0.01},
                                 blur_sigma=1, time_aware=(tag == "burgers"), time_bin=10, flow_interpolation="burgers",
                                 t0_flow_location="middle")
        assert obj.has_native_plan
        outs = []
        for _ in range(2):
            loss, grad = obj.value_and_grad_numpy(x)
            hv = obj.hvp_numpy(x, v)
            outs.append((np.float64(loss).tobytes(), np.asarray(grad).tobytes(), np.asarray(hv).tobytes()))
        return outs, float(loss), np.asarray(grad), np.asarray(hv)

    runs = [run(True) for _ in range(3)]
    first = runs[0][0][0]
    for outs, *_ in runs:
        for o in outs:
            assert o == first, "the patch plan on a deterministic handle produced different bits"
    _, loss, grad, hv = runs[0]
    _, loss_d, grad_d, hv_d = run(False)
    assert abs(loss - g[k + "__loss"]) <= TOL * abs(g[k + "__loss"])
    assert rel_max(grad, np.asarray(g[k + "__grad"]).reshape(-1)) <= TOL
    assert abs(loss - loss_d) <= 1e-6 * abs(loss_d)
    assert rel_max(grad, grad_d) <= 1e-5
    assert rel_max(hv, hv_d) <= 2e-4


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("scheme", ["burgers", "upwind"])
def test_leaf_adjoints_bit_repeatable(dtype, scheme):
    """cmax_set_leaf_deterministic (round 5): the stand-alone adjoints of the propagation step and of the voxel chain -- which have no
    handle to read a mode from -- take the order-free step kernels: three runs, identical bytes, and equal to the default (atomic)
    form to accumulation noise."""
    from event_based_optical_flow_amd import functional as F

    H, W, Tn = 70, 93, 7
    rng = np.random.default_rng(17)
    flow = torch.tensor(rng.normal(0.0, 0.8, (2, H, W)), dtype=dtype, device="cuda")
    tol = 1e-12 if dtype == torch.float64 else 2e-5

    def grads():
        f = flow.clone().requires_grad_()
        V = F.construct_dense_flow_voxel(f, Tn, scheme, "middle")
        cot = torch.tensor(np.random.default_rng(3).normal(size=tuple(V.shape)), dtype=dtype, device="cuda")
        (g_vox,) = torch.autograd.grad((V * cot).sum(), f)
        f2 = flow.clone().requires_grad_()
        out = F.flow_step(f2, -0.3, scheme)
        (g_step,) = torch.autograd.grad((out * cot[0]).sum(), f2)
        torch.cuda.synchronize()
        return g_vox.cpu().numpy(), g_step.cpu().numpy()

    ref_vox, ref_step = grads()  # default mode
    prev = F.set_leaf_deterministic(True)
    try:
        assert prev is False
        runs = [grads() for _ in range(3)]
    finally:
        F.set_leaf_deterministic(False)
    for gv, gs in runs[1:]:
        assert gv.tobytes() == runs[0][0].tobytes() and gs.tobytes() == runs[0][1].tobytes()
    assert np.abs(runs[0][0] - ref_vox).max() <= tol * np.abs(ref_vox).max()
    assert np.abs(runs[0][1] - ref_step).max() <= tol * np.abs(ref_step).max()
