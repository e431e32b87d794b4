"""GPU parity of the leaf operators (Warp / EventImageConverter / costs / flow voxel) through the
C ABI, against the golden fixtures produced by the reference and against the CPU oracle.

fp64 inputs run fp64 kernels, so the tolerances here are tight (1e-9 .. 1e-12): the only
difference from the reference is summation order."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import event_based_optical_flow_amd as E  # noqa: E402
from event_based_optical_flow_amd import functional as F  # noqa: E402
from oracle import oracle as orc  # noqa: E402

DEV = "cuda"


def T(a, dtype=torch.float64):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=DEV)


# ---------------------------------------------------------------------------------------------
# the reference's own known-answer tests, through the mirrored API
# ---------------------------------------------------------------------------------------------
def test_ka_dense_warp_numpy_and_torch(golden):
    g = golden("ka_warp_dense")  # reference tests/test_warp.py:96-139
    warper = E.Warp((3, 4), normalize_t=True)
    warped, feat = warper.warp_event(g["events"], g["flow"], "dense-flow")
    assert isinstance(warped, np.ndarray) and "none" in feat
    np.testing.assert_allclose(warped, g["expected"], rtol=1e-12, atol=1e-12)
    warped_t, _ = warper.warp_event(torch.from_numpy(g["events"]), torch.from_numpy(g["flow"]), "dense-flow")
    assert isinstance(warped_t, torch.Tensor) and warped_t.device.type == "cpu"
    assert torch.allclose(warped_t, torch.from_numpy(g["expected"]))


def test_ka_dense_warp_batch():
    # reference tests/test_warp.py:142-195 (per-batch dt normalisation)
    warper = E.Warp((3, 4), normalize_t=True)
    events = np.array([[[1, 2, 0], [2, 3, 0.2]], [[0, 1, 0.6], [1, 0, 1.2]]])
    flow1 = np.array([[[1.0, -0.5, 2, 8], [-2, 0, 2.0, 0], [2, 1, -2, 0]], [[-10, 1.0, 3, 2], [0, 2, -0.9, 0], [0, 10, -3, 0]]])
    flow = np.stack([flow1, flow1])
    expected = np.array([[[1.0, 2.0, 0], [2, 3, 1.0]], [[0, 1, 0], [3, 0, 1.0]]])
    warped, _ = warper.warp_event(events, flow, "dense-flow")
    np.testing.assert_allclose(warped, expected, rtol=1e-12, atol=1e-12)


def test_ka_bilinear_votes(golden):
    g = golden("ka_vote")  # reference tests/test_event_image_converter.py:17-110
    imager = E.EventImageConverter((3, 4))
    img = imager.bilinear_vote_numpy(g["ev_int"], weight=g["w_int"])
    np.testing.assert_array_equal(img, g["exp_int"])
    img = imager.bilinear_vote_tensor(torch.from_numpy(g["ev_int"]), weight=torch.from_numpy(g["w_int"]))
    assert torch.allclose(img, torch.from_numpy(g["exp_int"]))
    img = imager.bilinear_vote_numpy(g["ev_float"], weight=g["w_float"])
    np.testing.assert_allclose(img, g["exp_float"], rtol=1e-12, atol=1e-15)
    # batched
    ev = np.stack([g["ev_int"], g["ev_float"]])
    wt = np.stack([g["w_int"], g["w_float"]])
    img = imager.bilinear_vote_numpy(ev, weight=wt)
    np.testing.assert_allclose(img, np.stack([g["exp_int"], g["exp_float"]]), rtol=1e-12, atol=1e-15)
    img_t = imager.bilinear_vote_tensor(torch.from_numpy(ev), weight=torch.from_numpy(wt))
    assert torch.allclose(img_t, torch.from_numpy(np.stack([g["exp_int"], g["exp_float"]])))


def test_create_iwe_shape_and_numpy_torch_agreement():
    # reference tests/test_event_image_converter.py:7-14, 113-122
    imager = E.EventImageConverter((100, 200))
    events = np.stack([E.utils.generate_events(100, 99, 199, seed=s) for s in range(4)])
    assert imager.create_iwe(events, sigma=0).shape == (4, 100, 200)
    imager = E.EventImageConverter((10, 20))
    ev = E.utils.generate_events(100, 9, 19, seed=5)
    ev[:, :2] += np.random.default_rng(0).random((100, 1))
    ev = ev.astype(np.float32)
    np.testing.assert_allclose(imager.bilinear_vote_numpy(ev), imager.bilinear_vote_tensor(torch.from_numpy(ev)).numpy(), rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------------------------
# golden fixtures (values produced by the reference)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,direction", [("first", "first"), ("middle", "middle"), ("last", "last"), ("f0p3", 0.3)])
@pytest.mark.parametrize("kind", ["int", "frac"])
def test_warps_golden(golden, tag, direction, kind):
    g = golden("warp")
    size = tuple(int(v) for v in g["image_size"])
    ev = g["events"] if kind == "int" else g["events_frac"]
    warper = E.Warp(size, normalize_t=True)
    for model, key, motion in (("2d-translation", "2dof", g["theta"]), ("dense-flow", "dense", g["flow"]),
                               ("dense-flow-voxel", "voxel", g["voxel"])):
        warped, _ = warper.warp_event(T(ev), T(motion), model, direction)
        np.testing.assert_allclose(warped.cpu().numpy(), g[f"{key}_{kind}_{tag}"], rtol=1e-11, atol=1e-11)
    if kind == "int":
        warped, _ = E.Warp(size, normalize_t=False).warp_event(ev, g["theta"], "2d-translation", direction)
        np.testing.assert_allclose(warped, g[f"2dof_int_raw_{tag}"], rtol=1e-11, atol=1e-11)


def test_warp_fp32_matches_fp64():
    size = (26, 34)
    ev = E.utils.generate_events(5000, *size, seed=3)
    flow = E.utils.generate_dense_optical_flow(size, 5, seed=4)
    warper = E.Warp(size, normalize_t=True)
    w64, _ = warper.warp_event(ev, flow, "dense-flow", "middle")
    w32, _ = warper.warp_event(ev.astype(np.float32), flow.astype(np.float32), "dense-flow", "middle")
    assert w32.dtype == np.float32
    np.testing.assert_allclose(w32, w64, rtol=1e-5, atol=1e-5)  # reference tolerance, tests/test_warp.py:214


@pytest.mark.parametrize("pad", [0, 3])
def test_votes_golden(golden, pad):
    g = golden("vote")
    size = tuple(int(v) for v in g["image_size"])
    imager = E.EventImageConverter(size, outer_padding=pad)
    ev = g["events"]
    tol = dict(rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(imager.bilinear_vote_tensor(T(ev)).cpu().numpy(), g[f"vote_pad{pad}"], **tol)
    np.testing.assert_allclose(imager.bilinear_vote_tensor(T(ev), weight=T(g["weight"])).cpu().numpy(), g[f"vote_w_pad{pad}"], **tol)
    np.testing.assert_allclose(imager.count_event_numpy(ev), g[f"count_pad{pad}"], **tol)
    np.testing.assert_allclose(imager.bilinear_vote_numpy(ev), g[f"vote_numpy_pad{pad}"], **tol)
    np.testing.assert_array_equal(imager.create_eventmask(T(ev)).cpu().numpy(), g[f"mask_pad{pad}"])
    for sigma in (1, 0.7):
        np.testing.assert_allclose(imager.create_iwe(T(ev), "bilinear_vote", sigma).cpu().numpy(), g[f"iwe_s{sigma}_pad{pad}"], **tol)
    # autograd through the vote: coordinates and weights
    te = T(ev).requires_grad_()
    tw = T(g["weight"]).requires_grad_()
    img = imager.bilinear_vote_tensor(te, weight=tw)
    ge, gw = torch.autograd.grad((img * T(g[f"G_pad{pad}"])).sum(), [te, tw])
    np.testing.assert_allclose(ge.cpu().numpy()[:, :2], g[f"gxy_pad{pad}"], **tol)
    np.testing.assert_allclose(gw.cpu().numpy(), g[f"gw_pad{pad}"], **tol)


def test_vote_edge_cases():
    imager = E.EventImageConverter((8, 9))
    # empty input
    img = imager.bilinear_vote_tensor(torch.zeros((0, 4), dtype=torch.float64, device=DEV))
    assert img.shape == (8, 9) and float(img.abs().sum()) == 0.0
    # everything far outside, NaN and huge coordinates: no vote, no crash
    ev = torch.tensor([[-50.0, 3.0, 0, 0], [3.0, 1e12, 0, 0], [float("nan"), 2.0, 0, 0], [1e300, -1e300, 0, 0]],
                      dtype=torch.float64, device=DEV)
    assert float(imager.bilinear_vote_tensor(ev).abs().sum()) == 0.0
    # unknown method / type
    with pytest.raises(NotImplementedError):
        imager.create_iwe(torch.zeros((1, 4), device=DEV), method="nope")
    with pytest.raises(RuntimeError):
        imager.create_iwe([[1.0, 2.0, 0, 0]])


COSTS = ["image_variance", "gradient_magnitude", "normalized_image_variance", "normalized_gradient_magnitude",
         "multi_focal_normalized_image_variance", "multi_focal_normalized_gradient_magnitude"]


@pytest.mark.parametrize("name", COSTS)
@pytest.mark.parametrize("direction", ["minimize", "natural", "maximize"])
@pytest.mark.parametrize("omit", [True, False])
def test_costs_golden(golden, name, direction, omit):
    g = golden("costs")
    t = {k: T(g[v]).requires_grad_() for k, v in (("iwe", "iwe"), ("forward_iwe", "iwe2"), ("middle_iwe", "iwe3"), ("orig_iwe", "orig"))}
    arg = {"iwe": t["iwe"], "backward_iwe": t["iwe"], "forward_iwe": t["forward_iwe"], "middle_iwe": t["middle_iwe"],
           "orig_iwe": t["orig_iwe"], "omit_boundary": omit}
    cost = E.costs.functions[name](direction=direction, store_history=True, precision="64")
    loss = cost.calculate(arg)
    tag = f"{name}__{direction}__omit{int(omit)}"
    np.testing.assert_allclose(loss.item(), g[tag + "__loss"], rtol=1e-11)
    assert cost.get_history()["loss"] == [loss.item()]
    used = [k for k in ("iwe", "forward_iwe", "middle_iwe") if tag + "__g_" + k in g]
    grads = torch.autograd.grad(loss, [t[k] for k in used], allow_unused=True)
    for k, gr in zip(used, grads):
        ref = g[tag + "__g_" + k]
        got = gr.cpu().numpy() if gr is not None else np.zeros_like(ref)
        np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-13 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("name", ["image_variance", "gradient_magnitude"])
@pytest.mark.parametrize("direction", ["minimize", "maximize"])
@pytest.mark.parametrize("omit", [True, False])
def test_costs_on_a_stack_golden(golden, name, direction, omit):
    """[B, H, W] stacks through the cost classes (VERDICT r1 missing #6): the reference's loss and autograd gradient."""
    g = golden("costs_batched")
    t = T(g["stack"]).requires_grad_()
    cost = E.costs.functions[name](direction=direction, precision="64")
    loss = cost.calculate({"iwe": t, "omit_boundary": omit})
    tag = f"{name}__{direction}__omit{int(omit)}"
    np.testing.assert_allclose(loss.item(), g[tag + "__loss"], rtol=1e-11)
    (gr,) = torch.autograd.grad(loss, t)
    ref = g[tag + "__g"]
    np.testing.assert_allclose(gr.cpu().numpy(), ref, rtol=1e-9, atol=1e-13 * np.abs(ref).max())
    if name == "image_variance" and omit and direction == "minimize":
        v = E.costs.ImageVariance(direction="minimize").calculate({"iwe": g["stack"], "omit_boundary": True})
        assert isinstance(v, float)
        np.testing.assert_allclose(v, g["image_variance_numpy__minimize__omit1"], rtol=1e-12)
    if name == "gradient_magnitude":
        with pytest.raises(NotImplementedError):  # cv2.Sobel of the numpy branch is not batch-aware
            E.costs.GradientMagnitude().calculate({"iwe": g["stack"], "omit_boundary": omit})
        four_d = cost.calculate({"iwe": T(g["stack"])[:, None], "omit_boundary": omit})  # [B, 1, H, W] as in gradient_magnitude.py:64-67
        np.testing.assert_allclose(four_d.item(), g[tag + "__loss"], rtol=1e-11)


def test_costs_numpy_branch_and_errors(golden):
    g = golden("costs")
    v = E.costs.ImageVariance(direction="minimize").calculate({"iwe": g["iwe"], "omit_boundary": True})
    assert isinstance(v, float)
    np.testing.assert_allclose(v, g["image_variance_numpy__minimize__omit1"], rtol=1e-12)
    with pytest.raises(NotImplementedError):
        E.costs.ImageVariance().calculate({"iwe": [1.0], "omit_boundary": True})
    with pytest.raises(KeyError):
        E.costs.GradientMagnitude().calculate({"iwe": g["iwe"]})


@pytest.mark.parametrize("shape", ["4x4", "8x8", "2x2", "1x1"])
@pytest.mark.parametrize("omit", [1, 0])
def test_total_variation_golden(golden, shape, omit):
    g = golden("costs")
    tag = f"tv_{shape}_omit{omit}"
    fl = T(g[tag + "__flow"]).requires_grad_()
    loss = E.costs.TotalVariation(direction="minimize", precision="64").calculate({"flow": fl, "omit_boundary": bool(omit)})
    np.testing.assert_allclose(loss.item(), g[tag + "__loss"], rtol=1e-12)
    (gr,) = torch.autograd.grad(loss, fl)
    np.testing.assert_allclose(gr.cpu().numpy(), g[tag + "__g"], rtol=1e-11, atol=1e-14)


def test_hybrid_history():
    # reference tests/costs/test_hybrid.py:12-67
    size = (26, 34)
    ev = T(E.utils.generate_events(1000, *size, seed=1))
    iwe = E.EventImageConverter(size).create_iwe(ev, sigma=0)
    cost = E.costs.HybridCost("minimize", {"image_variance": 1.0, "gradient_magnitude": 0.5}, store_history=True)
    single = E.costs.ImageVariance("minimize", store_history=True)
    for _ in range(3):
        cost.calculate({"iwe": iwe, "omit_boundary": True})
        single.calculate({"iwe": iwe, "omit_boundary": True})
    h = cost.get_history()
    assert set(h) == {"loss", "image_variance", "gradient_magnitude"} and len(h["loss"]) == 3
    np.testing.assert_allclose(h["image_variance"], single.get_history()["loss"], rtol=1e-12)


@pytest.mark.parametrize("fname", ["rand", "smooth", "withzeros"])
@pytest.mark.parametrize("dt", [0.1, -0.1, 0.01, -0.037, 0.0])
@pytest.mark.parametrize("scheme", ["burgers", "upwind"])
def test_flow_steps_golden(golden, fname, dt, scheme):
    g = golden("flow_voxel")
    fl = T(g[f"flow_{fname}"]).requires_grad_()
    tag = f"{fname}_dt{dt}"
    out = F.flow_step(fl, dt, scheme)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g[f"{scheme}_step_{tag}"], rtol=1e-11, atol=1e-12)
    if dt != 0.0:
        (gr,) = torch.autograd.grad((out * T(g[f"{scheme}_cot_{tag}"])).sum(), fl)
        np.testing.assert_allclose(gr.cpu().numpy(), g[f"{scheme}_vjp_{tag}"], rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("fname", ["smooth", "withzeros"])
@pytest.mark.parametrize("Tn,loc", [(10, "middle"), (5, "middle"), (4, "first")])
@pytest.mark.parametrize("scheme", ["burgers", "upwind"])
def test_voxel_golden(golden, fname, Tn, loc, scheme):
    g = golden("flow_voxel")
    fl = T(g[f"flow_{fname}"]).requires_grad_()
    tag = f"{scheme}_{fname}_T{Tn}_{loc}"
    V = E.utils.construct_dense_flow_voxel_torch(fl, Tn, scheme, loc)
    np.testing.assert_allclose(V.detach().cpu().numpy(), g[f"voxel_{tag}"], rtol=1e-10, atol=1e-11)
    (gr,) = torch.autograd.grad((V * T(g[f"voxel_cot_{tag}"])).sum(), fl)
    np.testing.assert_allclose(gr.cpu().numpy(), g[f"voxel_vjp_{tag}"], rtol=1e-8, atol=1e-10)


def test_voxel_reference_properties():
    # reference tests/utils/test_flow_utils.py:52-90: voxel[t0] == flow; T=1 returns the flow
    fl = T(E.utils.generate_dense_optical_flow((20, 30), 20, seed=7))
    for scheme in ("burgers", "upwind"):
        V = E.utils.construct_dense_flow_voxel_torch(fl, 1, scheme)
        np.testing.assert_array_equal(V[0].cpu().numpy(), fl.cpu().numpy())
        V = E.utils.construct_dense_flow_voxel_torch(fl, 60, scheme, "middle")
        np.testing.assert_array_equal(V[30].cpu().numpy(), fl.cpu().numpy())
        V = E.utils.construct_dense_flow_voxel_torch(fl, 60, scheme, "first")
        np.testing.assert_array_equal(V[0].cpu().numpy(), fl.cpu().numpy())


# ---------------------------------------------------------------------------------------------
# whole objective through the leaf API + torch.autograd (what the reference's solver executes)
# ---------------------------------------------------------------------------------------------
YAML_HYBRID = {"multi_focal_normalized_gradient_magnitude": 1.0, "total_variation": 0.01}
OBJ_CASES = [(c, s) for c in ("image_variance", "gradient_magnitude") for s in (0, 1)] + [
    (c, 1) for c in ("normalized_image_variance", "multi_focal_normalized_image_variance",
                     "multi_focal_normalized_gradient_magnitude", "hybrid")]
MOTIONS = {"2dof": ("2d-translation", "theta"), "dense_smooth": ("dense-flow", "flow_smooth"), "voxel": ("dense-flow-voxel", "voxel")}
_KEY_DIR = {"iwe": "first", "forward_iwe": "last", "middle_iwe": "middle"}


def _leaf_objective(size, events, motion, model, cost, sigma, coarse):
    """get_arg_for_cost (reference src/solver/patch_contrast_base.py:289-352) on the mirrored API."""
    warper = E.Warp(size, calculate_feature=True, normalize_t=True)
    imager = E.EventImageConverter(size)
    arg = {"omit_boundary": True, "clip": True}
    keys = cost.required_keys
    if "orig_iwe" in keys:
        arg["orig_iwe"] = imager.create_iwe(events, "bilinear_vote", sigma)
    if "iwe" in keys or "backward_iwe" in keys:
        w, _ = warper.warp_event(events, motion, model, direction="first")
        arg["iwe"] = arg["backward_iwe"] = imager.create_iwe(w, "bilinear_vote", sigma)
    if "forward_iwe" in keys:
        w, _ = warper.warp_event(events, motion, model, direction="last")
        arg["forward_iwe"] = imager.create_iwe(w, "bilinear_vote", sigma)
    if "middle_iwe" in keys:
        w, _ = warper.warp_event(events, motion, model, direction="middle")
        arg["middle_iwe"] = imager.create_iwe(w, "bilinear_vote", sigma)
    if "flow" in keys:
        arg["flow"] = coarse
    return cost.calculate(arg), arg


@pytest.mark.parametrize("mname", list(MOTIONS))
@pytest.mark.parametrize("cost_name,sigma", OBJ_CASES)
def test_leaf_objective_golden(golden, mname, cost_name, sigma):
    g = golden("objective")
    model, mkey = MOTIONS[mname]
    size = tuple(int(v) for v in g["image_size"])
    ev = T(g["events"])
    motion = T(g[mkey]).requires_grad_()
    coarse = T(g["coarse"]).requires_grad_()
    kw = dict(direction="minimize", store_history=False, precision="64")
    cost = E.costs.HybridCost(cost_with_weight=YAML_HYBRID, **kw) if cost_name == "hybrid" else E.costs.functions[cost_name](**kw)
    loss, arg = _leaf_objective(size, ev, motion, model, cost, sigma, coarse)
    tag = f"{mname}__{cost_name}__s{sigma}"
    np.testing.assert_allclose(loss.item(), g[tag + "__loss"], rtol=1e-10)
    ins = [motion] + ([coarse] if "flow" in cost.required_keys else [])
    grads = torch.autograd.grad(loss, ins)
    ref = g[tag + "__grad"]
    np.testing.assert_allclose(grads[0].cpu().numpy(), ref, rtol=1e-7, atol=1e-11 * max(1.0, np.abs(ref).max()))
    if len(grads) > 1:
        np.testing.assert_allclose(grads[1].cpu().numpy(), g[tag + "__grad_coarse"], rtol=1e-9, atol=1e-14)
    for k in ("iwe", "forward_iwe", "middle_iwe", "orig_iwe"):
        if tag + "__" + k in g and k in arg:
            np.testing.assert_allclose(arg[k].detach().cpu().numpy(), g[tag + "__" + k], rtol=1e-10, atol=1e-12)


def test_motion_model_errors():
    warper = E.Warp((10, 20), normalize_t=True)
    assert warper.get_motion_vector_size("2d-translation") == 2  # reference tests/test_warp.py:8-14
    ev = torch.zeros((3, 4), device=DEV)
    with pytest.raises(E.MotionModelKeyError):
        warper.warp_event(ev, torch.zeros(2, device=DEV), "affine")
    with pytest.raises(ValueError):
        warper.warp_event(ev, torch.zeros(2, device=DEV), "2d-translation", direction="sideways")


@pytest.mark.parametrize("n_bin", [4, 10])
@pytest.mark.parametrize("direction", ["first", "middle", "last"])
def test_warp_voxel_optimized_golden(golden, n_bin, direction):
    """a7 ("dense-flow-voxel-optimized", src/warp.py:398-481): one flow propagated bin by bin with Burgers steps (cmax_flow_step)
    and the voxel warp kernel on it -- numpy in / numpy out and tensor in / tensor out against the reference's values, and a
    gradient flows back to the flow through the chain."""
    g = golden("warp_voxel_optimized")
    size = tuple(int(v) for v in g["image_size"])
    warper = E.Warp(size, normalize_t=True)
    ref = g[f"T{n_bin}_{direction}"]
    out_np, feat = warper.warp_event(g["events"], g["flow"], "dense-flow-voxel-optimized", direction, flow_propagate_bin=n_bin)
    assert isinstance(out_np, np.ndarray) and isinstance(feat, dict)
    np.testing.assert_allclose(out_np, ref, rtol=0, atol=1e-9)
    flow = torch.from_numpy(g["flow"]).to(DEV).requires_grad_()
    out_t, _ = warper.warp_event(torch.from_numpy(g["events"]).to(DEV), flow, "dense-flow-voxel-optimized", direction, flow_propagate_bin=n_bin)
    np.testing.assert_allclose(out_t.detach().cpu().numpy(), ref, rtol=0, atol=1e-9)
    (gflow,) = torch.autograd.grad(out_t[:, :2].sum(), flow)
    assert gflow.shape == flow.shape and torch.isfinite(gflow).all() and float(gflow.abs().max()) > 0
    with pytest.raises(ValueError):
        warper.warp_event(g["events"], g["flow"], "dense-flow-voxel-optimized", direction)


@pytest.mark.parametrize("sigma", [1, 2, 0.6])
def test_numpy_branch_blur_golden(golden, sigma):
    """create_iwe on numpy events with sigma > 0 = scipy.ndimage.gaussian_filter (reference line 122-124)."""
    g = golden("blur_numpy")
    size = tuple(int(v) for v in g["image_size"])
    iwe = E.EventImageConverter(size).create_iwe(g["events"], "bilinear_vote", sigma)
    assert isinstance(iwe, np.ndarray)
    np.testing.assert_allclose(iwe, g[f"iwe_numpy_s{sigma}"], rtol=1e-10, atol=1e-13)


@pytest.mark.parametrize("scheme", ["burgers", "upwind"])
@pytest.mark.parametrize("t0", ["middle", "first"])
def test_voxel_chain_second_order_against_difference_quotients(scheme, t0):
    """cmax_voxel_construct_tan / _adj_tan (dual numbers) in fp64: the tangent voxel equals the central difference of
    the voxel, and dgF equals the central difference of the first-order adjoint plus its part that is linear in dgV.
    Smooth flows keep the difference quotients away from the kinks of max / min / sign."""
    import event_based_optical_flow_amd.functional as F

    H, W, Tn = 23, 31, 7
    rng = np.random.default_rng(7)
    f = torch.tensor(E.utils.generate_smooth_flow((H, W), 9.0, grid=3, seed=3), dtype=torch.float64, device="cuda") + 0.37
    df = torch.tensor(E.utils.generate_smooth_flow((H, W), 1.0, grid=4, seed=4), dtype=torch.float64, device="cuda")
    V, dV = F.voxel_construct_tan(f, df, Tn, scheme, t0)
    V0 = F.construct_dense_flow_voxel(f, Tn, scheme, t0)
    assert torch.equal(V, V0)
    eps = 1e-6
    Vp = F.construct_dense_flow_voxel(f + eps * df, Tn, scheme, t0)
    Vm = F.construct_dense_flow_voxel(f - eps * df, Tn, scheme, t0)
    fd = (Vp - Vm) / (2 * eps)
    assert (dV - fd).abs().max().item() <= 1e-6 * fd.abs().max().item()
    # adjoint and its tangent
    gV = torch.tensor(rng.normal(size=(Tn, 2, H, W)), dtype=torch.float64, device="cuda")
    dgV = torch.tensor(rng.normal(size=(Tn, 2, H, W)), dtype=torch.float64, device="cuda")
    gF, dgF = F.voxel_construct_adj_tan(V, dV, gV, dgV, scheme, t0)

    def adj(flow, seed):  # first-order adjoint through autograd of the leaf operator
        x = flow.clone().requires_grad_()
        (g,) = torch.autograd.grad(F.construct_dense_flow_voxel(x, Tn, scheme, t0), x, grad_outputs=seed)
        return g

    g0 = adj(f, gV)
    assert (gF - g0).abs().max().item() <= 1e-12 * g0.abs().max().item()
    fd2 = (adj(f + eps * df, gV) - adj(f - eps * df, gV)) / (2 * eps) + adj(f, dgV)
    assert (dgF - fd2).abs().max().item() <= 2e-6 * fd2.abs().max().item()


def test_two_dof_sources_in_the_outer_padding_vote_like_the_reference():
    """VERDICT r3 #8.  The reference's 2-DoF warp has no bounds check on the SOURCE (src/warp.py:506-520) and bilinear_vote masks per
    corner of the TARGET (src/event_image_converter.py:355-372): with outer_padding > 0 an event whose source lies in the pad -- or
    further out, if the motion brings it in -- still votes.  The leaf operators (Warp + EventImageConverter) keep exactly that
    (against the oracle's restatement of those lines); the fused path cannot pack such events and says so: the pyramid solver hands
    its batches over with on_dropped="raise" (solver/pyramid.py)."""
    import torch

    size, pad, n = (40, 52), 6, 6000
    rng = np.random.default_rng(12)
    ev = np.stack([rng.uniform(-5.0, size[0] + 5.0, n).round(), rng.uniform(-5.0, size[1] + 5.0, n).round(), np.sort(rng.uniform(0, 0.05, n)),
                   rng.integers(0, 2, n).astype(np.float64)], axis=1)
    ev[:40, 0] = -30.0  # far outside: comes in only through the motion
    theta = np.array([31.0, -4.0])
    off = ((ev[:, 0] < 0) | (ev[:, 0] >= size[0]) | (ev[:, 1] < 0) | (ev[:, 1] >= size[1])).sum()
    assert off > 500
    warped_ref, _ = orc.warp_event(ev, theta, "2d-translation", "first", size)
    iwe_ref = orc.create_iwe(warped_ref, size, outer_padding=pad, sigma=0)
    warp = E.Warp(size, normalize_t=True)
    imager = E.EventImageConverter(size, outer_padding=pad)
    warped, _ = warp.warp_event(torch.from_numpy(ev).cuda(), torch.from_numpy(theta).cuda(), "2d-translation", direction="first")
    iwe = imager.create_iwe(warped, method="bilinear_vote", sigma=0).cpu().numpy()
    assert iwe.shape == iwe_ref.shape == (size[0] + 2 * pad, size[1] + 2 * pad)
    np.testing.assert_allclose(iwe, iwe_ref, rtol=0, atol=1e-9 * max(1.0, np.abs(iwe_ref).max()))
    # ... votes of off-sensor sources are in there: the image without them differs
    keep = (ev[:, 0] >= 0) & (ev[:, 0] < size[0]) & (ev[:, 1] >= 0) & (ev[:, 1] < size[1])
    iwe_on = orc.create_iwe(orc.warp_event(ev[keep], theta, "2d-translation", "first", size)[0], size, outer_padding=pad, sigma=0)
    assert np.abs(iwe_ref - iwe_on).max() > 0.5
    # the fused path refuses such a batch when asked to (what the solver asks for) instead of dropping the events silently
    with pytest.raises(ValueError):
        E.CMaxHandle(size, pad).set_keep_outside(False).set_events(ev, on_dropped="raise")
    # ... and BY DEFAULT (round 5) the fused 2-DoF objective keeps them like the leaf operators: same image
    hk = E.CMaxHandle(size, pad).set_events(ev)
    assert hk.batch_info()["outside"] == int(off) and hk.batch_info()["dropped"] == 0
    iwe_fused = hk.iwe(theta, "2d-translation", direction="first", sigma=0).cpu().numpy()
    np.testing.assert_allclose(iwe_fused, iwe_ref, rtol=0, atol=1e-4 * np.abs(iwe_ref).max())
