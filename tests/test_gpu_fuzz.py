"""Randomised differential test of the fused objective against the fp64 CPU restatement: image size, padding,
batch size, fractional sources, motion model, cost, blur, motion magnitude and time bins drawn per seed.  Every draw
is a (loss, gradient) comparison at the parity gate of the fused path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import event_based_optical_flow_amd as E  # noqa: E402
from oracle import oracle as orc  # noqa: E402

COSTS = ["image_variance", "gradient_magnitude", "normalized_image_variance", "normalized_gradient_magnitude",
         "multi_focal_normalized_image_variance", "multi_focal_normalized_gradient_magnitude"]
MODELS = ["2d-translation", "dense-flow", "dense-flow-voxel"]


def draw_case(seed):
    rng = np.random.default_rng(1000 + seed)
    H, W = int(rng.integers(6, 90)), int(rng.integers(6, 120))
    pad = int(rng.choice([0, 0, 1, 3]))
    n = int(rng.choice([2, 9, 300, 3000, 25000]))
    model = MODELS[seed % 3]
    cost = COSTS[(seed // 3) % len(COSTS)]
    sigma = int(rng.integers(0, 2))
    mag = float(rng.choice([0.5, 4.0, 15.0]))
    vel = rng.uniform(-mag, mag, 2)
    ev = E.utils.generate_structured_events(n, H, W, tuple(vel), n_dots=max(3, n // 60), seed=seed, tmin=0.3, tmax=0.37)
    frac = bool(rng.random() < 0.4)
    if frac:  # fractional source coordinates (rectified events)
        ev[:, 0] = np.minimum(ev[:, 0] + rng.uniform(0, 0.99, n), H - 1e-3)
        ev[:, 1] = np.minimum(ev[:, 1] + rng.uniform(0, 0.99, n), W - 1e-3)
    T = 0
    if model == "2d-translation":
        motion = vel * rng.uniform(0.5, 1.3)
    else:
        f0 = E.utils.generate_smooth_flow((H, W), mag, grid=3, seed=seed + 7)
        f0 = -(f0 * 0.3 + vel[:, None, None])  # dense models warp with minus the flow
        if model == "dense-flow":
            motion = f0
        else:
            T = int(rng.choice([1, 3, 10]))
            motion = np.stack([f0 * (1.0 + 0.05 * k) for k in range(T)])
    # (drawn last, so that the draws above are those of rounds 1-4) the reference time: "first" as get_arg_for_cost warps, or -- costs
    # with ONE reference time -- a fraction of the batch that fp32 cannot hold (Warp.calculate_reftime takes any float, src/warp.py:216-218)
    direction = "first"
    if not cost.startswith("multi_focal") and rng.random() < 0.35:
        direction = float(rng.choice([0.3, 0.7321, 1.0 / 3.0, -0.25, 1.6]))
    return dict(H=H, W=W, pad=pad, n=n, model=model, cost=cost, sigma=sigma, ev=ev, motion=motion, T=T, frac=frac, direction=direction)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("CMAX_FUZZ_SEEDS", "54"))))
def test_random_configuration_against_oracle(seed):
    c = draw_case(seed)
    size = (c["H"], c["W"])
    if c["model"] != "2d-translation":  # the flow as the device holds it: rounded to fp32 (a 2-DoF theta crosses the ABI in fp64)
        c["motion"] = np.asarray(c["motion"], dtype=np.float32).astype(np.float64)
    ref = orc.objective(c["ev"], c["motion"], c["model"], size, cost=c["cost"], sigma=c["sigma"], outer_padding=c["pad"],
                        warp_direction=c["direction"])
    h = E.CMaxHandle(size, c["pad"]).set_events(c["ev"], time_bin=c["T"])
    obj = E.ContrastObjective(h, c["model"], cost=c["cost"], sigma=c["sigma"], warp_direction=c["direction"])
    m = torch.as_tensor(np.ascontiguousarray(c["motion"]), dtype=torch.float64, device="cuda").requires_grad_()
    loss = obj(m)
    (g,) = torch.autograd.grad(loss, m)
    g = g.cpu().numpy()
    info = {k: c[k] for k in ("H", "W", "pad", "n", "model", "cost", "sigma", "T")}
    if not np.isfinite(ref["loss"]):  # degenerate draw (constant image under a normalised cost): both sides agree it is not finite
        assert not np.isfinite(loss.item()), info
        return
    # floor(x' + 1e-6) is discontinuous: an event whose warped coordinate lies within fp32 rounding of a cell border votes into the
    # neighbouring cell in any fp32 evaluation (the reference's own fp32 path too) and takes its derivative from the wrong side of the
    # kink.  K1 decides those cells in fp64 -- from the caller's fp64 theta / the fp32 flow the device holds (rounds 3-4), and since
    # round 5 from the SOURCE COORDINATE and the REFERENCE TIME as the reference's fp64 arithmetic sees them (fractional sources keep the
    # low part of their residual, a reference time that is no dyadic fraction crosses as a pair of floats): the PLAIN gate for every draw.
    tol = 1e-4
    assert abs(loss.item() - ref["loss"]) <= tol * max(abs(ref["loss"]), 1e-12), (info, loss.item(), ref["loss"])
    gmax = np.abs(ref["grad"]).max()
    if gmax > 0:
        err = np.abs(g - ref["grad"])
        assert err.max() <= tol * gmax, (info, c["frac"], c["direction"], err.max(), gmax)
    else:
        assert np.abs(g).max() == 0, info
    # the same evaluation delivered to the host (cmax_objective_host; 2-DoF variance: the pinned-memory finishing kernel) and,
    # where the objective has one, in its raw form: identical to the device-result form up to the atomics' summation order
    desc = E.make_descriptor(c["cost"], c["model"], sigma=float(c["sigma"]), time_bin=c["T"], warp_direction=c["direction"])
    res_h, grad_h = h.evaluate_host(desc, m.detach())
    scale = max(abs(loss.item()), 1e-12)
    assert abs(res_h[0] - loss.item()) <= 2e-6 * scale, (info, res_h[0], loss.item())
    if gmax > 0:
        assert np.abs(grad_h - g).max() <= 2e-5 * np.abs(g).max() + 1e-30, info
    if h.has_raw(desc):
        call, raw, finalize = h.prepare_raw(desc, m.detach())
        call()
        res_r, grad_r = finalize()
        assert abs(res_r[0] - loss.item()) <= 2e-6 * scale and np.abs(grad_r - g).max() <= 2e-5 * np.abs(g).max() + 1e-30, info


# ---- the optimiser's objective (native one-call plan and autograd-chained path) ----------------------------------
def draw_solver_case(seed):
    rng = np.random.default_rng(5000 + seed)
    scale = int(rng.integers(1, 4))
    crop = (int(rng.choice([32, 48, 64])), int(rng.choice([32, 64, 80])))
    H, W = crop[0] + int(rng.integers(0, 7)), crop[1] + int(rng.integers(0, 11))
    psize = (crop[0] // 2 ** scale, crop[1] // 2 ** scale)
    pis = (len(np.arange(0, crop[0], psize[0])), len(np.arange(0, crop[1], psize[1])))
    shift = ((H - crop[0]) // 2, (W - crop[1]) // 2)
    n = int(rng.choice([40, 800, 6000]))
    vel = rng.uniform(-6, 6, 2)
    ev = E.utils.generate_structured_events(n, H, W, tuple(vel), n_dots=max(3, n // 50), seed=seed, tmin=1.0, tmax=1.0 + float(rng.choice([0.02, 0.5])))
    t_scale = ev[:, 2].max() - ev[:, 2].min()
    x = (rng.uniform(-1, 1, (2,) + pis) * 2.0 + vel[:, None, None]) / t_scale  # pixel / time unit
    time_aware = bool(seed % 2)
    cost = str(rng.choice(["hybrid", "hybrid", "image_variance", "multi_focal_normalized_image_variance"]))
    cww = {"multi_focal_normalized_gradient_magnitude": 1.0, "total_variation": float(rng.choice([0.01, 0.3]))} if cost == "hybrid" else None
    return dict(H=H, W=W, ev=ev, x=x.reshape(-1), pis=pis, psize=psize, shift=shift, time_aware=time_aware, cost=cost, cww=cww,
                sigma=int(rng.integers(0, 2)), T=int(rng.choice([2, 5, 10])), scheme=str(rng.choice(["burgers", "upwind"])),
                t0=str(rng.choice(["first", "middle"])), t_scale=t_scale)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("CMAX_FUZZ_SEEDS", "24"))))
def test_random_solver_objective_against_oracle(seed):
    from event_based_optical_flow_amd.solver import PatchFlowObjective
    from event_based_optical_flow_amd.solver.scipy_autograd import TorchWrapper

    c = draw_solver_case(seed)
    size = (c["H"], c["W"])
    ref_loss, ref_grad = orc.solver_objective(c["ev"], c["x"], size, c["pis"], c["psize"], c["psize"], c["shift"], cost=c["cost"],
                                              cost_with_weight=c["cww"], sigma=c["sigma"], time_aware=c["time_aware"], time_bin=c["T"],
                                              flow_interpolation=c["scheme"], t0_flow_location=c["t0"])
    h = E.CMaxHandle(size).set_events(c["ev"], time_bin=c["T"] if c["time_aware"] else 0)
    obj = PatchFlowObjective(h, c["t_scale"], c["pis"], c["psize"], c["psize"], c["shift"], cost=c["cost"], cost_with_weight=c["cww"],
                             blur_sigma=c["sigma"], time_aware=c["time_aware"], time_bin=c["T"], flow_interpolation=c["scheme"],
                             t0_flow_location=c["t0"])
    info = {k: c[k] for k in ("H", "W", "pis", "psize", "shift", "time_aware", "cost", "sigma", "T", "scheme", "t0")}
    assert obj.has_native_plan, info
    if not np.isfinite(ref_loss):  # degenerate draw (an IWE without contrast under a normalised cost): both sides must say so
        w = TorchWrapper(obj, precision="float64")
        w.get_input(c["x"])
        assert not np.isfinite(w.get_value_and_grad(c["x"])[0]), info
        return
    gmax = np.abs(ref_grad).max()
    for path in ("native", "native again", "autograd"):
        w = TorchWrapper(obj, precision="float64")
        w.force_autograd = path == "autograd"
        w.get_input(c["x"])
        loss, grad = w.get_value_and_grad(c["x"])
        assert abs(loss - ref_loss) <= 2e-4 * abs(ref_loss), (path, info, loss, ref_loss)
        err = np.abs(grad - ref_grad)
        # a cell-border event (fp32 floor, see above) shows up in the few patches that see its pixel
        assert (err > 2e-4 * gmax).sum() <= 4 and err.max() <= 2e-2 * gmax, (path, info, err.max(), gmax)
    # exact Hessian-vector products: finite, and the Hessian they come from is symmetric (<u, H v> = <v, H u>)
    rng = np.random.default_rng(seed)
    u, v = rng.normal(size=c["x"].size), rng.normal(size=c["x"].size)
    hu, hv = obj.hvp_numpy(c["x"], u), obj.hvp_numpy(c["x"], v)
    assert np.isfinite(hu).all() and np.isfinite(hv).all(), info
    a, b = float(v @ hu), float(u @ hv)
    assert abs(a - b) <= 2e-3 * max(abs(a), abs(b)) + 2e-4 * min(np.abs(v * hu).sum(), np.abs(u * hv).sum()), (info, a, b)


# ---- one long-lived handle fed different batches, time bins and models in sequence --------------------------------
def test_one_handle_many_batches():
    rng = np.random.default_rng(77)
    H, W = 57, 83
    h = E.CMaxHandle((H, W), 2)
    for it in range(40):
        n = int(rng.choice([3, 50, 700, 9000, 60000]))
        T = int(rng.choice([0, 0, 1, 4, 10]))
        vel = rng.uniform(-8, 8, 2)
        ev = E.utils.generate_structured_events(n, H, W, tuple(vel), n_dots=max(3, n // 70), seed=it)
        if rng.random() < 0.3:
            ev[:, 0] = np.minimum(ev[:, 0] + rng.uniform(0, 0.99, n), H - 1e-3)
        h.set_events(ev, time_bin=T)
        for rep in range(int(rng.integers(1, 4))):
            if T > 0 and rng.random() < 0.7:
                model, motion = "dense-flow-voxel", np.stack([-(E.utils.generate_smooth_flow((H, W), 3.0, seed=it + k) + vel[:, None, None]) for k in range(T)])
            elif rng.random() < 0.5:
                model, motion = "dense-flow", -(E.utils.generate_smooth_flow((H, W), 3.0, seed=it) + vel[:, None, None])
            else:
                model, motion = "2d-translation", vel * rng.uniform(0.6, 1.2)
            cost = str(rng.choice(COSTS))
            sigma = int(rng.integers(0, 2))
            ref = orc.objective(ev, motion, model, (H, W), cost=cost, sigma=sigma, outer_padding=2)
            obj = E.ContrastObjective(h, model, cost=cost, sigma=sigma)
            m = torch.as_tensor(np.ascontiguousarray(motion), dtype=torch.float64, device="cuda").requires_grad_()
            loss = obj(m)
            (g,) = torch.autograd.grad(loss, m)
            info = (it, rep, n, T, model, cost, sigma)
            assert abs(loss.item() - ref["loss"]) <= 3e-4 * abs(ref["loss"]), (info, loss.item(), ref["loss"])
            gmax = np.abs(ref["grad"]).max()
            err = np.abs(g.cpu().numpy() - ref["grad"])
            # at most a handful of cell-border events (fp32 floor) may differ by a whole vote
            assert (err > 3e-4 * gmax).sum() <= 8 and err.max() <= 5e-2 * gmax, (info, err.max(), gmax, int((err > 3e-4 * gmax).sum()))
        if T == 0 and rng.random() < 0.5:  # re-bin the same batch in place
            h.set_time_bins(int(rng.choice([2, 6])))
            T2 = h.time_bin
            motion = np.stack([-(E.utils.generate_smooth_flow((H, W), 2.0, seed=it) + vel[:, None, None])] * T2)
            ref = orc.objective(ev, motion, "dense-flow-voxel", (H, W), cost="image_variance", sigma=0, outer_padding=2)
            loss = E.ContrastObjective(h, "dense-flow-voxel", cost="image_variance", sigma=0)(torch.as_tensor(motion, device="cuda"))
            assert abs(loss.item() - ref["loss"]) <= 3e-4 * abs(ref["loss"]), (it, "rebinned", loss.item(), ref["loss"])


# ---- time slabs and events off the sensor on random batches (round 4) ----------------------------------------------------
@pytest.mark.parametrize("seed", range(6))
def test_random_time_slabs_and_outside_events_against_oracle(seed):
    """cmax_set_time_slabs is only an ORDER of the batch (2-DoF and dense objectives, any cost, exact product included); with
    cmax_set_keep_outside a 2-DoF batch keeps its events from off the sensor.  Random sizes, slab counts, motions up to 120 px."""
    rng = np.random.default_rng(4100 + seed)
    H, W = int(rng.integers(40, 130)), int(rng.integers(40, 170))
    pad = int(rng.choice([0, 3]))
    n = int(rng.choice([500, 20000, 150000]))
    ev = E.utils.generate_events(n, H, W, 0.0, 0.05, seed=seed)
    outside = seed % 2 == 1
    if outside:
        sel = rng.random(n) < 0.25
        ev[sel, 0] = rng.uniform(-30.0, H + 30.0, int(sel.sum()))
        ev[sel, 1] = rng.uniform(-30.0, W + 30.0, int(sel.sum()))
    h = E.CMaxHandle((H, W), pad)
    if outside:
        h.set_keep_outside(True)
    h.set_events(ev, on_dropped="ignore")
    for slabs in (0, int(rng.choice([2, 3, 5, 8]))):
        h.set_time_slabs(slabs)
        for model in (("2d-translation",) if outside else ("2d-translation", "dense-flow")):
            mag = float(rng.choice([4.0, 40.0, 120.0]))
            motion = rng.uniform(-mag, mag, 2) if model == "2d-translation" else E.utils.generate_smooth_flow((H, W), mag, grid=3, seed=seed + 7)
            cost, sigma = str(rng.choice(COSTS)), int(rng.integers(0, 2))
            ref = orc.objective(ev, motion, model, (H, W), cost=cost, sigma=sigma, outer_padding=pad)
            desc = E.make_descriptor(cost, model, sigma=float(sigma))
            res, grad = h.evaluate(desc, motion)
            info = (seed, H, W, pad, n, slabs, model, cost, sigma, mag, outside)
            assert abs(res[0].item() - ref["loss"]) <= 3e-4 * abs(ref["loss"]), (info, res[0].item(), ref["loss"])
            gmax = np.abs(ref["grad"]).max()
            err = np.abs(grad.double().cpu().numpy() - ref["grad"])
            assert (err > 3e-4 * gmax).sum() <= 8 and err.max() <= 5e-2 * gmax, (info, err.max(), gmax)
    # the exact Hessian-vector product does not depend on the order either
    desc = E.make_descriptor("image_variance", "2d-translation", sigma=1.0)
    theta, v = np.array([11.0, -6.0]), np.array([0.3, 0.8])
    h.set_time_slabs(0)
    hv0 = h.hvp(desc, theta, v).double().cpu().numpy()
    h.set_time_slabs(4)
    hv4 = h.hvp(desc, theta, v).double().cpu().numpy()
    assert np.abs(hv4 - hv0).max() <= 1e-5 * np.abs(hv0).max(), (seed, hv0, hv4)


# ---- per-patch search on random boxes ------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(8))
def test_random_patch_search_against_oracle(seed):
    rng = np.random.default_rng(900 + seed)
    H, W = int(rng.integers(20, 100)), int(rng.integers(20, 130))
    n = int(rng.choice([30, 2000, 20000]))
    ev = E.utils.generate_structured_events(n, H, W, (7.0, -4.0), n_dots=max(3, n // 40), seed=seed, tmin=2.0, tmax=2.04)
    if seed % 2:
        ev[:, 1] = np.minimum(ev[:, 1] + rng.uniform(0, 0.99, n), W - 1e-3)
    n_patch, n_cand = int(rng.integers(1, 12)), int(rng.integers(1, 9))
    x0, y0 = rng.integers(-4, H - 4, n_patch), rng.integers(-4, W - 4, n_patch)
    boxes = np.stack([x0, x0 + rng.integers(1, 40, n_patch), y0, y0 + rng.integers(1, 50, n_patch)], axis=1)
    size = (int(rng.integers(8, 60)), int(rng.integers(8, 80)))
    cand = rng.uniform(-400, 400, (n_patch, n_cand, 2))
    sigma = float(rng.choice([0.0, 1.0, 2.0]))
    h = E.CMaxHandle((H, W)).set_events(ev, time_bin=int(rng.choice([0, 5])))
    loss, gm, count = h.patch_search(boxes, size, cand, sigma)
    loss_o, gm_o, count_o = orc.patch_search(ev, boxes, size, cand, sigma)
    np.testing.assert_array_equal(count.cpu().numpy(), count_o)
    err = np.abs(gm.cpu().numpy() - gm_o)
    assert err.max() <= 2e-4 * max(gm_o.max(), 1e-12), (H, W, n, size, sigma, err.max(), gm_o.max())


# ---- leaf operators on random shapes, fp64 and fp32 ----------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("seed", range(12))
def test_random_leaf_operators_against_oracle(seed, dtype):
    from event_based_optical_flow_amd import functional as F

    rng = np.random.default_rng(300 + seed)
    H, W = int(rng.integers(3, 70)), int(rng.integers(3, 90))
    pad = int(rng.choice([0, 0, 2]))
    n = int(rng.choice([1, 5, 400, 12000]))
    tol = 1e-10 if dtype == torch.float64 else 2e-4
    ev = E.utils.generate_events(n, H, W, tmin=0.1, tmax=0.4, seed=seed)
    if seed % 2:
        ev[:, 0] = np.minimum(ev[:, 0] + rng.uniform(0, 0.99, n), H - 1e-3)
        ev[:, 1] = np.minimum(ev[:, 1] + rng.uniform(0, 0.99, n), W - 1e-3)
    if dtype == torch.float32:
        ev = ev.astype(np.float32).astype(np.float64)  # identical inputs for both sides
    t_ev = torch.as_tensor(ev, dtype=dtype, device="cuda")

    def close(got, ref, what, scale=None):
        got, ref = np.asarray(got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else got, dtype=np.float64), np.asarray(ref)
        s = np.abs(ref).max() if scale is None else scale
        assert np.abs(got - ref).max() <= tol * max(s, 1e-30), (what, H, W, pad, n, np.abs(got - ref).max(), s)

    # warps (all three models, a random reference time)
    direction = [0.0, 0.5, 1.0, 0.27][seed % 4]
    dname = {0.0: "first", 0.5: "middle", 1.0: "last"}.get(direction, direction)
    theta = rng.uniform(-9, 9, 2)
    flow = rng.uniform(-9, 9, (2, H, W))
    T = int(rng.choice([1, 4]))
    voxel = rng.uniform(-9, 9, (T, 2, H, W))
    warped = None
    if n > 1:
        for model, motion in (("2d-translation", theta), ("dense-flow", flow), ("dense-flow-voxel", voxel)):
            ref, _ = orc.warp_event(ev, motion, model, dname, (H, W))
            got = F.warp_events(t_ev, torch.as_tensor(motion, dtype=dtype, device="cuda"), model, (H, W), dname, normalize_t=True)
            close(got[:, :3], ref[:, :3], "warp " + model, scale=max(H, W))
        warped = ref
    # votes (with padding, per-event weights; fp32 compared on warped coordinates rounded to fp32)
    pts = (warped if warped is not None else ev).copy()
    if dtype == torch.float32:
        pts = pts.astype(np.float32).astype(np.float64)
    wts = rng.uniform(0.2, 2.0, n)
    if dtype == torch.float32:
        wts = wts.astype(np.float32).astype(np.float64)
    Hp, Wp = H + 2 * pad, W + 2 * pad
    ref_img = orc.vote(pts, (H, W), pad, weight=wts, eps=1e-6)
    got_img = F.vote(torch.as_tensor(pts, dtype=dtype, device="cuda"), (Hp, Wp), (pad, pad), weight=torch.as_tensor(wts, dtype=dtype, device="cuda"))
    if dtype == torch.float64:
        close(got_img, ref_img, "vote")
    else:  # the fp32 floor may move a border event by one cell
        err = np.abs(got_img.cpu().numpy() - ref_img)
        assert (err > 2e-4 * max(ref_img.max(), 1e-30)).sum() <= 8, ("vote fp32", H, W, n)
    # blur, contrast values and image gradients, total variation
    img = rng.uniform(0, 5, (Hp, Wp))
    if dtype == torch.float32:
        img = img.astype(np.float32).astype(np.float64)
    t_img = torch.as_tensor(img, dtype=dtype, device="cuda")
    sigma = float(rng.choice([0.5, 1.0, 2.5]))
    close(F.gaussian_blur3(t_img, sigma), orc.blur3(img, sigma), "blur3")
    for omit in (False, True):
        if omit and min(Hp, Wp) <= 2:
            continue
        for cost_id, fn in ((0, lambda a, o: orc.variance(a, o, 1)), (1, orc.gradmag)):
            v_ref, G_ref = fn(img, omit)
            if not np.isfinite(v_ref):
                continue
            ti = t_img.clone().requires_grad_()
            v = F.contrast(ti, cost_id, omit)
            (G,) = torch.autograd.grad(v, ti)
            assert abs(v.item() - v_ref) <= tol * max(abs(v_ref), 1e-30), ("contrast", cost_id, omit, Hp, Wp, v.item(), v_ref)
            close(G, G_ref, f"contrast grad {cost_id} {omit}")
    if min(H, W) > 2:
        fl = rng.uniform(-3, 3, (2, H, W))
        if dtype == torch.float32:
            fl = fl.astype(np.float32).astype(np.float64)
        tf = torch.as_tensor(fl, dtype=dtype, device="cuda").requires_grad_()
        for omit in (False, True):
            v_ref, G_ref = orc.total_variation(fl, omit)
            v = F.total_variation(tf, omit)
            (G,) = torch.autograd.grad(v, tf)
            assert abs(v.item() - v_ref) <= tol * abs(v_ref), ("tv", omit, H, W)
            close(G, G_ref, "tv grad")
        # Burgers / upwind propagation and the voxel chain with its adjoint
        for scheme, step, step_adj in (("burgers", orc.burgers_step, orc.burgers_step_adj), ("upwind", orc.upwind_step, orc.upwind_step_adj)):
            dt = float(rng.choice([-0.1, 0.1, 0.25]))
            close(F.flow_step(tf.detach(), dt, scheme), step(fl, dt), scheme + " step")
            Tb = int(rng.choice([1, 2, 5]))
            t0 = str(rng.choice(["first", "middle"]))
            V_ref = orc.construct_dense_flow_voxel(fl, Tb, scheme, t0)
            tf2 = tf.detach().clone().requires_grad_()
            V = F.construct_dense_flow_voxel(tf2, Tb, scheme, t0)
            close(V, V_ref, scheme + " voxel")
            gV = rng.uniform(-1, 1, V_ref.shape)
            if dtype == torch.float32:
                gV = gV.astype(np.float32).astype(np.float64)
            (gF,) = torch.autograd.grad(V, tf2, grad_outputs=torch.as_tensor(gV, dtype=dtype, device="cuda"))
            close(gF, orc.construct_dense_flow_voxel_adj(V_ref, gV, scheme, t0), scheme + " voxel adjoint")


# ---- non-finite / absurd inputs must neither fault nor poison the handle ---------------------------------------------------
@pytest.mark.parametrize("model", MODELS)
def test_non_finite_inputs_do_not_poison_the_handle(model):
    rng = np.random.default_rng(4)
    H, W, T = 45, 61, 4
    ev = E.utils.generate_structured_events(8000, H, W, (5.0, -3.0), n_dots=150, seed=2)
    bad = ev.copy()
    bad[::97, 0] = np.nan
    bad[5::131, 1] = np.inf
    bad[7::151, 2] = np.nan
    bad[11::173, 0] = -1e30
    bad[13::191, 1] = 1e30
    tb = T if model == "dense-flow-voxel" else 0
    h = E.CMaxHandle((H, W)).set_events(ev, time_bin=tb)
    shape = {"2d-translation": (2,), "dense-flow": (2, H, W), "dense-flow-voxel": (T, 2, H, W)}[model]
    good = -np.broadcast_to(np.array([5.0, -3.0]).reshape((2,) + (1,) * (len(shape) - 1) if model != "dense-flow-voxel" else (1, 2, 1, 1)), shape).copy()
    if model == "2d-translation":
        good = -good
    ref = orc.objective(ev, good, model, (H, W), cost="gradient_magnitude", sigma=1)

    def run(motion, handle=h):
        obj = E.ContrastObjective(handle, model, cost="gradient_magnitude", sigma=1)
        m = torch.as_tensor(np.ascontiguousarray(motion), dtype=torch.float64, device="cuda").requires_grad_()
        loss = obj(m)
        (g,) = torch.autograd.grad(loss, m)
        torch.cuda.synchronize()
        return loss.item(), g.cpu().numpy()

    for poison in (np.nan, np.inf, -np.inf, 1e30, -3e38, 1e-40):
        m = good.copy()
        m.reshape(-1)[:: max(1, m.size // 7)] = poison
        run(m)  # value is meaningless; it must come back
        loss, g = run(good)
        assert abs(loss - ref["loss"]) <= 1e-4 * abs(ref["loss"]), (poison, loss, ref["loss"])
        assert np.abs(g - ref["grad"]).max() <= 1e-4 * np.abs(ref["grad"]).max(), poison
    # a batch with non-finite coordinates / timestamps: those events are dropped or harmless, the rest is evaluated
    h2 = E.CMaxHandle((H, W)).set_events(bad, time_bin=tb)
    run(good, h2)
    h2.set_events(ev, time_bin=tb)
    loss, g = run(good, h2)
    assert abs(loss - ref["loss"]) <= 1e-4 * abs(ref["loss"])


# ---- exact Hessian-vector product of the fused objective: symmetry and agreement with a difference quotient ----------------
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("CMAX_FUZZ_SEEDS", "18"))))
def test_random_fused_hvp_is_symmetric(seed):
    rng = np.random.default_rng(7000 + seed)
    H, W = int(rng.integers(12, 70)), int(rng.integers(12, 90))
    pad = int(rng.choice([0, 0, 2]))
    n = int(rng.choice([60, 2000, 30000]))
    model = MODELS[seed % 3]
    cost = COSTS[(seed // 3) % len(COSTS)]
    sigma = int(rng.integers(0, 2))
    vel = rng.uniform(-6, 6, 2)
    ev = E.utils.generate_structured_events(n, H, W, tuple(vel), n_dots=max(3, n // 60), seed=seed)
    T = int(rng.choice([2, 5])) if model == "dense-flow-voxel" else 0
    if model == "2d-translation":
        motion = vel * 0.9
    else:
        f0 = -(E.utils.generate_smooth_flow((H, W), 2.0, grid=3, seed=seed) + vel[:, None, None] * 0.9)
        motion = f0 if model == "dense-flow" else np.stack([f0 * (1 + 0.03 * k) for k in range(T)])
    h = E.CMaxHandle((H, W), pad).set_events(ev, time_bin=T)
    obj = E.ContrastObjective(h, model, cost=cost, sigma=sigma)
    assert obj.has_exact_hvp
    m = torch.as_tensor(np.ascontiguousarray(motion), dtype=torch.float64, device="cuda")
    if model == "2d-translation":
        u, v = torch.tensor([1.0, 0.3], dtype=torch.float64, device="cuda"), torch.tensor([-0.4, 1.0], dtype=torch.float64, device="cuda")
    else:  # smooth directions (what an optimiser on a patch grid produces)
        u = torch.as_tensor(np.broadcast_to(E.utils.generate_smooth_flow((H, W), 1.0, grid=3, seed=seed + 50), m.shape).copy(), device="cuda")
        v = torch.as_tensor(np.broadcast_to(E.utils.generate_smooth_flow((H, W), 1.0, grid=4, seed=seed + 90), m.shape).copy(), device="cuda")
    hu, hv = obj.hvp(m, u).double(), obj.hvp(m, v).double()
    info = (H, W, pad, n, model, cost, sigma, T)
    assert torch.isfinite(hu).all() and torch.isfinite(hv).all(), info
    a, b = float((v * hu).sum()), float((u * hv).sum())
    assert abs(a - b) <= 2e-3 * max(abs(a), abs(b)) + 2e-4 * min(float((v * hu).abs().sum()), float((u * hv).abs().sum())), (info, a, b)


# ---- time slices: votes of disjoint slices add up to the whole batch, the gradient parts add up to the whole gradient ----------
@pytest.mark.parametrize("seed", range(9))
def test_random_time_slices_sum_to_the_whole(seed):
    rng = np.random.default_rng(8000 + seed)
    H, W = int(rng.integers(20, 80)), int(rng.integers(20, 100))
    n = int(rng.choice([500, 8000, 50000]))
    model = MODELS[seed % 3]
    cost = COSTS[(seed // 3 + seed) % len(COSTS)]
    sigma = int(rng.integers(0, 2))
    vel = rng.uniform(-6, 6, 2)
    ev = E.utils.generate_structured_events(n, H, W, tuple(vel), n_dots=max(3, n // 60), seed=seed)
    T = 4 if model == "dense-flow-voxel" else 0
    if model == "2d-translation":
        motion = vel
    else:
        f0 = -(E.utils.generate_smooth_flow((H, W), 2.0, grid=3, seed=seed) + vel[:, None, None])
        motion = f0 if model == "dense-flow" else np.stack([f0] * T)
    desc = E.make_descriptor(cost, model, sigma=sigma, time_bin=T)
    m = torch.as_tensor(np.ascontiguousarray(motion), dtype=torch.float32, device="cuda")
    whole = E.CMaxHandle((H, W)).set_events(ev, time_bin=T)
    res_w, grad_w = whole.evaluate(desc, m, True)
    k = int(rng.integers(2, 6))
    cuts = np.sort(rng.choice(np.arange(1, n), size=k - 1, replace=False))
    bounds = [0] + list(cuts) + [n]
    tmin, tmax = float(ev[:, 2].min()), float(ev[:, 2].max())
    handles = [E.CMaxHandle((H, W)).set_events(ev[a:b], tmin, tmax, time_bin=T) for a, b in zip(bounds[:-1], bounds[1:])]
    images = sum(hd.objective_vote(desc, m) for hd in handles)
    parts = [hd.objective_finish(desc, m, images, True) for hd in handles]
    grad = sum(p[1].double() for p in parts)
    info = (H, W, n, model, cost, sigma, k)
    for res, _ in parts:
        assert abs(res[0].item() - res_w[0].item()) <= 1e-5 * abs(res_w[0].item()), (info, res[0].item(), res_w[0].item())
    gmax = grad_w.abs().max().item()
    assert (grad - grad_w.double()).abs().max().item() <= 2e-4 * gmax, (info, (grad - grad_w.double()).abs().max().item(), gmax)


# ---- sensors with more than 4096 source tiles: the sort's global-atomic path and the three-launch scan ---------------------------
@pytest.mark.parametrize("model,time_bin", [("2d-translation", 0), ("dense-flow", 0), ("dense-flow-voxel", 3)])
def test_large_sensor_takes_the_global_tile_path(model, time_bin):
    H, W = 1100, 1210  # 69 x 76 = 5244 tiles
    rng = np.random.default_rng(12)
    n = 40000
    vel = np.array([6.0, -4.0])
    ev = E.utils.generate_structured_events(n, H, W, tuple(vel), n_dots=600, seed=5)
    ev[100:150, 0] = H + 3.0  # some events off the sensor (not the first / last of the batch: the time extremes stay)
    if model == "2d-translation":
        motion = vel
    else:
        f0 = -(E.utils.generate_smooth_flow((H, W), 2.0, grid=3, seed=1) + vel[:, None, None])
        motion = f0 if model == "dense-flow" else np.stack([f0] * time_bin)
    h = E.CMaxHandle((H, W)).set_keep_outside(False).set_events(ev, time_bin=time_bin)  # (dropped on request: dense models need it)
    assert h.n_events == n - 50
    keep = np.concatenate([ev[:100], ev[150:]])
    keep_tmm = (float(ev[:, 2].min()), float(ev[:, 2].max()))
    ref = orc.objective(keep, motion, model, (H, W), cost="image_variance", sigma=0)
    obj = E.ContrastObjective(h, model, cost="image_variance", sigma=0)
    m = torch.as_tensor(np.ascontiguousarray(motion), dtype=torch.float64, device="cuda").requires_grad_()
    loss = obj(m)
    (g,) = torch.autograd.grad(loss, m)
    # the dropped events still count for the batch's time extremes (they are part of the batch): evaluate the oracle on
    # the kept events with the same normalisation by keeping the first / last timestamps inside `keep`
    if float(keep[:, 2].min()) == keep_tmm[0] and float(keep[:, 2].max()) == keep_tmm[1]:
        assert abs(loss.item() - ref["loss"]) <= 1e-4 * abs(ref["loss"]), (loss.item(), ref["loss"])
        err = np.abs(g.cpu().numpy() - ref["grad"])
        gmax = np.abs(ref["grad"]).max()
        assert (err > 2e-4 * gmax).sum() <= 8 and err.max() <= 5e-2 * gmax, (err.max(), gmax)
    if time_bin == 0:  # re-bin in place through the same path
        h.set_time_bins(4)
        motion4 = np.stack([-(E.utils.generate_smooth_flow((H, W), 2.0, grid=3, seed=1) + vel[:, None, None])] * 4)
        ref4 = orc.objective(keep, motion4, "dense-flow-voxel", (H, W), cost="image_variance", sigma=0)
        l4 = E.ContrastObjective(h, "dense-flow-voxel", cost="image_variance", sigma=0)(torch.as_tensor(motion4, device="cuda"))
        assert abs(l4.item() - ref4["loss"]) <= 1e-4 * abs(ref4["loss"]), (l4.item(), ref4["loss"])


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("CMAX_FUZZ_OWNED_SEEDS", "4"))))
def test_random_owned_groups_configuration_against_oracle(seed):
    """The same differential test at batch sizes that reach the group-aligned work list (>= 522k events, <= 7 events per
    pixel): owned-groups K3, the statistics inside its launch (plain variance), the one-kernel blurred variance with the
    mean from K1's vote sums, normalised costs on the same list -- sensor size, padding, flow magnitude (votes that leave
    the image), time bins, direction and omit_boundary drawn per seed.  CMAX_FUZZ_OWNED_SEEDS=n runs more draws."""
    from _border import ambiguity_bound, raw_image_grad

    rng = np.random.default_rng(7000 + seed)
    H, W = int(rng.integers(270, 360)), int(rng.integers(330, 420))
    pad = int(rng.choice([0, 0, 2]))
    n = int(rng.integers(530_000, 640_000))
    model = ["dense-flow", "dense-flow-voxel"][seed % 2]
    cost = ["image_variance", "image_variance", "normalized_image_variance", "gradient_magnitude"][(seed // 2) % 4]
    sigma = int(rng.integers(0, 2))
    omit = bool(rng.integers(0, 2))
    direction = ["minimize", "maximize", "natural"][int(rng.integers(0, 3))]
    mag = float(rng.choice([3.0, 15.0, 45.0]))
    ev = E.utils.generate_events(n, H, W, 0.1, 0.16, seed=seed + 500)
    f0 = E.utils.generate_smooth_flow((H, W), mag, grid=4, seed=seed + 17)
    T = 0
    if model == "dense-flow":
        motion = f0
    else:
        T = int(rng.choice([2, 5]))
        motion = np.stack([f0 * (1.0 + 0.07 * k) for k in range(T)])
    size = (H, W)
    h = E.CMaxHandle(size, pad).set_events(ev, time_bin=T) if T else E.CMaxHandle(size, pad).set_events(ev)
    info = dict(H=H, W=W, pad=pad, n=n, model=model, cost=cost, sigma=sigma, omit=omit, direction=direction, mag=mag, T=T)
    assert h.batch_info()["owned_groups"], info
    desc = E.make_descriptor(cost, model, direction=direction, sigma=float(sigma), omit_boundary=omit, time_bin=T)
    ref = orc.objective(ev, motion, model, size, cost=cost, sigma=sigma, outer_padding=pad, omit_boundary=omit, direction=direction)
    tol = 1e-4
    for rep in range(2):
        res, grad = h.evaluate(desc, motion)
        assert abs(res[0].item() - ref["loss"]) <= tol * abs(ref["loss"]), (info, rep, res[0].item(), ref["loss"])
        g = grad.double().cpu().numpy()
        gmax = np.abs(ref["grad"]).max()
        err = np.abs(g - ref["grad"])
        if err.max() > tol * gmax:
            # events within fp32 rounding of a bilinear cell border (tests/_border.py): a few entries, never a pattern
            n_over = int((err > tol * gmax).sum())
            assert n_over <= max(8, int(2e-5 * err.size)), (info, rep, n_over, err.max() / gmax)
            if pad == 0 and not cost.startswith("normalized"):  # ... and each within the bound of its border events
                bound, _ = ambiguity_bound(ev, motion, model, size, raw_image_grad(ref, sigma))
                assert ((err - 1.01 * bound).max()) <= tol * gmax, (info, rep, err.max() / gmax)
