"""Generate the golden fixtures tests/golden/*.npz by RUNNING THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference):   python tests/golden/gen_golden.py
Fixtures are data only (inputs + the reference's outputs, fp64); seeds are recorded.  The
reference code is imported in place through ref_import.py (stubs + 3 shims, see there).

What is pinned:
  ka_*      the reference's own known-answer test arrays (tests/test_warp.py:96-195,
            tests/test_event_image_converter.py:17-110), re-verified against the reference here
  warp_*    Warp.warp_event for 2-DoF / dense / voxel, directions first/middle/last/float
  vote_*    bilinear_vote_tensor / count_event_tensor incl. weights, padding, out-of-image votes
  blur_*    create_image_from_events_tensor(sigma>0) (shim (1))
  cost_*    every cost class: value + autograd gradient w.r.t. the IWE(s) / flow
  burgers_* / upwind_* / voxel_*   single steps, voxel construction, and their VJPs
  obj_*     one full objective evaluation through the reference's get_arg_for_cost +
            cost.calculate + torch.autograd.grad, for model x cost x sigma
  hvp_*     one vhp through torch.autograd.functional.vhp
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

src = ref_import.import_reference()
from src import costs, event_image_converter, utils, warp  # noqa: E402
from src.solver.patch_contrast_base import PatchContrastMaximization  # noqa: E402

SEED = 46  # src/utils/misc.py:18


def save(name, **arrays):
    arrays["shims"] = np.array(ref_import.SHIMS)
    arrays["seed"] = np.array(SEED)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in arrays.items() if k not in ("shims", "seed")})


def make_events(n, H, W, rng, fractional=False, tmin=0.0, tmax=0.05):
    """Semantics of src/utils/event_utils.py:18-47 (integer pixel coords, sorted uniform t)."""
    x = rng.integers(0, H, n).astype(np.float64)
    y = rng.integers(0, W, n).astype(np.float64)
    if fractional:
        x = np.clip(x + rng.uniform(0, 0.999, n), 0, H - 1e-3)
        y = np.clip(y + rng.uniform(0, 0.999, n), 0, W - 1e-3)
    t = np.sort(rng.uniform(tmin, tmax, n))
    p = rng.integers(0, 2, n).astype(np.float64)
    return np.stack([x, y, t, p], axis=1)


def smooth_flow(H, W, rng, mag):
    """Bilinear upsample of a coarse random grid -- the kind of field the solver produces."""
    g = torch.from_numpy(rng.uniform(-mag, mag, (1, 2, 5, 6)))
    f = torch.nn.functional.interpolate(g, size=(H, W), mode="bilinear", align_corners=True)[0]
    return f.numpy().copy()


# ----------------------------------------------------------------------------------------------
def gen_known_answers():
    # tests/test_warp.py:96-139
    events = np.array([[1, 2, 0], [2, 3, 0.2], [0, 1, 0.6], [1, 0, 1.0]])
    flow = np.array(
        [[[1.0, -0.5, 2, 8], [-2, 0, 2.0, 0], [2, 1, -2, 0]], [[-10, 1.0, 3, 2], [0, 2, -0.9, 0], [0, 10, -3, 0]]]
    )
    expected = np.array([[1.0, 2.0, 0], [2.0, 3.0, 0.2], [0.3, 0.4, 0.6], [3, 0, 1.0]])
    w = warp.Warp((3, 4), normalize_t=True)
    got, _ = w.warp_event(torch.from_numpy(events), torch.from_numpy(flow), "dense-flow")
    assert torch.allclose(got, torch.from_numpy(expected))
    save("ka_warp_dense", events=events, flow=flow, expected=expected, image_size=np.array([3, 4]))

    # tests/test_event_image_converter.py:17-42 and 45-69
    imager = event_image_converter.EventImageConverter((3, 4))
    ev_i = np.array([[1.0, 2], [0, 1], [1, 0]])
    w_i = np.array([1, 2, 0.8])
    exp_i = np.array([[0, 2, 0, 0], [0.8, 0, 1, 0], [0, 0, 0, 0]])
    got = imager.bilinear_vote_tensor(torch.from_numpy(ev_i), weight=torch.from_numpy(w_i))
    assert torch.allclose(got, torch.from_numpy(exp_i))
    ev_f = np.array([[1.2, 2], [0, 1.9], [0.5, 0.6]])
    w_f = np.array([-1.0, 1.0, 1.5])
    exp_f = np.array([[0.3, 0.55, 0.9, 0], [0.3, 0.45, -0.8, 0], [0, 0, -0.2, 0]])
    got = imager.bilinear_vote_tensor(torch.from_numpy(ev_f), weight=torch.from_numpy(w_f))
    assert torch.allclose(got, torch.from_numpy(exp_f))
    save("ka_vote", ev_int=ev_i, w_int=w_i, exp_int=exp_i, ev_float=ev_f, w_float=w_f, exp_float=exp_f,
         image_size=np.array([3, 4]))


def gen_warps(rng):
    H, W = 26, 34
    ev = make_events(800, H, W, rng)
    ev_frac = make_events(300, H, W, rng, fractional=True)
    theta = np.array([12.3, -7.7])
    flow = rng.uniform(-5, 5, (2, H, W))
    T = 10
    voxel = utils.construct_dense_flow_voxel_numpy(smooth_flow(H, W, rng, 6.0), T, "burgers", "middle")
    out = {"events": ev, "events_frac": ev_frac, "theta": theta, "flow": flow, "voxel": voxel,
           "image_size": np.array([H, W])}
    w = warp.Warp((H, W), normalize_t=True)
    w_raw = warp.Warp((H, W), normalize_t=False)
    for d in ["first", "middle", "last", 0.3]:
        tag = d if isinstance(d, str) else "f0p3"
        for name, e in (("int", ev), ("frac", ev_frac)):
            te = torch.from_numpy(e)
            out[f"2dof_{name}_{tag}"] = w.warp_event(te, torch.from_numpy(theta), "2d-translation", d)[0].numpy()
            out[f"dense_{name}_{tag}"] = w.warp_event(te, torch.from_numpy(flow), "dense-flow", d)[0].numpy()
            out[f"voxel_{name}_{tag}"] = w.warp_event(te, torch.from_numpy(voxel), "dense-flow-voxel", d)[0].numpy()
        out[f"2dof_int_raw_{tag}"] = w_raw.warp_event(torch.from_numpy(ev), torch.from_numpy(theta), "2d-translation", d)[0].numpy()
    # numpy branch agrees (sanity, not stored)
    np.testing.assert_allclose(w.warp_event(ev, flow, "dense-flow", "middle")[0], out["dense_int_middle"], rtol=0, atol=1e-12)
    save("warp", **out)


def gen_votes(rng):
    H, W = 20, 30
    n = 1500
    # coordinates spread beyond the image on every side (-3 .. H+2) incl. exact integers and negatives
    xy = np.stack([rng.uniform(-3, H + 2, n), rng.uniform(-3, W + 2, n)], axis=1)
    xy[:100] = np.round(xy[:100])
    ev = np.concatenate([xy, np.zeros((n, 2))], axis=1)
    wt = rng.uniform(-1, 2, n)
    out = {"events": ev, "weight": wt, "image_size": np.array([H, W])}
    for pad in (0, 3):
        imager = event_image_converter.EventImageConverter((H, W), outer_padding=pad)
        te = torch.from_numpy(ev)
        out[f"vote_pad{pad}"] = imager.bilinear_vote_tensor(te).numpy()
        out[f"vote_w_pad{pad}"] = imager.bilinear_vote_tensor(te, weight=torch.from_numpy(wt)).numpy()
        # count_event_tensor raises on current torch (int64 `vals` scatter-added into a float image,
        # event_image_converter.py:251-254) -> the numpy branch is the only runnable count path
        out[f"count_pad{pad}"] = imager.count_event_numpy(ev)
        out[f"vote_numpy_pad{pad}"] = imager.bilinear_vote_numpy(ev)  # eps = 1e-8 branch
        out[f"mask_pad{pad}"] = imager.create_eventmask(te).numpy()
        for sigma in (1, 0.7):
            out[f"iwe_s{sigma}_pad{pad}"] = imager.create_iwe(te, "bilinear_vote", sigma).numpy()
        # autograd of the vote w.r.t. coordinates and weights, random cotangent
        G = rng.normal(size=out[f"vote_pad{pad}"].shape)
        te_g = torch.from_numpy(ev).requires_grad_()
        tw_g = torch.from_numpy(wt).requires_grad_()
        img = imager.bilinear_vote_tensor(te_g, weight=tw_g)
        ge, gw = torch.autograd.grad((img * torch.from_numpy(G)).sum(), [te_g, tw_g])
        out[f"G_pad{pad}"] = G
        out[f"gxy_pad{pad}"] = ge.numpy()[:, :2]
        out[f"gw_pad{pad}"] = gw.numpy()
    save("vote", **out)


def gen_costs(rng):
    H, W = 26, 34
    ev = make_events(3000, H, W, rng, fractional=True)
    imager = event_image_converter.EventImageConverter((H, W))
    iwe = imager.create_iwe(torch.from_numpy(ev), "bilinear_vote", 0).numpy()
    iwe2 = imager.create_iwe(torch.from_numpy(make_events(3000, H, W, rng, fractional=True)), "bilinear_vote", 1).numpy()
    iwe3 = imager.create_iwe(torch.from_numpy(make_events(2500, H, W, rng, fractional=True)), "bilinear_vote", 1).numpy()
    orig = imager.create_iwe(torch.from_numpy(make_events(3000, H, W, rng)), "bilinear_vote", 1).numpy()
    out = {"iwe": iwe, "iwe2": iwe2, "iwe3": iwe3, "orig": orig}
    kw = dict(store_history=False, precision="64", cuda_available=False)
    for name in ["image_variance", "gradient_magnitude", "normalized_image_variance", "normalized_gradient_magnitude",
                 "multi_focal_normalized_image_variance", "multi_focal_normalized_gradient_magnitude"]:
        for direction in ["minimize", "natural", "maximize"]:
            for omit in (True, False):
                c = costs.functions[name](direction=direction, **kw)
                t = {k: torch.from_numpy(v).requires_grad_() for k, v in
                     (("iwe", iwe), ("forward_iwe", iwe2), ("middle_iwe", iwe3), ("orig_iwe", orig))}
                arg = {"iwe": t["iwe"], "backward_iwe": t["iwe"], "forward_iwe": t["forward_iwe"],
                       "middle_iwe": t["middle_iwe"], "orig_iwe": t["orig_iwe"], "omit_boundary": omit}
                loss = c.calculate(arg)
                used = [k for k in ("iwe", "forward_iwe", "middle_iwe") if k in c.required_keys or
                        (k == "iwe" and "backward_iwe" in c.required_keys)]
                gs = torch.autograd.grad(loss, [t[k] for k in used], allow_unused=True)
                tag = f"{name}__{direction}__omit{int(omit)}"
                out[tag + "__loss"] = np.array(loss.item())
                for k, g in zip(used, gs):
                    out[tag + "__g_" + k] = g.numpy() if g is not None else np.zeros_like(iwe)
    # numpy-branch variance (biased) for the ddof=0 path
    out["image_variance_numpy__minimize__omit1"] = np.array(
        costs.functions["image_variance"](direction="minimize").calculate({"iwe": iwe, "omit_boundary": True}))
    # total variation on a patch-flow array
    for shape in ((2, 4, 4), (2, 8, 8), (2, 2, 2), (2, 1, 1)):
        fl = rng.uniform(-3, 3, shape)
        for omit in (True, False):
            c = costs.functions["total_variation"](direction="minimize", **kw)
            tf = torch.from_numpy(fl).requires_grad_()
            loss = c.calculate({"flow": tf, "omit_boundary": omit})
            (g,) = torch.autograd.grad(loss, tf)
            tag = f"tv_{shape[1]}x{shape[2]}_omit{int(omit)}"
            out[tag + "__flow"] = fl
            out[tag + "__loss"] = np.array(loss.item())
            out[tag + "__g"] = g.numpy()
    save("costs", **out)


def gen_flow_voxel(rng):
    H, W = 16, 20
    out = {}
    flows = {"rand": rng.uniform(-1, 1, (2, H, W)), "smooth": smooth_flow(H, W, rng, 8.0)}
    flows["withzeros"] = flows["smooth"].copy()
    flows["withzeros"][:, 5:9, 7:12] = 0.0  # exercises sign(0) / max-min ties
    for fname, fl in flows.items():
        out[f"flow_{fname}"] = fl
        for dt in (0.1, -0.1, 0.01, -0.037, 0.0):
            tag = f"{fname}_dt{dt}"
            for scheme, fn in (("burgers", utils.inviscid_burger_flow_to_voxel_torch), ("upwind", utils.upwind_flow_to_voxel_torch)):
                tf = torch.from_numpy(fl).requires_grad_()
                o = fn(tf, dt, 1, 1)
                out[f"{scheme}_step_{tag}"] = o.detach().numpy()
                if dt != 0.0:
                    cot = rng.normal(size=(2, H, W))
                    (g,) = torch.autograd.grad((o * torch.from_numpy(cot)).sum(), tf)
                    out[f"{scheme}_cot_{tag}"] = cot
                    out[f"{scheme}_vjp_{tag}"] = g.numpy()
        for scheme in ("burgers", "upwind"):
            if fname == "rand":
                continue
            for T, loc in ((10, "middle"), (5, "middle"), (4, "first")):
                tf = torch.from_numpy(fl).requires_grad_()
                V = utils.construct_dense_flow_voxel_torch(tf, T, scheme, loc)
                Vn = utils.construct_dense_flow_voxel_numpy(fl, T, scheme, loc)
                np.testing.assert_allclose(V.detach().numpy(), Vn, rtol=0, atol=1e-10)
                cot = rng.normal(size=tuple(V.shape))
                (g,) = torch.autograd.grad((V * torch.from_numpy(cot)).sum(), tf)
                tag = f"{scheme}_{fname}_T{T}_{loc}"
                out[f"voxel_{tag}"] = V.detach().numpy()
                out[f"voxel_cot_{tag}"] = cot
                out[f"voxel_vjp_{tag}"] = g.numpy()
    save("flow_voxel", **out)


def _fake_solver(H, W, cost_name, sigma, pad=0, cost_with_weight=None, method="bilinear_vote"):
    """The attributes get_arg_for_cost / calculate_cost read (patch_contrast_base.py:273-352),
    built exactly as SolverBase.__init__ builds them (solver/base.py:139-147, 177-209)."""
    kw = dict(direction="minimize", store_history=False, image_size=(H + pad, W + pad), percentile=1.0,
              precision="64", cuda_available=False)
    if cost_name == "hybrid":
        cf = costs.HybridCost(cost_with_weight=cost_with_weight, **kw)
    else:
        cf = costs.functions[cost_name](**kw)
    return types.SimpleNamespace(
        cost_func=cf,
        imager=event_image_converter.EventImageConverter((H, W), outer_padding=pad),
        warper=warp.Warp((H, W), calculate_feature=True, normalize_t=True),
        iwe_config={"method": method, "blur_sigma": sigma},
    )


def _ref_objective(fs, ev, motion, model, coarse):
    te = torch.from_numpy(ev)
    tm = torch.from_numpy(motion).requires_grad_()
    tc = torch.from_numpy(coarse).requires_grad_() if coarse is not None else None
    arg = PatchContrastMaximization.get_arg_for_cost(fs, te, tm, model, tc)
    loss = fs.cost_func.calculate(arg)
    ins = [tm] + ([tc] if tc is not None and "flow" in fs.cost_func.required_keys else [])
    gs = torch.autograd.grad(loss, ins, allow_unused=True)
    iwes = {k: v.detach().numpy() for k, v in arg.items() if k.endswith("iwe")}
    return loss.item(), gs, iwes


def gen_objectives(rng):
    out = {}
    H, W = 32, 40
    ev = make_events(3000, H, W, rng)
    theta = np.array([23.4, -17.7])  # large enough to push events out of the image
    flow_rand = rng.uniform(-5, 5, (2, H, W))
    flow_smooth = smooth_flow(H, W, rng, 20.0)
    T = 10
    voxel = utils.construct_dense_flow_voxel_numpy(smooth_flow(H, W, rng, 12.0), T, "burgers", "middle")
    coarse = rng.uniform(-3, 3, (2, 4, 4))
    out.update(events=ev, theta=theta, flow_rand=flow_rand, flow_smooth=flow_smooth, voxel=voxel, coarse=coarse,
               image_size=np.array([H, W]))
    motions = {"2dof": ("2d-translation", theta), "dense_rand": ("dense-flow", flow_rand),
               "dense_smooth": ("dense-flow", flow_smooth), "voxel": ("dense-flow-voxel", voxel)}
    yaml_hybrid = {"multi_focal_normalized_gradient_magnitude": 1.0, "total_variation": 0.01}
    cases = [(c, s, None) for c in ("image_variance", "gradient_magnitude") for s in (0, 1)]
    cases += [(c, 1, None) for c in ("normalized_image_variance", "normalized_gradient_magnitude",
                                     "multi_focal_normalized_image_variance",
                                     "multi_focal_normalized_gradient_magnitude")]
    cases += [("hybrid", 1, yaml_hybrid)]
    for mname, (model, motion) in motions.items():
        for cost_name, sigma, cww in cases:
            fs = _fake_solver(H, W, cost_name, sigma, cost_with_weight=cww)
            loss, gs, iwes = _ref_objective(fs, ev, motion, model, coarse)
            tag = f"{mname}__{cost_name}__s{sigma}"
            out[tag + "__loss"] = np.array(loss)
            out[tag + "__grad"] = gs[0].numpy()
            if len(gs) > 1 and gs[1] is not None:
                out[tag + "__grad_coarse"] = gs[1].numpy()
            if cost_name in ("image_variance", "multi_focal_normalized_gradient_magnitude") and mname in ("2dof", "dense_smooth", "voxel"):
                for k, v in iwes.items():
                    out[tag + "__" + k] = v
    # padding + count method + fractional source coordinates
    evf = make_events(1500, H, W, rng, fractional=True)
    out["events_frac"] = evf
    for pad in (0, 4):
        fs = _fake_solver(H, W, "image_variance", 1, pad=pad)
        loss, gs, iwes = _ref_objective(fs, evf, theta, "2d-translation", None)
        out[f"frac_pad{pad}__loss"] = np.array(loss)
        out[f"frac_pad{pad}__grad"] = gs[0].numpy()
        out[f"frac_pad{pad}__iwe"] = iwes["iwe"]
    save("objective", **out)


def gen_hvp(rng):
    H, W = 32, 40
    ev = make_events(2000, H, W, rng)
    fs = _fake_solver(H, W, "image_variance", 1)
    te = torch.from_numpy(ev)

    def f(m):
        arg = PatchContrastMaximization.get_arg_for_cost(fs, te, m, "2d-translation", None)
        return fs.cost_func.calculate(arg)

    theta = torch.tensor([13.4, -7.7], dtype=torch.float64)
    v = torch.tensor([0.3, 1.1], dtype=torch.float64)
    loss, hv = torch.autograd.functional.vhp(f, theta, v)
    save("hvp", events=ev, theta=theta.numpy(), v=v.numpy(), loss=np.array(loss.item()), vhp=hv.numpy(),
         image_size=np.array([H, W]))


def gen_solver_objective(rng):
    """Whole solver objective (the optimiser's `fun`): PyramidalPatchContrastMaximization.objective_scipy
    (src/solver/patch_contrast_pyramid.py:430-462) = patch -> dense interpolation (+ Burgers voxel) ->
    get_arg_for_cost -> hybrid cost, with the shipped YAML parameters on a reduced image."""
    from src import solver as ref_solver

    H, W = 68, 90
    out = {"image_size": np.array([H, W])}
    ev = make_events(5000, H, W, rng)
    out["events"] = ev
    for tag, time_aware in (("plain", False), ("burgers", True)):
        slv_cfg = {
            "method": "pyramidal_patch_contrast_maximization", "time_aware": time_aware,
            "patch": {"initialize": "random", "scale": 4, "crop_height": 64, "crop_width": 80, "filter_type": "bilinear"},
            "motion_model": "2d-translation", "warp_direction": "first", "parameters": ["trans_x", "trans_y"],
            "cost": "hybrid", "outer_padding": 0,
            "cost_with_weight": {"multi_focal_normalized_gradient_magnitude": 1.0, "total_variation": 0.01},
            "iwe": {"method": "bilinear_vote", "blur_sigma": 1},
        }
        if time_aware:
            slv_cfg.update({"time_bin": 10, "flow_interpolation": "burgers", "t0_flow_location": "middle"})
        opt_cfg = {"n_iter": 40, "method": "Newton-CG", "max_iter": 25,
                   "parameters": {"trans_x": {"min": -150, "max": 150}, "trans_y": {"min": -150, "max": 150}}}
        slv = ref_solver.collections["pyramidal_patch_contrast_maximization"]((H, W), {}, slv_cfg, opt_cfg, {}, None)
        slv._device = "cpu"
        te = torch.from_numpy(ev)
        for scale in (1, 3):
            slv.overload_patch_configuration(scale)
            ph, pw = slv.patch_image_size
            x = rng.uniform(-300, 300, 2 * ph * pw)  # pixel / second: t_scale = 0.05 s -> +-15 px over the batch
            tx = torch.from_numpy(x).requires_grad_()
            loss = slv.objective_scipy(tx, te, {}, suppress_log=True)
            (g,) = torch.autograd.grad(loss, tx)
            dense = slv.interpolate_dense_flow_from_patch_tensor(torch.from_numpy(x))
            k = f"{tag}_s{scale}"
            out[k + "__x"] = x
            out[k + "__loss"] = np.array(loss.item())
            out[k + "__grad"] = g.numpy()
            out[k + "__dense"] = dense.numpy()
            out[k + "__patch_image_size"] = np.array([ph, pw])
            out[k + "__patch_size"] = np.array(slv.patch_size)
            out[k + "__sliding_window"] = np.array(slv.sliding_window)
        out[tag + "__patch_shift"] = np.array(slv.patch_shift)
    # vhp of the plain objective at the coarsest scale (Newton-CG's hessp)
    slv.overload_patch_configuration(1)
    save("solver_objective", **out)


def gen_solver_hvp():
    """vhp of the whole solver objective (Newton-CG's hessp through TorchWrapper.get_hvp, torch_wrapper.py:51-73) at
    the inputs of solver_objective.npz, plain and Burgers -> solver_hvp.npz.  The inputs are READ from that fixture."""
    from src import solver as ref_solver

    g = np.load(os.path.join(HERE, "solver_objective.npz"))
    H, W = (int(v) for v in g["image_size"])
    ev = g["events"]
    te = torch.from_numpy(ev)
    rng = np.random.default_rng(SEED + 3)
    out = {}
    for tag, time_aware in (("plain", False), ("burgers", True)):
        slv_cfg = {
            "method": "pyramidal_patch_contrast_maximization", "time_aware": time_aware,
            "patch": {"initialize": "random", "scale": 4, "crop_height": 64, "crop_width": 80, "filter_type": "bilinear"},
            "motion_model": "2d-translation", "warp_direction": "first", "parameters": ["trans_x", "trans_y"],
            "cost": "hybrid", "outer_padding": 0,
            "cost_with_weight": {"multi_focal_normalized_gradient_magnitude": 1.0, "total_variation": 0.01},
            "iwe": {"method": "bilinear_vote", "blur_sigma": 1},
        }
        if time_aware:
            slv_cfg.update({"time_bin": 10, "flow_interpolation": "burgers", "t0_flow_location": "middle"})
        opt_cfg = {"n_iter": 40, "method": "Newton-CG", "max_iter": 25,
                   "parameters": {"trans_x": {"min": -150, "max": 150}, "trans_y": {"min": -150, "max": 150}}}
        slv = ref_solver.collections["pyramidal_patch_contrast_maximization"]((H, W), {}, slv_cfg, opt_cfg, {}, None)
        slv._device = "cpu"
        for scale in (1, 3):
            slv.overload_patch_configuration(scale)
            k = f"{tag}_s{scale}"
            x = torch.from_numpy(g[k + "__x"])
            v = torch.from_numpy(rng.normal(size=x.shape))
            loss, hv = torch.autograd.functional.vhp(lambda z: slv.objective_scipy(z, te, {}, suppress_log=True), x, v)
            assert abs(loss.item() - float(g[k + "__loss"])) <= 1e-12 * abs(loss.item())
            out[k + "__v"] = v.numpy()
            out[k + "__vhp"] = hv.numpy()
    save("solver_hvp", **out)


def gen_blur_numpy():
    """numpy-branch create_iwe(sigma>0) = scipy gaussian_filter (real scipy, no shim) -> blur_numpy.npz"""
    rng = np.random.default_rng(47)
    H, W = 20, 30
    xy = np.stack([rng.uniform(-1, H, 800), rng.uniform(-1, W, 800)], 1)
    ev = np.concatenate([xy, np.zeros((800, 2))], 1)
    im = event_image_converter.EventImageConverter((H, W))
    out = {"events": ev, "image_size": np.array([H, W])}
    for s in (1, 2, 0.6):
        out[f"iwe_numpy_s{s}"] = im.create_iwe(ev, "bilinear_vote", s)
    out["shims"] = np.array("none (scipy.ndimage.gaussian_filter is the real one)")
    out["seed"] = np.array(47)
    np.savez_compressed(os.path.join(HERE, "blur_numpy.npz"), **out)


def gen_hvp_cases():
    """vhp of the objective w.r.t. the motion (torch.autograd.functional.vhp, what Newton-CG receives,
    src/solver/scipy_autograd/torch_wrapper.py:51-73) for the inputs of objective.npz -> hvp_cases.npz"""
    g = dict(np.load(os.path.join(HERE, "objective.npz")))
    H, W = (int(v) for v in g["image_size"])
    te = torch.from_numpy(g["events"])
    rng = np.random.default_rng(SEED + 2)
    out = {}
    cases = [("2dof", "2d-translation", "theta", c, s) for c, s in
             (("image_variance", 0), ("image_variance", 1), ("gradient_magnitude", 1), ("normalized_image_variance", 1),
              ("multi_focal_normalized_gradient_magnitude", 1))]
    cases += [("dense_smooth", "dense-flow", "flow_smooth", c, 1) for c in
              ("image_variance", "gradient_magnitude", "multi_focal_normalized_gradient_magnitude")]
    cases += [("voxel", "dense-flow-voxel", "voxel", "image_variance", 1)]
    for mname, model, mkey, cost_name, sigma in cases:
        fs = _fake_solver(H, W, cost_name, sigma)

        def f(m):
            arg = PatchContrastMaximization.get_arg_for_cost(fs, te, m, model, None)
            return fs.cost_func.calculate(arg)

        x = torch.from_numpy(g[mkey])
        v = torch.from_numpy(rng.normal(size=x.shape))
        loss, hv = torch.autograd.functional.vhp(f, x, v)
        tag = f"{mname}__{cost_name}__s{sigma}"
        out[tag + "__v"] = v.numpy()
        out[tag + "__vhp"] = hv.numpy()
        out[tag + "__loss"] = np.array(loss.item())
        print(tag, float(loss), float(hv.abs().max()))
    out["shims"] = np.array(ref_import.SHIMS)
    out["seed"] = np.array(SEED + 2)
    np.savez_compressed(os.path.join(HERE, "hvp_cases.npz"), **out)


def gen_patch_search():
    """Small-patch cost of the finer-scale re-initialisation: PyramidalPatchContrastMaximization.
    calculate_cost_for_small_patch on the cropped, origin-shifted events of a patch, called as objective_initial does
    (src/solver/patch_contrast_pyramid.py:320-414): NormalizedGradientMagnitude on numpy arrays (cv2.Sobel: shim (3))."""
    from src import solver as ref_solver

    rng = np.random.default_rng(SEED + 5)
    H, W = 68, 90
    # moving point features (so that candidates differ in contrast) + uniform noise events
    n_feat, per_feat, vel, span = 150, 40, np.array([140.0, -90.0]), 0.05
    p0 = np.stack([rng.uniform(0, H, n_feat), rng.uniform(0, W, n_feat)], axis=1)
    t = rng.uniform(0, span, (n_feat, per_feat))
    xy = p0[:, None, :] + vel[None, None, :] * t[..., None]
    feat = np.concatenate([np.floor(xy.reshape(-1, 2)), t.reshape(-1, 1), np.ones((n_feat * per_feat, 1))], axis=1)
    feat = feat[(feat[:, 0] >= 0) & (feat[:, 0] < H) & (feat[:, 1] >= 0) & (feat[:, 1] < W)]
    ev = np.concatenate([feat, make_events(1500, H, W, rng)], axis=0)
    ev = ev[np.argsort(ev[:, 2], kind="stable")]
    out = {"image_size": np.array([H, W]), "events": ev, "velocity": vel}
    slv_cfg = {
        "method": "pyramidal_patch_contrast_maximization", "time_aware": False,
        "patch": {"initialize": "random", "scale": 4, "crop_height": 64, "crop_width": 80, "filter_type": "bilinear"},
        "motion_model": "2d-translation", "warp_direction": "first", "parameters": ["trans_x", "trans_y"],
        "cost": "hybrid", "outer_padding": 0,
        "cost_with_weight": {"multi_focal_normalized_gradient_magnitude": 1.0, "total_variation": 0.01},
        "iwe": {"method": "bilinear_vote", "blur_sigma": 1},
    }
    opt_cfg = {"n_iter": 40, "method": "Newton-CG", "max_iter": 25,
               "parameters": {"trans_x": {"min": -150, "max": 150}, "trans_y": {"min": -150, "max": 150}}}
    slv = ref_solver.collections["pyramidal_patch_contrast_maximization"]((H, W), {}, slv_cfg, opt_cfg, {}, None)
    for scale in (2, 3):
        slv.overload_patch_configuration(scale)
        n_patch, n_cand = slv.n_patch, 6
        boxes = np.array([[slv.patches[i].x_min, slv.patches[i].x_max, slv.patches[i].y_min, slv.patches[i].y_max]
                          for i in range(n_patch)])
        cand = rng.uniform(-250, 250, (n_patch, n_cand, 2))
        cand[:, 0] = vel  # the true motion
        cand[:, 1] = 0.0
        loss = np.zeros((n_patch, n_cand))
        count = np.zeros(n_patch, dtype=np.int64)
        for i in range(n_patch):
            fe = utils.crop_event(ev, *boxes[i])
            fe = utils.set_event_origin_to_zero(np.copy(fe), boxes[i][0], boxes[i][2], 0)
            count[i] = len(fe)
            for c in range(n_cand):
                if len(fe) == 0:
                    loss[i, c] = np.nan
                    continue
                motion_array = np.array(cand[i, c])
                motion_array *= np.max(fe[:, 2]) - np.min(fe[:, 2])  # objective_initial, lines 359-361
                loss[i, c] = slv.calculate_cost_for_small_patch(np.copy(fe), motion_array, "2d-translation")
        k = f"s{scale}"
        out[k + "__boxes"] = boxes
        out[k + "__patch_size"] = np.array(slv.patch_size)
        out[k + "__cand"] = cand
        out[k + "__loss"] = loss
        out[k + "__count"] = count
    out["sigma"] = np.array(1.0)
    save("patch_search", **out)


def _ref_pyramid_solver(H, W, time_aware, scale=4, crop=(64, 80), cost=None):
    """The reference's PyramidalPatchContrastMaximization with the shipped YAML parameters
    (configs/mvsec_indoor_no_timeaware.yaml / mvsec_indoor_burgers.yaml) on an H x W sensor."""
    from src import solver as ref_solver

    slv_cfg = {
        "method": "pyramidal_patch_contrast_maximization", "time_aware": time_aware,
        "patch": {"initialize": "random", "scale": scale, "crop_height": crop[0], "crop_width": crop[1], "filter_type": "bilinear"},
        "motion_model": "2d-translation", "warp_direction": "first", "parameters": ["trans_x", "trans_y"],
        "cost": "hybrid", "outer_padding": 0,
        "cost_with_weight": {"multi_focal_normalized_gradient_magnitude": 1.0, "total_variation": 0.01},
        "iwe": {"method": "bilinear_vote", "blur_sigma": 1},
    }
    if cost is not None:  # the YAML with `cost:` overridden (BASELINE configs[0] says "variance cost")
        slv_cfg["cost"] = cost
        slv_cfg.pop("cost_with_weight")
    if time_aware:
        slv_cfg.update({"time_bin": 10, "flow_interpolation": "burgers", "t0_flow_location": "middle"})
    opt_cfg = {"n_iter": 40, "method": "Newton-CG", "max_iter": 25,
               "parameters": {"trans_x": {"min": -150, "max": 150}, "trans_y": {"min": -150, "max": 150}}}
    slv = ref_solver.collections["pyramidal_patch_contrast_maximization"]((H, W), {}, slv_cfg, opt_cfg, {}, None)
    slv._device = "cpu"
    return slv


def _moving_dot_events(n, H, W, flow_fn, rng, n_dots, period=0.05, jitter=0.5):
    """Events of dots that move with the displacement field flow_fn(x, y) -> (vx, vy) in pixel per batch period:
    an event emitted at time t sits at centre + (t / period) * v(centre).  Integer pixel coordinates, sorted in t."""
    t = np.sort(rng.uniform(0.0, period, n))
    dot = rng.integers(0, n_dots, n)
    cx, cy = rng.uniform(4, H - 4, n_dots), rng.uniform(4, W - 4, n_dots)
    vx, vy = flow_fn(cx, cy)
    x = cx[dot] + (t / period) * vx[dot] + rng.normal(0, jitter, n)
    y = cy[dot] + (t / period) * vy[dot] + rng.normal(0, jitter, n)
    ev = np.empty((n, 4))
    ev[:, 0] = np.clip(np.round(x), 0, H - 1)
    ev[:, 1] = np.clip(np.round(y), 0, W - 1)
    ev[:, 2] = t
    ev[:, 3] = rng.integers(0, 2, n)
    return ev


def gen_solver_optimize():
    """An OPTIMISER RESULT of the reference: run_scipy at the coarsest scale (src/solver/patch_contrast_pyramid.py:252-318
    with scale 1: seeded start, no Optuna, no skimage) = scipy_autograd.minimize(objective_scipy, x0, "Newton-CG",
    {gtol 1e-5, maxiter 25, eps 0.01}, float64) -> solver_optimize.npz: x0, final x, final loss, iteration counts, for the
    plain and the Burgers YAML objective.  Scene: dots moving with a smooth 2 x 2-patch flow (so the optimum is meaningful)."""
    from src.solver import scipy_autograd

    rng = np.random.default_rng(SEED + 5)
    H, W = 68, 90
    period = 0.05

    def flow_fn(cx, cy):  # pixel per batch period, smooth across the sensor
        return 6.0 + 4.0 * cx / H - 2.0 * cy / W, -5.0 + 3.0 * cy / W + 2.0 * cx / H

    ev = _moving_dot_events(20000, H, W, flow_fn, rng, n_dots=150, period=period)
    out = {"events": ev, "image_size": np.array([H, W]), "period": np.array(period), "seed": np.array(SEED + 5),
           "shims": np.array(ref_import.SHIMS)}
    te = torch.from_numpy(ev)
    for tag, time_aware in (("plain", False), ("burgers", True)):
        slv = _ref_pyramid_solver(H, W, time_aware)
        slv.overload_patch_configuration(1)
        ph, pw = slv.patch_image_size
        # the reference's x is pixel / second, and the flow it describes is the NEGATIVE scene velocity
        # (interpolate_dense_flow_from_patch_tensor negates): start within +-30 % of (-8, 3) px per batch on every patch.
        # With this start the reference's own run is stable -- perturbing x0 by 1e-7 relative moves its end point by
        # < 0.005 px and its final loss by < 4e-6 relative (checked when the fixture was made); a start far from the
        # optimum is not (the objective has kinks at every pixel-cell border, Newton-CG's line search amplifies them)
        base = -np.array([[8.0] * (ph * pw), [-3.0] * (ph * pw)]).reshape(-1) / period
        x0 = base * (1.0 + 0.3 * rng.uniform(-1.0, 1.0, 2 * ph * pw))
        res = scipy_autograd.minimize(lambda x: slv.objective_scipy(x, te, {}, suppress_log=True), x0, method="Newton-CG",
                                      options={"gtol": 1e-5, "disp": False, "maxiter": 25, "eps": 0.01}, precision="float64",
                                      torch_device="cpu")
        dense = slv.interpolate_dense_flow_from_patch_tensor(torch.from_numpy(np.asarray(res.x).reshape(-1)))
        out[tag + "__x0"] = x0
        out[tag + "__x"] = np.asarray(res.x).reshape(-1)
        out[tag + "__loss0"] = np.array(slv.objective_scipy(torch.from_numpy(x0), te, {}, suppress_log=True).item())
        out[tag + "__loss"] = np.array(float(res.fun))
        out[tag + "__nit"] = np.array(int(res.nit))
        out[tag + "__nfev"] = np.array(int(res.nfev))
        out[tag + "__dense"] = dense.numpy()  # pixel / second
        out[tag + "__patch_image_size"] = np.array([ph, pw])
        out[tag + "__patch_size"] = np.array(slv.patch_size)
        out[tag + "__sliding_window"] = np.array(slv.sliding_window)
        out[tag + "__patch_shift"] = np.array(slv.patch_shift)
        print(tag, "loss0", float(out[tag + "__loss0"]), "->", float(res.fun), "nit", res.nit, "nfev", res.nfev, "x", np.round(res.x * period, 2))
    save("solver_optimize", **out)


def gen_solver_objective_cfg1():
    """BASELINE configs[0] at ITS size: 260 x 346 sensor, 30 000 events, the shipped YAML (patch scale 4 -> 16 x 16
    patches at the finest scale, hybrid cost): value and gradient of objective_scipy at scales 1 (coarsest) and 4
    (finest), plain and Burgers -> solver_objective_cfg1.npz"""
    rng = np.random.default_rng(SEED + 6)
    H, W = 260, 346

    def flow_fn(cx, cy):
        return 8.0 * np.sin(cx / 60.0) + 3.0, -6.0 * np.cos(cy / 80.0)

    ev = _moving_dot_events(30000, H, W, flow_fn, rng, n_dots=600)
    out = {"events": ev, "image_size": np.array([H, W]), "seed": np.array(SEED + 6), "shims": np.array(ref_import.SHIMS)}
    te = torch.from_numpy(ev)
    for tag, time_aware in (("plain", False), ("burgers", True)):
        slv = _ref_pyramid_solver(H, W, time_aware, scale=5, crop=(256, 336))  # configs/mvsec_indoor_*.yaml: scale 5, crop 256 x 336
        for scale in (1, 4):
            slv.overload_patch_configuration(scale)
            ph, pw = slv.patch_image_size
            x = rng.uniform(-200, 200, 2 * ph * pw)  # pixel / second: +-10 px over the 0.05 s batch
            tx = torch.from_numpy(x).requires_grad_()
            loss = slv.objective_scipy(tx, te, {}, suppress_log=True)
            (g,) = torch.autograd.grad(loss, tx)
            k = f"{tag}_s{scale}"
            out[k + "__x"] = x
            out[k + "__loss"] = np.array(loss.item())
            out[k + "__grad"] = g.numpy()
            out[k + "__patch_image_size"] = np.array([ph, pw])
            out[k + "__patch_size"] = np.array(slv.patch_size)
            out[k + "__sliding_window"] = np.array(slv.sliding_window)
            print(k, "patches", ph, pw, "loss", loss.item())
        out[tag + "__patch_shift"] = np.array(slv.patch_shift)
    save("solver_objective_cfg1", **out)


def gen_solver_objective_cfg1_variance():
    """BASELINE configs[0] read literally: configs/mvsec_indoor_no_timeaware.yaml with `cost: image_variance` (the YAML
    ships `hybrid`; SURVEY 8d: "run both"), 260 x 346, 30 000 events, scales 1 and 4, plain and Burgers ->
    solver_objective_cfg1_variance.npz.  Same scene as solver_objective_cfg1 (its own RNG stream)."""
    rng = np.random.default_rng(SEED + 9)
    H, W = 260, 346

    def flow_fn(cx, cy):
        return 8.0 * np.sin(cx / 60.0) + 3.0, -6.0 * np.cos(cy / 80.0)

    ev = _moving_dot_events(30000, H, W, flow_fn, rng, n_dots=600)
    out = {"events": ev, "image_size": np.array([H, W]), "seed": np.array(SEED + 9), "shims": np.array(ref_import.SHIMS)}
    te = torch.from_numpy(ev)
    for tag, time_aware in (("plain", False), ("burgers", True)):
        slv = _ref_pyramid_solver(H, W, time_aware, scale=5, crop=(256, 336), cost="image_variance")
        for scale in (1, 4):
            slv.overload_patch_configuration(scale)
            ph, pw = slv.patch_image_size
            x = rng.uniform(-200, 200, 2 * ph * pw)
            tx = torch.from_numpy(x).requires_grad_()
            loss = slv.objective_scipy(tx, te, {}, suppress_log=True)
            (g,) = torch.autograd.grad(loss, tx)
            k = f"{tag}_s{scale}"
            out[k + "__x"] = x
            out[k + "__loss"] = np.array(loss.item())
            out[k + "__grad"] = g.numpy()
            out[k + "__patch_image_size"] = np.array([ph, pw])
            out[k + "__patch_size"] = np.array(slv.patch_size)
            out[k + "__sliding_window"] = np.array(slv.sliding_window)
            print(k, "patches", ph, pw, "loss", loss.item())
        out[tag + "__patch_shift"] = np.array(slv.patch_shift)
    save("solver_objective_cfg1_variance", **out)


def gen_cfg2_fp32_reference():
    """The reference's OWN torch path on the bench's headline stream (BASELINE configs[1]: 1M uniform events, seed 46,
    260 x 346, theta = (12.3, -7.7), image_variance, sigma 0), evaluated in fp64 (the solver's dtype) AND in fp32:
    Warp.warp_event -> EventImageConverter.create_iwe -> ImageVariance.calculate -> torch.autograd.grad, nothing of ours in
    between.  Stored: both losses and gradients, the IWE's checksum, and how many events the fp32 run puts into a different
    cell than the fp64 run.  Backs the statement in tests/_border.py / DESIGN section 4: ANY fp32 evaluation of this
    objective -- the reference's included -- takes the derivative of cell-border events from the neighbouring cell, which
    moves a 2-DoF gradient by ~1e-3 relative; the HIP path decides those cells in fp64 (warp_one) and must match the fp64
    row.  The event stream is regenerated from its seed by the test (generator: event_based_optical_flow_amd/utils)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from event_based_optical_flow_amd.utils import generate_events
    from src import costs as ref_costs
    from src import event_image_converter as ref_eic
    from src import warp as ref_warp

    H, W, n = 260, 346, 1_000_000
    ev = generate_events(n, H, W, 0.0, 0.05, seed=46)
    theta = np.array([12.3, -7.7])
    out = {"image_size": np.array([H, W]), "n": np.array(n), "seed": np.array(46), "theta": theta,
           "events_checksum": np.array([ev[:, 0].sum(), ev[:, 1].sum(), ev[:, 2].sum()]), "shims": np.array(ref_import.SHIMS)}
    cells = {}
    for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
        te = torch.from_numpy(ev).to(dt)
        m = torch.tensor(theta, dtype=dt, requires_grad=True)
        warper = ref_warp.Warp((H, W), normalize_t=True)
        imager = ref_eic.EventImageConverter((H, W))
        cost = ref_costs.functions["image_variance"](direction="minimize")
        warped, _ = warper.warp_event(te, m, "2d-translation", direction="first")
        iwe = imager.create_iwe(warped, method="bilinear_vote", sigma=0)
        loss = cost.calculate({"iwe": iwe, "omit_boundary": True})
        (g,) = torch.autograd.grad(loss, m)
        out[tag + "__loss"] = np.array(loss.item())
        out[tag + "__grad"] = g.double().numpy()
        out[tag + "__iwe_sum"] = np.array(iwe.double().sum().item())
        w = warped.detach()
        cells[tag] = (torch.floor(w[:, 0] + 1e-6).long(), torch.floor(w[:, 1] + 1e-6).long())
        print(tag, "loss", loss.item(), "grad", g.tolist())
    moved = ((cells["f64"][0] != cells["f32"][0]) | (cells["f64"][1] != cells["f32"][1])).sum().item()
    out["events_in_another_cell_in_fp32"] = np.array(moved)
    g64, g32 = out["f64__grad"], out["f32__grad"]
    out["fp32_grad_rel_err"] = np.array(np.abs(g32 - g64).max() / np.abs(g64).max())
    print("events in another cell in fp32:", moved, " fp32 gradient error vs fp64:", float(out["fp32_grad_rel_err"]))
    save("cfg2_fp32_reference", **out)


def gen_warp_voxel_optimized():
    """a7: Warp.warp_event(..., "dense-flow-voxel-optimized", flow_propagate_bin=n) -> warp_voxel_optimized.npz.  The reference's
    function raises AttributeError at its first statement (`self.feature_base`, src/warp.py:422: Warp has feature_2dof / feature_dense
    only), so it is run here with that ONE attribute supplied (= feature_dense, whose calculate_feature(skip=True) is what the
    sibling voxel warp returns, src/warp.py:394-396) -- everything after that line is the reference's own code."""
    rng = np.random.default_rng(SEED + 10)
    H, W = 26, 34
    ev = make_events(900, H, W, rng)
    flow = smooth_flow(H, W, rng, 6.0)
    out = {"events": ev, "flow": flow, "image_size": np.array([H, W]), "seed": np.array(SEED + 10), "shims": np.array(ref_import.SHIMS),
           "note": np.array("Warp.feature_base = Warp.feature_dense supplied (missing attribute in the reference, src/warp.py:422)")}
    w = warp.Warp((H, W), normalize_t=True)
    w.feature_base = w.feature_dense
    for n_bin in (4, 10):
        for d in ["first", "middle", "last"]:
            a = w.warp_event(torch.from_numpy(ev), torch.from_numpy(flow), "dense-flow-voxel-optimized", d, flow_propagate_bin=n_bin)[0].numpy()
            b = w.warp_event(ev, flow, "dense-flow-voxel-optimized", d, flow_propagate_bin=n_bin)[0]
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-12)  # torch and numpy branches agree
            out[f"T{n_bin}_{d}"] = a
    save("warp_voxel_optimized", **out)


def gen_hvp_inv():
    """Hybrid costs with an "inv" weight (src/costs/hybrid.py:51-53: the term contributes 1 / cost): value, gradient and
    vhp of the objective w.r.t. the motion, inputs of objective.npz -> hvp_inv.npz"""
    g = dict(np.load(os.path.join(HERE, "objective.npz")))
    H, W = (int(v) for v in g["image_size"])
    te = torch.from_numpy(g["events"])
    rng = np.random.default_rng(SEED + 7)
    out = {}
    cases = [("2dof", "2d-translation", "theta", {"image_variance": "inv"}),
             ("2dof", "2d-translation", "theta", {"gradient_magnitude": 1.0, "image_variance": "inv"}),
             ("dense_smooth", "dense-flow", "flow_smooth", {"normalized_gradient_magnitude": "inv", "image_variance": 0.5})]
    for ci, (mname, model, mkey, cww) in enumerate(cases):
        fs = _fake_solver(H, W, "hybrid", 1, cost_with_weight=cww)

        def f(m):
            arg = PatchContrastMaximization.get_arg_for_cost(fs, te, m, model, None)
            return fs.cost_func.calculate(arg)

        x = torch.from_numpy(g[mkey])
        v = torch.from_numpy(rng.normal(size=x.shape))
        loss, hv = torch.autograd.functional.vhp(f, x, v)
        xg = x.clone().requires_grad_()
        (grad,) = torch.autograd.grad(f(xg), xg)
        tag = f"case{ci}"
        out[tag + "__model"] = np.array(model)
        out[tag + "__motion_key"] = np.array(mkey)
        out[tag + "__costs"] = np.array(list(cww.keys()))
        out[tag + "__weights"] = np.array([str(w) for w in cww.values()])
        out[tag + "__v"] = v.numpy()
        out[tag + "__vhp"] = hv.numpy()
        out[tag + "__grad"] = grad.numpy()
        out[tag + "__loss"] = np.array(loss.item())
        print(tag, model, cww, float(loss), float(hv.abs().max()))
    out["shims"] = np.array(ref_import.SHIMS)
    out["seed"] = np.array(SEED + 7)
    np.savez_compressed(os.path.join(HERE, "hvp_inv.npz"), **out)


def gen_costs_batched():
    """The base contrast costs on a STACK of images [B, H, W] (src/costs/image_variance.py:38-40 crops `[..., 1:-1, 1:-1]`
    and takes torch.var over every element; gradient_magnitude.py:62-75 treats the leading axis as the batch and takes
    the mean over everything) -> costs_batched.npz: loss and autograd gradient w.r.t. the stack."""
    rng = np.random.default_rng(SEED + 8)
    H, W = 22, 30
    imager = event_image_converter.EventImageConverter((H, W))
    stack = np.stack([imager.create_iwe(torch.from_numpy(make_events(n, H, W, rng, fractional=True)), "bilinear_vote", s).numpy()
                      for n, s in ((1500, 0), (2500, 1), (900, 0))])
    out = {"stack": stack}
    kw = dict(store_history=False, precision="64", cuda_available=False)
    for name in ("image_variance", "gradient_magnitude"):
        for direction in ("minimize", "maximize"):
            for omit in (True, False):
                c = costs.functions[name](direction=direction, **kw)
                t = torch.from_numpy(stack).requires_grad_()
                loss = c.calculate({"iwe": t, "omit_boundary": omit})
                (g,) = torch.autograd.grad(loss, t)
                tag = f"{name}__{direction}__omit{int(omit)}"
                out[tag + "__loss"] = np.array(loss.item())
                out[tag + "__g"] = g.numpy()
    # numpy branch of the variance: np.var (biased) over the whole cropped stack
    out["image_variance_numpy__minimize__omit1"] = np.array(
        costs.functions["image_variance"](direction="minimize").calculate({"iwe": stack, "omit_boundary": True}))
    out["shims"] = np.array(ref_import.SHIMS)
    out["seed"] = np.array(SEED + 8)
    np.savez_compressed(os.path.join(HERE, "costs_batched.npz"), **out)


def gen_outside_sensor():
    """2-DoF objective on a batch a third of which starts up to 25 px OFF the sensor: the reference's warp has no bounds test on the
    source (warp.py:506-515), the vote masks what lands outside the padded image (event_image_converter.py:340-380)."""
    rng = np.random.default_rng(SEED + 17)
    H, W = 48, 64
    ev = make_events(8000, H, W, rng, fractional=True)
    off = rng.random(ev.shape[0]) < 0.33
    ev[off, 0] = rng.uniform(-25.0, H + 25.0, int(off.sum()))
    ev[off, 1] = rng.uniform(-25.0, W + 25.0, int(off.sum()))
    ev[::7, 0] = np.floor(ev[::7, 0])
    theta = np.array([17.0, -21.0])
    out = dict(events=ev, theta=theta, image_size=np.array([H, W]), shims=np.array(ref_import.SHIMS))
    for pad in (0, 6):
        for cost_name, sigma in (("image_variance", 0), ("gradient_magnitude", 1), ("normalized_image_variance", 1)):
            fs = _fake_solver(H, W, cost_name, sigma, pad=pad)
            loss, gs, iwes = _ref_objective(fs, ev, theta, "2d-translation", None)
            tag = f"pad{pad}__{cost_name}__s{sigma}"
            out[tag + "__loss"] = np.array(loss)
            out[tag + "__grad"] = gs[0].numpy()
            out[tag + "__iwe"] = iwes["iwe"]
    np.savez_compressed(os.path.join(HERE, "outside_sensor.npz"), **out)


def gen_global_best():
    """patch.initialize "global-best" / "grid-best" of the reference's pyramid solver (src/solver/patch_contrast_pyramid.py:292-305):
    initialize_guess_from_whole_image / initialize_guess_from_patch (src/solver/patch_contrast_base.py:164-187, 126-162) RUN on a
    moving-dot scene with the shipped YAML cost -- the reference's pick -- plus the loss of every grid candidate, obtained from the
    same objective_scipy_for_patch the two functions loop over (so a near-tie can be told from a disagreement).
    The batch lasts 0.4 s: the +-150 px/s box of the grid is +-60 px of displacement (a large-motion search)."""
    rng = np.random.default_rng(SEED + 9)
    H, W = 68, 90
    period = 0.4

    def flow_fn(cx, cy):  # pixel per batch period: ~ (24, -12) px = (60, -30) px / s
        return 24.0 + 0.0 * cx, -12.0 + 0.0 * cy

    ev = _moving_dot_events(12000, H, W, flow_fn, rng, n_dots=120, period=period)
    ev[0, 2], ev[-1, 2] = 0.0, period
    slv = _ref_pyramid_solver(H, W, False)
    slv.overload_patch_configuration(1)
    out = {"events": ev, "image_size": np.array([H, W]), "period": np.array(period)}

    def losses(events_np, field):
        slv.events = torch.from_numpy(events_np).double().requires_grad_().to(slv._device)
        grid = np.zeros((len(field), len(field)))
        for i in range(len(field)):
            for j in range(len(field)):
                guess = torch.from_numpy(np.array([field[i], field[j]])).double().requires_grad_().to(slv._device)
                grid[i, j] = float(slv.objective_scipy_for_patch(guess, suppress_log=True))
        return grid

    import logging
    logging.disable(logging.INFO)  # the two functions log every candidate
    field_w = np.arange(-150, 150, 10)
    best_w = slv.initialize_guess_from_whole_image(ev)
    out["whole__field"] = field_w
    out["whole__best"] = best_w.detach().cpu().numpy() if isinstance(best_w, torch.Tensor) else np.asarray(best_w)
    out["whole__loss"] = losses(ev, field_w)
    patch_index = slv.n_patch // 2 - 1
    field_p = np.arange(-150, 150, 30)
    best_p = slv.initialize_guess_from_patch(ev, patch_index=patch_index)
    pt = slv.patches[patch_index]
    cropped = utils.crop_event(ev, pt.x_min, pt.x_max, pt.y_min, pt.y_max)
    out["patch__index"] = np.array(patch_index)
    out["patch__box"] = np.array([pt.x_min, pt.x_max, pt.y_min, pt.y_max])
    out["patch__n_events"] = np.array(len(cropped))
    out["patch__field"] = field_p
    out["patch__best"] = best_p.detach().cpu().numpy() if isinstance(best_p, torch.Tensor) else np.asarray(best_p)
    out["patch__loss"] = losses(cropped, field_p)
    logging.disable(logging.NOTSET)
    out["n_patch"] = np.array(slv.n_patch)
    print("global-best", out["whole__best"], "loss", out["whole__loss"].min(), "grid-best", out["patch__best"], "loss", out["patch__loss"].min())
    save("global_best", **out)


def gen_flow_error():
    """What main.py asks of a solver after optimize() (main.py:100-103, 186-189): calculate_flow_error of the reference's pyramid solver
    (src/solver/patch_contrast_pyramid.py:560-660: EPE / nPE / AE through utils.calculate_flow_error_numpy, GT_FWL / PRED_FWL through
    imager.create_iwe + Warp + NormalizedImageVariance, numpy branch) for a plain and a time-aware (Burgers) solver, with and without
    the event mask -> flow_error.npz"""
    rng = np.random.default_rng(SEED + 12)
    H, W = 68, 90
    period = 0.05
    # shim (4), this fixture only: the numpy branch of the patch interpolation calls cv2.resize(src, None, None, fx, fy, INTER_LINEAR)
    # (patch_contrast_base.py:438-453), absent here like cv2.Sobel -> half-pixel-centre bilinear up-sampling = shim (2)'s arithmetic
    import cv2

    def _cv2_resize(src_img, dsize, dst=None, fx=0, fy=0, interpolation=1):
        assert dsize is None and interpolation == cv2.INTER_LINEAR
        t = torch.from_numpy(np.ascontiguousarray(src_img))[None, None]
        size = [int(round(src_img.shape[0] * fy)), int(round(src_img.shape[1] * fx))]
        return torch.nn.functional.interpolate(t, size=size, mode="bilinear", align_corners=False)[0, 0].numpy()

    cv2.resize = _cv2_resize

    def flow_fn(cx, cy):
        return 6.0 + 4.0 * cx / H - 2.0 * cy / W, -5.0 + 3.0 * cy / W + 2.0 * cx / H

    ev = _moving_dot_events(12000, H, W, flow_fn, rng, n_dots=120, period=period)
    xs, ys = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    gx, gy = flow_fn(xs, ys)
    gt_flow = np.stack([gx, gy], axis=-1)  # [H, W, 2] displacement over the batch (MVSEC convention)
    gt_flow[:5, :7] = 0.0        # invalid ground truth (zeros) is excluded by the metric
    gt_flow_inf = gt_flow.copy()  # ... infinities are meant to be, but `flow_gt * total_mask` turns them into NaN (inf * False): recorded as it is
    gt_flow_inf[-3:, -4:] = np.inf
    out = {"events": ev, "gt_flow": gt_flow, "image_size": np.array([H, W]), "timescale": np.array(period), "seed": np.array(SEED + 12)}
    for tag, time_aware in (("plain", False), ("burgers", True)):
        slv = _ref_pyramid_solver(H, W, time_aware, scale=3)
        s = slv.patch_scales - 1
        slv.overload_patch_configuration(s)
        slv.current_scale = s
        ph, pw = slv.patch_image_size
        # motion in pixel / second; the dense flow the solver builds from it is its negative (interpolate_dense_flow_from_patch_*)
        motion = -np.stack([np.full((ph, pw), 7.0), np.full((ph, pw), -4.0)]) / period * (1.0 + 0.2 * rng.uniform(-1, 1, (2, ph, pw)))
        best = {s: motion}
        with_mask = slv.calculate_flow_error(best, gt_flow, timescale=period, events=ev)
        without = slv.calculate_flow_error(best, gt_flow, timescale=period)
        pred_only = slv.calculate_fwl_pred(best, ev, period)
        if not time_aware:
            with np.errstate(invalid="ignore"):
                inf_case = slv.calculate_flow_error(best, gt_flow_inf, timescale=period)
            out["gt_flow_inf"] = gt_flow_inf
            for k, v in inf_case.items():
                out[f"plain__inf__{k}"] = np.array(float(v))
        out[tag + "__scale"] = np.array(s)
        out[tag + "__motion"] = motion
        out[tag + "__dense"] = np.asarray(slv.motion_to_dense_flow(best, period))  # [2,H,W] or [T,2,H,W]
        for k, v in with_mask.items():
            out[f"{tag}__mask__{k}"] = np.array(float(v))
        for k, v in without.items():
            out[f"{tag}__nomask__{k}"] = np.array(float(v))
        out[tag + "__fwl_pred_only"] = np.array(float(pred_only["PRED_FWL"]))
        print(tag, {k: round(float(v), 5) for k, v in with_mask.items()})
    out["extra_shim"] = np.array("cv2.resize(INTER_LINEAR) -> F.interpolate(bilinear, align_corners=False)")
    save("flow_error", **out)


def gen_core():
    torch.manual_seed(SEED)
    np.random.seed(SEED)
    rng = np.random.default_rng(SEED)
    gen_known_answers()
    gen_warps(rng)
    gen_votes(rng)
    gen_costs(rng)
    gen_flow_voxel(rng)
    gen_objectives(rng)
    gen_hvp(rng)


if __name__ == "__main__":
    # python tests/golden/gen_golden.py [core] [solver] [blur_numpy] [hvp_cases] [solver_hvp] [patch_search] [solver_optimize]
    #                                    [solver_cfg1] [hvp_inv] [costs_batched] [solver_cfg1_variance] [cfg2_fp32] [warp_voxel_optimized] [outside_sensor] [global_best] [flow_error]   (no argument = everything)
    which = [a for a in sys.argv[1:] if not a.startswith("-")] or ["core", "solver", "blur_numpy", "hvp_cases", "solver_hvp",
                                                                    "patch_search", "solver_optimize", "solver_cfg1", "hvp_inv", "costs_batched",
                                                                    "solver_cfg1_variance", "cfg2_fp32", "warp_voxel_optimized", "outside_sensor", "global_best", "flow_error"]
    if "core" in which:
        gen_core()
    if "solver" in which:
        gen_solver_objective(np.random.default_rng(SEED + 1))
    if "blur_numpy" in which:
        gen_blur_numpy()
    if "hvp_cases" in which:
        gen_hvp_cases()
    if "solver_hvp" in which:
        gen_solver_hvp()
    if "patch_search" in which:
        gen_patch_search()
    if "solver_optimize" in which:
        gen_solver_optimize()
    if "solver_cfg1" in which:
        gen_solver_objective_cfg1()
    if "hvp_inv" in which:
        gen_hvp_inv()
    if "costs_batched" in which:
        gen_costs_batched()
    if "solver_cfg1_variance" in which:
        gen_solver_objective_cfg1_variance()
    if "cfg2_fp32" in which:
        gen_cfg2_fp32_reference()
    if "warp_voxel_optimized" in which:
        gen_warp_voxel_optimized()
    if "outside_sensor" in which:
        gen_outside_sensor()
    if "global_best" in which:
        gen_global_best()
    if "flow_error" in which:
        gen_flow_error()
