"""Full-size parity: the BASELINE configurations at THEIR sizes against the fp64 oracle.

The host picks workgroup size, segment size and accumulator layout from the batch; these tests
run exactly the instantiations the bench numbers come from -- t512 2-DoF deferred (cfg2), b512 dense grad-mag on big segments
(cfg3), t512 voxel with blur and small accumulators (cfg4), t512 dense variance on a sparse 720p batch (cfg5 shard), b512 at 20M
events (cfg5) -- plus the cuts no BASELINE configuration reaches (test_work_list_variants_against_the_oracle) and compare IWE,
loss and gradient with oracle/cmax_oracle.c (scalar C, ~6e7 events/s: seconds per case).

Tolerance (BASELINE north_star): 1e-4 relative, fp32 device path against the fp64 oracle -- for the IWE, the loss and
the gradient, with NO slack: |g - g_ref| <= 1e-4 max|g_ref| for every entry of every gradient (round 4).  The handful of
events whose warped coordinate lies within fp32 rounding of a bilinear cell border (the objective's gradient is discontinuous
there: tests/_border.py counts them) are warped again in fp64 by the kernels (warp_exact: 2-DoF since round 3, dense and voxel
since round 4), so they take their derivative from the cell the reference's fp64 arithmetic puts them in.  The oracle is
evaluated on the motion THE DEVICE HOLDS: the flow / voxel rounded to fp32 (the boundary hands the library fp32 flow fields;
the reference itself computes in the dtype of its inputs), a 2-DoF theta in fp64 (cmax_objective_t::motion_dtype)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import event_based_optical_flow_amd as E  # noqa: E402
from oracle import oracle as orc  # noqa: E402

from _border import ambiguity_bound, raw_image_grad  # noqa: E402

TOL = 1e-4


def f32(x):
    """the motion as the device holds it: rounded to fp32 (values), fp64 (container) for the oracle"""
    return np.asarray(x, dtype=np.float32).astype(np.float64)


def rel_max(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def check(tag, h, res, grad, ref, bound=None, n_amb=0):
    """The plain 1e-4 gate (bound = None: every caller since round 4).  n_amb: cell-border events of the batch, for the log."""
    e_iwe = rel_max(h.last_iwe(0).cpu().numpy(), ref["iwes"]["iwe"])
    e_loss = abs(res[0].item() - ref["loss"]) / abs(ref["loss"])
    g = grad.double().cpu().numpy()
    gmax = np.abs(ref["grad"]).max()
    err = np.abs(g - ref["grad"])
    e_grad = err.max() / gmax
    slack = 0.0 if bound is None else 1.01 * bound
    e_gate = ((err - slack).max()) / gmax  # what is left after the border events' own terms
    n_over = int((err > TOL * gmax).sum())
    print(f"[fullsize] {tag}: rel err iwe {e_iwe:.2e} loss {e_loss:.2e} grad {e_grad:.2e} | gated {e_gate:.2e} "
          f"({n_over} of {err.size} gradient entries above 1e-4, {n_amb} cell-border events)")
    assert e_iwe <= TOL and e_loss <= TOL and e_gate <= TOL, (tag, e_iwe, e_loss, e_grad, e_gate)
    assert n_over == 0 or bound is not None, (tag, n_over)


def test_cfg2_full_size_bench_workload(golden):
    """EXACTLY what bench.py times: 1M uniform events (seed 46), 260x346, theta = (12.3, -7.7), variance, sigma 0 -- at the
    PLAIN 1e-4 gate (no slack for cell-border events: round 3, warp_one decides their cells in fp64), against the oracle and
    against the reference's own fp64 row for this stream (tests/golden/cfg2_fp32_reference.npz; its fp32 row is at 1.3e-3)."""
    g = golden("cfg2_fp32_reference")
    size, n = (260, 346), 1_000_000
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=46)
    np.testing.assert_allclose([ev[:, 0].sum(), ev[:, 1].sum(), ev[:, 2].sum()], g["events_checksum"], rtol=1e-14)
    theta = np.array([12.3, -7.7])
    h = E.CMaxHandle(size).set_events(ev)
    desc = E.make_descriptor("image_variance", "2d-translation")
    ref = orc.objective(ev, theta, "2d-translation", size, cost="image_variance", sigma=0)
    for rep in range(2):  # fp64 theta as the solver holds it (cmax_objective_t::motion_dtype = CMAX_F64)
        res, grad = h.evaluate(desc, theta)
        check(f"cfg2 1M 260x346 2-DoF variance, fp64 theta #{rep}", h, res, grad, ref)
    e_ref = rel_max(grad.cpu().numpy(), g["f64__grad"])
    e_ref32 = rel_max(g["f32__grad"], g["f64__grad"])
    print(f"[fullsize] cfg2 gradient against the reference's fp64 row: {e_ref:.2e} (the reference's own fp32 row: {e_ref32:.2e}, "
          f"{int(g['events_in_another_cell_in_fp32'])} events in another cell)")
    assert e_ref <= TOL and abs(res[0].item() - float(g["f64__loss"])) <= TOL * abs(float(g["f64__loss"]))
    # the way bench.py calls it: theta already fp32 on the device (prepared call).  Same gate against the oracle on THAT theta.
    theta32 = torch.tensor(theta, dtype=torch.float32, device="cuda")
    call, res32, grad32 = h.prepare(desc, theta32)
    assert call.motion_is_callers
    call()
    torch.cuda.synchronize()
    ref32 = orc.objective(ev, theta32.double().cpu().numpy(), "2d-translation", size, cost="image_variance", sigma=0)
    check("cfg2 1M, fp32 theta (bench.py's prepared call)", h, res32, grad32, ref32)


def test_cfg3_full_size_dense_gradmag():
    """5M events, 480x640, dense smooth flow, gradient magnitude: t512 K1 / K3 <dense, grad-mag (kFoldScale)>."""
    size, n = (480, 640), 5_000_000
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=46)
    flow = E.utils.generate_smooth_flow(size, 20, seed=1046)
    h = E.CMaxHandle(size).set_events(ev)
    res, grad = h.evaluate(E.make_descriptor("gradient_magnitude", "dense-flow"), flow)
    ref = orc.objective(ev, f32(flow), "dense-flow", size, cost="gradient_magnitude", sigma=0)
    _, n_amb = ambiguity_bound(ev, f32(flow), "dense-flow", size, raw_image_grad(ref, 0))
    check("cfg3 5M 480x640 dense grad-mag", h, res, grad, ref, None, n_amb)


def test_cfg4_full_size_burgers_voxel():
    """2M events, 260x346, Burgers voxel T = 10 (t0 middle), variance with sigma 1: t512 K1, t1024 voxel K3."""
    size, n, Tn = (260, 346), 2_000_000, 10
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=46)
    f0 = E.utils.generate_smooth_flow(size, 20, seed=1046)
    voxel = orc.construct_dense_flow_voxel(f0 / 20.0, Tn, "burgers", "middle") * 20.0
    h = E.CMaxHandle(size).set_events(ev, time_bin=Tn)
    res, grad = h.evaluate(E.make_descriptor("image_variance", "dense-flow-voxel", sigma=1.0, time_bin=Tn), voxel)
    ref = orc.objective(ev, f32(voxel), "dense-flow-voxel", size, cost="image_variance", sigma=1)
    _, n_amb = ambiguity_bound(ev, f32(voxel), "dense-flow-voxel", size, raw_image_grad(ref, 1))
    check("cfg4 2M 260x346 voxel T=10 sigma 1", h, res, grad, ref, None, n_amb)


def test_cfg4_size_voxel_plain_variance():
    """cfg4's batch with the un-blurred variance: the 1024-thread owned voxel K3 with the statistics inside its launch
    (kFoldStatsInside), which no BASELINE configuration reaches at this workgroup size."""
    size, n, Tn = (260, 346), 2_000_000, 10
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=48)
    f0 = E.utils.generate_smooth_flow(size, 20, seed=1048)
    voxel = orc.construct_dense_flow_voxel(f0 / 20.0, Tn, "burgers", "middle") * 20.0
    h = E.CMaxHandle(size).set_events(ev, time_bin=Tn)
    assert h.batch_info()["owned_groups"]
    desc = E.make_descriptor("image_variance", "dense-flow-voxel", sigma=0.0, time_bin=Tn)
    ref = orc.objective(ev, f32(voxel), "dense-flow-voxel", size, cost="image_variance", sigma=0)
    _, n_amb = ambiguity_bound(ev, f32(voxel), "dense-flow-voxel", size, raw_image_grad(ref, 0))
    for rep in range(2):
        res, grad = h.evaluate(desc, voxel)
        check(f"cfg4-size 2M voxel T=10 plain variance #{rep}", h, res, grad, ref, None, n_amb)


def test_cfg5_shard_full_size_with_gradient():
    """One rank's share of cfg5: 2.5M events on 720x1280 (2.7 events per pixel), dense flow, variance + gradient."""
    size, n = (720, 1280), 2_500_000
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=46)
    flow = E.utils.generate_smooth_flow(size, 20, seed=1046)
    h = E.CMaxHandle(size).set_events(ev)
    res, grad = h.evaluate(E.make_descriptor("image_variance", "dense-flow"), flow)
    ref = orc.objective(ev, f32(flow), "dense-flow", size, cost="image_variance", sigma=0)
    _, n_amb = ambiguity_bound(ev, f32(flow), "dense-flow", size, raw_image_grad(ref, 0))
    check("cfg5 shard 2.5M 720x1280 dense variance", h, res, grad, ref, None, n_amb)


WORK_LISTS = [
    # (tag, size, n, T, model, expected segment_events, expected small accumulators, segments above 1024?)
    ("voxel 1.7M 260x346: standard segments of 4 groups, 512-thread K3 with 12-group accumulators", (260, 346), 1_700_000, 10, "dense-flow-voxel", 2040, False, False),
    ("voxel 3M 480x640: 12 sparse groups per MID segment (1022 workgroups: one round), 512-thread K3 with 12-group accumulators", (480, 640), 3_000_000, 10, "dense-flow-voxel", 3064, False, False),
    ("voxel 3.4M 480x640: 8 groups per standard segment, more than 1024 segments -> 1024-thread K3", (480, 640), 3_400_000, 10, "dense-flow-voxel", 2040, False, True),
    ("voxel 4M 260x346: one group per standard segment -> BIG segments of 3 groups, b1024 K3", (260, 346), 4_000_000, 10, "dense-flow-voxel", 4088, False, True),
    ("dense 4M 720p: one tile per standard segment -> BIG segments of 3 tiles, owned b512 K3", (720, 1280), 4_000_000, 0, "dense-flow", 4088, False, True),
    ("dense 3M 720p: two tiles per standard segment stay", (720, 1280), 3_000_000, 0, "dense-flow", 2040, False, True),
]


@pytest.mark.parametrize("tag,size,n,Tn,model,seg_events,small,wide", WORK_LISTS, ids=[w[0].split(":")[0] for w in WORK_LISTS])
def test_work_list_variants_against_the_oracle(tag, size, n, Tn, model, seg_events, small, wide):
    """Round 3: the host picks the segment size (2040 / 4088 events) and the voxel K3's accumulator array from the group sizes of
    the work list (profiles/r03_ablation.txt 14 / 17).  Each case asserts the cut it was written for
    (cmax_work_list_info) and holds loss, IWE and gradient of a variance objective to the oracle -- these are instantiations of
    the event kernels that no BASELINE configuration reaches: the owned b512 / b1024 K3, the 1024-thread voxel K3 with the large
    accumulators, the 512-thread one below 1024 segments."""
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=51)
    f0 = E.utils.generate_smooth_flow(size, 12, seed=1051)
    motion = orc.construct_dense_flow_voxel(f0 / 12.0, Tn, "burgers", "middle") * 12.0 if Tn else f0
    h = E.CMaxHandle(size).set_events(ev, time_bin=Tn)
    info = h.work_list_info()
    assert info["segment_events"] == seg_events and info["small_accumulators"] == small, (tag, info)
    assert (info["segments"] > 1024) == wide, (tag, info)
    assert h.batch_info()["owned_groups"]
    desc = E.make_descriptor("image_variance", model, sigma=0.0, time_bin=Tn)
    ref = orc.objective(ev, f32(motion), model, size, cost="image_variance", sigma=0)
    _, n_amb = ambiguity_bound(ev, f32(motion), model, size, raw_image_grad(ref, 0))
    res, grad = h.evaluate(desc, motion)
    check(tag, h, res, grad, ref, None, n_amb)


def test_work_list_rule_on_the_bench_configurations():
    """cfg2 (2674 events per tile: not group-aligned, 1M events) keeps standard segments, cfg3 (4170 per tile, 5M events) gets big
    ones, cfg4 (535 events per (tile, bin) group) the small accumulators on MID segments (round 4: 748 workgroups of five groups in one
    round instead of 1258 of three in two), cfg5's shard (694 per tile) mid segments of four tiles (900 instead of 1693)."""
    cases = [((260, 346), 1_000_000, 0, 2040, False), ((480, 640), 5_000_000, 0, 4088, False), ((260, 346), 2_000_000, 10, 3064, True),
             ((720, 1280), 2_500_000, 0, 3064, False)]
    for size, n, Tn, seg_events, small in cases:
        ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=46)
        info = E.CMaxHandle(size).set_events(ev, time_bin=Tn).work_list_info()
        assert info["segment_events"] == seg_events and info["small_accumulators"] == small, (size, n, Tn, info)


def test_cfg5_two_time_slices_of_5m_against_the_oracle():
    """cfg5's exchange pattern at a size the oracle does in seconds: 5M events on 720x1280 as two 2.5M-event time
    slices with the batch-wide extremes, images and gradients summed like C1 / C2, against the oracle on all 5M."""
    size, n = (720, 1280), 5_000_000
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=47)
    flow = E.utils.generate_smooth_flow(size, 20, seed=1047)
    desc = E.make_descriptor("image_variance", "dense-flow")
    tmin, tmax = ev[:, 2].min(), ev[:, 2].max()
    ranks = [E.CMaxHandle(size).set_events(ev[: n // 2], tmin, tmax), E.CMaxHandle(size).set_events(ev[n // 2:], tmin, tmax)]
    images = sum(h.objective_vote(desc, flow) for h in ranks)
    outs = [h.objective_finish(desc, flow, images) for h in ranks]
    ref = orc.objective(ev, f32(flow), "dense-flow", size, cost="image_variance", sigma=0)
    gsum = sum(g.double() for _, g in outs).cpu().numpy()
    e_iwe = rel_max(images[0].cpu().numpy(), ref["iwes"]["iwe"])
    e_loss = abs(outs[0][0][0].item() - ref["loss"]) / abs(ref["loss"])
    _, n_amb = ambiguity_bound(ev, f32(flow), "dense-flow", size, raw_image_grad(ref, 0))
    gmax = np.abs(ref["grad"]).max()
    e_grad = rel_max(gsum, ref["grad"])
    print(f"[fullsize] cfg5 2 x 2.5M time slices: rel err iwe {e_iwe:.2e} loss {e_loss:.2e} grad {e_grad:.2e} ({n_amb} cell-border events)")
    assert e_iwe <= TOL and e_loss <= TOL and e_grad <= TOL


def test_cfg5_full_20m_eight_slices():
    """cfg5 AS BASELINE STATES IT (configs[4]): 20M events on 720x1280, dense flow, variance, time-sliced x 8 -- eight
    handles on this GPU standing for the eight ranks, each with its contiguous 2.5M-event time slice and the batch-wide
    extremes; images summed the way C1 does, every rank finishing on the summed image, gradients summed the way C2
    does -- against the oracle on all 20M events.  Also the N = 1 point of the strong-scaling curve: ONE handle holding
    the whole batch (bench.py `also.cfg5_strong`), same gate."""
    size, n, world = (720, 1280), 20_000_000, 8
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=46)
    flow = E.utils.generate_smooth_flow(size, 20, seed=1046)
    desc = E.make_descriptor("image_variance", "dense-flow")
    ref = orc.objective(ev, f32(flow), "dense-flow", size, cost="image_variance", sigma=0)
    _, n_amb = ambiguity_bound(ev, f32(flow), "dense-flow", size, raw_image_grad(ref, 0))
    gmax = np.abs(ref["grad"]).max()
    tmin, tmax = ev[:, 2].min(), ev[:, 2].max()
    from event_based_optical_flow_amd.distributed import time_slice_bounds

    ranks = []
    for r in range(world):
        lo, hi = time_slice_bounds(n, world, r)
        ranks.append(E.CMaxHandle(size).set_events(torch.from_numpy(ev[lo:hi]).cuda(), tmin, tmax))
    assert sum(h.n_events for h in ranks) == n
    images = sum(h.objective_vote(desc, flow) for h in ranks)  # C1
    outs = [h.objective_finish(desc, flow, images) for h in ranks]
    gsum = sum(g.double() for _, g in outs).cpu().numpy()  # C2
    e_iwe = rel_max(images[0].cpu().numpy(), ref["iwes"]["iwe"])
    e_loss = max(abs(o[0][0].item() - ref["loss"]) / abs(ref["loss"]) for o in outs)
    err = np.abs(gsum - ref["grad"])
    n_over = int((err > TOL * gmax).sum())
    print(f"[fullsize] cfg5 20M = 8 x 2.5M time slices: rel err iwe {e_iwe:.2e} loss {e_loss:.2e} grad {err.max() / gmax:.2e} "
          f"({n_over} of {err.size} gradient entries above 1e-4, {n_amb} cell-border events)")
    assert e_iwe <= TOL and e_loss <= TOL and err.max() <= TOL * gmax
    for h in ranks:
        h.close()
    del ranks, images, outs
    # N = 1: the whole batch behind one handle
    h = E.CMaxHandle(size).set_events(torch.from_numpy(ev).cuda())
    assert h.n_events == n
    res, grad = h.evaluate(desc, flow)
    check("cfg5 20M 720x1280 dense variance, one handle", h, res, grad, ref, None, n_amb)
    h.close()


@pytest.mark.parametrize("model,sigma,omit", [("dense-flow", 0, True), ("dense-flow", 0, False), ("dense-flow-voxel", 0, True),
                                               ("dense-flow", 1, True), ("dense-flow", 1, False), ("dense-flow-voxel", 1, True)])
def test_mean_from_votes_along_the_border(model, sigma, omit):
    """The variance paths that take the image mean from K1's vote sums instead of from finished statistics (round 2:
    k_blur_stats_adj_var for sigma > 0, the statistics inside the K3 launch for sigma = 0 on owned groups): a strong
    flow on a small sensor, so that a large share of the votes lands in the two-pixel band along the border or leaves
    the image -- exactly the events whose share of the mean is not simply 'one per event'."""
    size, n, Tn = (288, 352), 700_000, 4  # 6.9 events per pixel: every 16 x 16 tile stays below one segment (owned groups)
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=77)
    f0 = E.utils.generate_smooth_flow(size, 40, seed=1077)
    if model == "dense-flow":
        motion, tb = f0, 0
    else:
        motion, tb = np.stack([f0 * (1.0 + 0.1 * k) for k in range(Tn)]), Tn
    h = E.CMaxHandle(size).set_events(ev, time_bin=tb) if tb else E.CMaxHandle(size).set_events(ev)
    assert h.batch_info()["owned_groups"]
    desc = E.make_descriptor("image_variance", model, sigma=float(sigma), omit_boundary=omit, time_bin=tb)
    ref = orc.objective(ev, f32(motion), model, size, cost="image_variance", sigma=sigma, omit_boundary=omit)
    _, n_amb = ambiguity_bound(ev, f32(motion), model, size, raw_image_grad(ref, sigma))
    for rep in range(3):  # the vote sums are double-buffered across evaluations: every one must see clean accumulators
        res, grad = h.evaluate(desc, motion)
        check(f"border {model} sigma {sigma} omit {int(omit)} #{rep}", h, res, grad, ref, None, n_amb)


# ---------------------------------------------------------------------------------------------------------------------------------
# Round 5 (VERDICT r4 #1b): every row the driver's bench line TIMES has an oracle test on the SAME workload -- inputs built by
# bench.make_inputs itself (seed, generator, motion), the handle prepared the way bench.run_workload prepares it.
# ---------------------------------------------------------------------------------------------------------------------------------
def _bench_inputs(name):
    import bench

    cfg = bench.WORKLOADS[name]
    ev, motion, T = bench.make_inputs(cfg, 0, 1)
    return cfg, ev, motion, T


@pytest.mark.parametrize("name", ["cfg3_rough", "cfg5_rough"])
def test_bench_rough_rows_against_the_oracle(name):
    """also.cfg3_rough / also.cfg5_rough: the per-pixel random flow F ~ U(-5, 5) of src/utils/flow_utils.py:20-30 -- neighbouring source
    pixels warp to unrelated places (no coalesced gathers, ragged LDS windows) -- at the bench's sizes (5M @480x640 grad-mag, 2.5M
    @720x1280 variance), plain 1e-4 gate on IWE, loss and every gradient entry."""
    cfg, ev, flow, _ = _bench_inputs(name)
    size = (cfg["H"], cfg["W"])
    assert ev.shape[0] == cfg["n"] and flow.shape == (2,) + size and np.abs(flow).max() <= 5.0
    h = E.CMaxHandle(size).set_events(ev)
    desc = E.make_descriptor(cfg["cost"], cfg["model"], sigma=cfg["sigma"])
    ref = orc.objective(ev, f32(flow), cfg["model"], size, cost=cfg["cost"], sigma=0)
    _, n_amb = ambiguity_bound(ev, f32(flow), cfg["model"], size, raw_image_grad(ref, 0))
    for rep in range(2):
        res, grad = h.evaluate(desc, flow)
        check(f"{name} #{rep}", h, res, grad, ref, None, n_amb)
    h.close()


@pytest.mark.parametrize("name", ["cfg2_theta150", "cfg2_theta80"])
def test_bench_large_motion_rows_in_time_slabs(name):
    """also.cfg2_theta150 / cfg2_theta80: cfg2's own 1M-event stream on 260x346 at theta = (150, -100) / (80, -50) px over the batch, events
    in 4 time slabs (cmax_set_time_slabs) -- half of the events leave the sensor at 150 px.  Plain gate, fp64 theta and the bench's
    prepared fp32 call; the un-slabbed order of the same handle must agree with the slab order to accumulation noise."""
    cfg, ev, theta, _ = _bench_inputs(name)
    size = (cfg["H"], cfg["W"])
    h = E.CMaxHandle(size).set_events(ev)
    h.set_time_slabs(cfg["slabs"])
    desc = E.make_descriptor(cfg["cost"], cfg["model"], sigma=cfg["sigma"])
    ref = orc.objective(ev, theta, cfg["model"], size, cost=cfg["cost"], sigma=0)
    res, grad = h.evaluate(desc, theta)
    check(f"{name} fp64 theta, {cfg['slabs']} slabs", h, res, grad, ref)
    th32 = torch.tensor(theta, dtype=torch.float32, device="cuda")
    call, res32, grad32 = h.prepare(desc, th32)
    call()
    torch.cuda.synchronize()
    ref32 = orc.objective(ev, th32.double().cpu().numpy(), cfg["model"], size, cost=cfg["cost"], sigma=0)
    check(f"{name} fp32 theta (bench.py's prepared call)", h, res32, grad32, ref32)
    h.set_time_slabs(0)  # back to the un-binned order: same events, sums associated differently
    res_u, grad_u = h.evaluate(desc, theta)
    check(f"{name} un-slabbed", h, res_u, grad_u, ref)
    assert rel_max(grad_u.cpu().numpy(), grad.cpu().numpy()) <= 1e-5
    h.close()


def test_bench_batch8_row_on_cfg2_stream():
    """also.cfg2_batch8: cmax_objective_batch with K = 8 candidate thetas (a line search's steps 0.6 .. 1.3 x theta, as bench.py draws
    them) on cfg2's 1M-event stream: every candidate's loss and gradient against the oracle at the plain gate, and equal to what
    eight single calls give."""
    cfg, ev, theta, _ = _bench_inputs("cfg2_batch8")
    size, K = (cfg["H"], cfg["W"]), int(cfg["batch"])
    thetas = np.asarray(theta, dtype=np.float64)[None, :] * np.linspace(0.6, 1.3, K)[:, None]
    h = E.CMaxHandle(size).set_events(ev)
    desc = E.make_descriptor(cfg["cost"], cfg["model"], sigma=cfg["sigma"])
    dev_thetas = torch.from_numpy(thetas).cuda().float().contiguous()  # fp32 on the device: what bench.py hands over
    call, results, grads = h.prepare_batch(desc, dev_thetas)
    for rep in range(3):  # the batch's vote images are double-buffered: every call must find clean ones
        call()
    torch.cuda.synchronize()
    results, grads = results.cpu().numpy(), grads.cpu().numpy()
    worst = [0.0, 0.0]
    for z in range(K):
        th = dev_thetas[z].double().cpu().numpy()
        ref = orc.objective(ev, th, cfg["model"], size, cost=cfg["cost"], sigma=0)
        e_loss = abs(results[z, 0] - ref["loss"]) / abs(ref["loss"])
        e_grad = rel_max(grads[z], ref["grad"])
        worst = [max(worst[0], e_loss), max(worst[1], e_grad)]
        assert e_loss <= TOL and e_grad <= TOL, (z, th, e_loss, e_grad)
        r1, g1 = h.evaluate(desc, dev_thetas[z])
        assert abs(r1[0].item() - results[z, 0]) <= 1e-6 * abs(results[z, 0]) and rel_max(g1.cpu().numpy(), grads[z]) <= 1e-5
    print(f"[fullsize] cfg2_batch8: worst rel err over 8 candidates loss {worst[0]:.2e} grad {worst[1]:.2e}")
    h.close()


@pytest.mark.parametrize("model", ["2d-translation", "dense-flow"])
def test_fractional_sources_and_a_non_dyadic_reference_time_at_the_plain_gate(model):
    """Round 5.  The leaf API accepts what the reference's pipeline never produces: source coordinates with fractional
    parts (rectified events before the cast of src/utils/event_utils.py:110-115) and a reference time anywhere in the batch
    (Warp.calculate_reftime, src/warp.py:216-218).  1M such events, reference time at 1/3 of the batch: every gradient entry at 1e-4 --
    the events on a cell border are decided from the fp64 source residual (rx + rx_lo) and the fp64 reference time (d + d_lo)."""
    size, n = (260, 346), 1_000_000
    rng = np.random.default_rng(21)
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=21)
    ev[:, 0] = np.minimum(ev[:, 0] + rng.uniform(0, 1, n), size[0] - 1e-3)
    ev[:, 1] = np.minimum(ev[:, 1] + rng.uniform(0, 1, n), size[1] - 1e-3)
    direction = 1.0 / 3.0
    motion = np.array([12.3, -7.7]) if model == "2d-translation" else f32(E.utils.generate_smooth_flow(size, 20, seed=1021))
    h = E.CMaxHandle(size).set_events(ev)
    assert h.batch_info()["fractional"]
    desc = E.make_descriptor("image_variance", model, warp_direction=direction)
    ref = orc.objective(ev, motion, model, size, cost="image_variance", sigma=0, warp_direction=direction)
    _, n_amb = ambiguity_bound(ev, motion, model, size, raw_image_grad(ref, 0), direction=direction)
    for rep in range(2):
        res, grad = h.evaluate(desc, motion)
        check(f"{model} 1M fractional sources, reference time 1/3 #{rep}", h, res, grad, ref, None, n_amb)
    h.close()
