"""Compact events (round 5): batches of >= 8M events are cut into big segments and the hot kernels read the 4.5-byte-per-event copy
of the sorted events (csrc/cmax_event_kernels.inc, "COMPACT EVENTS": 24-bit fixed-point time | pixel in tile, a tile nibble per slot,
one aligned region per segment).  The BASELINE rows that run this path (cfg3, cfg5 at 20M events, the rough rows) are held to the
oracle in test_gpu_fullsize.py; here the corners they do not reach, on an 8.2M-event batch of a small sensor (segments that are cut
inside source tiles, every segment full): 2-DoF with an fp64 theta and the event at tau = 1 (stored as 1 - 2^-24), fractional source
coordinates with a non-dyadic reference time, three reference times in one launch, K candidate motions per call, and the
deterministic mode -- each against oracle/cmax_oracle.c at the plain 1e-4 gate, no slack (every gradient entry)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import event_based_optical_flow_amd as E  # noqa: E402
from oracle import oracle as orc  # noqa: E402

TOL = 1e-4
SIZE, N = (128, 160), 8_200_000


def f32(x):
    return np.asarray(x, dtype=np.float32).astype(np.float64)


def rel_max(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def gate(tag, h, res, grad, ref):
    e_iwe = rel_max(h.last_iwe(0).cpu().numpy(), ref["iwes"]["iwe"])
    e_loss = abs(res[0].item() - ref["loss"]) / abs(ref["loss"])
    e_grad = rel_max(grad.double().cpu().numpy(), ref["grad"])
    print(f"[compact] {tag}: rel err iwe {e_iwe:.2e} loss {e_loss:.2e} grad {e_grad:.2e}")
    assert e_iwe <= TOL and e_loss <= TOL and e_grad <= TOL, (tag, e_iwe, e_loss, e_grad)


@pytest.fixture(scope="module")
def batch():
    ev = E.utils.generate_events(N, SIZE[0], SIZE[1], 0.0, 0.05, seed=77)
    h = E.CMaxHandle(SIZE).set_events(ev)
    info = h.work_list_info()
    assert info["segment_events"] == 4088, info  # big segments: the compact copy is what K1 / K3 read
    yield ev, h
    h.close()


def test_two_dof_fp64_theta_middle_reference(batch):
    ev, h = batch
    theta = np.array([7.3, -4.1])
    desc = E.make_descriptor("image_variance", "2d-translation", warp_direction="middle")
    ref = orc.objective(ev, theta, "2d-translation", SIZE, cost="image_variance", sigma=0, warp_direction="middle")
    for rep in range(2):
        res, grad = h.evaluate(desc, theta)
        gate(f"2-DoF 8.2M events, fp64 theta, reference time middle #{rep}", h, res, grad, ref)


def test_dense_three_reference_times(batch):
    ev, h = batch
    flow = f32(E.utils.generate_smooth_flow(SIZE, 8, seed=1077))
    desc = E.make_descriptor("multi_focal_normalized_image_variance", "dense-flow")
    ref = orc.objective(ev, flow, "dense-flow", SIZE, cost="multi_focal_normalized_image_variance", sigma=0)
    res, grad = h.evaluate(desc, flow)
    e_loss = abs(res[0].item() - ref["loss"]) / abs(ref["loss"])
    e_grad = rel_max(grad.double().cpu().numpy(), ref["grad"])
    print(f"[compact] dense 8.2M events, three reference times: rel err loss {e_loss:.2e} grad {e_grad:.2e}")
    assert e_loss <= TOL and e_grad <= TOL


def test_dense_deterministic_mode(batch):
    ev, h = batch
    flow = f32(E.utils.generate_smooth_flow(SIZE, 8, seed=1078))
    desc = E.make_descriptor("image_variance", "dense-flow")
    ref = orc.objective(ev, flow, "dense-flow", SIZE, cost="image_variance", sigma=0)
    h.set_deterministic(True)
    try:
        res0, grad0 = h.evaluate(desc, flow)
        gate("dense 8.2M events, deterministic mode", h, res0, grad0, ref)
        r0, g0 = res0.clone(), grad0.clone()
        res1, grad1 = h.evaluate(desc, flow)
        assert torch.equal(r0, res1) and torch.equal(g0, grad1)  # bit-identical from run to run
    finally:
        h.set_deterministic(False)
    res, grad = h.evaluate(desc, flow)
    gate("dense 8.2M events, default mode", h, res, grad, ref)


def test_candidate_batch_on_big_segments(batch):
    ev, h = batch
    K = 4
    thetas = np.array([7.3, -4.1])[None, :] * np.linspace(0.7, 1.2, K)[:, None]
    desc = E.make_descriptor("image_variance", "2d-translation")
    dev = torch.from_numpy(thetas).cuda().float().contiguous()
    call, results, grads = h.prepare_batch(desc, dev)
    for rep in range(2):
        call()
    torch.cuda.synchronize()
    results, grads = results.cpu().numpy(), grads.cpu().numpy()
    for z in range(K):
        ref = orc.objective(ev, dev[z].double().cpu().numpy(), "2d-translation", SIZE, cost="image_variance", sigma=0)
        e_loss, e_grad = abs(results[z, 0] - ref["loss"]) / abs(ref["loss"]), rel_max(grads[z], ref["grad"])
        print(f"[compact] candidate {z} of {K}: rel err loss {e_loss:.2e} grad {e_grad:.2e}")
        assert e_loss <= TOL and e_grad <= TOL, (z, e_loss, e_grad)


@pytest.mark.parametrize("model", ["2d-translation", "dense-flow"])
def test_fractional_sources_on_big_segments(model):
    rng = np.random.default_rng(78)
    ev = E.utils.generate_events(N, SIZE[0], SIZE[1], 0.0, 0.05, seed=78)
    ev[:, 0] = np.minimum(ev[:, 0] + rng.uniform(0, 1, N), SIZE[0] - 1e-3)
    ev[:, 1] = np.minimum(ev[:, 1] + rng.uniform(0, 1, N), SIZE[1] - 1e-3)
    direction = 1.0 / 3.0
    motion = np.array([7.3, -4.1]) if model == "2d-translation" else f32(E.utils.generate_smooth_flow(SIZE, 8, seed=1079))
    h = E.CMaxHandle(SIZE).set_events(ev)
    assert h.batch_info()["fractional"] and h.work_list_info()["segment_events"] == 4088
    desc = E.make_descriptor("image_variance", model, warp_direction=direction)
    ref = orc.objective(ev, motion, model, SIZE, cost="image_variance", sigma=0, warp_direction=direction)
    res, grad = h.evaluate(desc, motion)
    gate(f"{model} 8.2M events, fractional sources, reference time 1/3", h, res, grad, ref)
    h.close()


def test_compact_copy_follows_the_work_list():
    """The compact copy belongs to ONE work list: re-binning a big batch (binned work lists read the 8-byte events), returning to the
    un-binned order, a smaller batch on the same handle (standard segments: no compact copy) and a big batch again must each evaluate
    what the oracle does -- nothing may read a region packed for another cut."""
    ev = E.utils.generate_events(N, SIZE[0], SIZE[1], 0.0, 0.05, seed=79)
    flow = f32(E.utils.generate_smooth_flow(SIZE, 8, seed=1080))
    desc = E.make_descriptor("image_variance", "dense-flow")
    ref = orc.objective(ev, flow, "dense-flow", SIZE, cost="image_variance", sigma=0)
    h = E.CMaxHandle(SIZE).set_events(ev)
    res, grad = h.evaluate(desc, flow)
    gate("big batch", h, res, grad, ref)
    h.set_time_bins(4)  # (tile, bin) groups: the voxel layout of the same events
    voxel = np.stack([flow] * 4)
    vdesc = E.make_descriptor("image_variance", "dense-flow-voxel", time_bin=4)
    vref = orc.objective(ev, voxel, "dense-flow-voxel", SIZE, cost="image_variance", sigma=0)
    res, grad = h.evaluate(vdesc, voxel)
    gate("the same batch in 4 time bins (voxel objective)", h, res, grad, vref)
    h.set_time_bins(0)
    res, grad = h.evaluate(desc, flow)
    gate("back in the un-binned order", h, res, grad, ref)
    small = E.utils.generate_events(600_000, SIZE[0], SIZE[1], 0.0, 0.05, seed=80)
    h.set_events(small)
    assert h.work_list_info()["segment_events"] < 4088
    sref = orc.objective(small, flow, "dense-flow", SIZE, cost="image_variance", sigma=0)
    res, grad = h.evaluate(desc, flow)
    gate("a 600k-event batch on the same handle", h, res, grad, sref)
    h.set_events(ev)
    res, grad = h.evaluate(desc, flow)
    gate("the big batch again", h, res, grad, ref)
    h.close()


def test_short_runs_on_big_segments_take_the_plain_events():
    """a dense objective that is neither `owned` (two reference times) nor long-run (4 events per pixel) runs the
    kGradStrided K3, whose slot layout reads the 8-byte events; K1 of the same evaluation must then warp with the same fp32 time
    (K3 follows K1's cells and windows without tests).  4.2M events on a 1024 x 1024 sensor with one hot tile (> 2040 events: the
    standard cut is not group-aligned, so the work list is cut into big segments) -- such a handle gets no compact copy."""
    size, n = (1024, 1024), 4_200_000
    rng = np.random.default_rng(80)
    ev = E.utils.generate_events(n, size[0], size[1], 0.0, 0.05, seed=80)
    hot = rng.choice(n, 6000, replace=False)  # one hot source tile
    ev[hot, 0] = rng.integers(512, 528, hot.size)
    ev[hot, 1] = rng.integers(256, 272, hot.size)
    flow = f32(E.utils.generate_smooth_flow(size, 9, seed=1081))
    h = E.CMaxHandle(size).set_events(ev)
    assert h.work_list_info()["segment_events"] == 4088, h.work_list_info()
    for cost in ("multi_focal_normalized_image_variance", "image_variance"):  # three reference times (not owned) | one (owned)
        desc = E.make_descriptor(cost, "dense-flow")
        ref = orc.objective(ev, flow, "dense-flow", size, cost=cost, sigma=0)
        for rep in range(2):
            res, grad = h.evaluate(desc, flow)
            e_loss = abs(res[0].item() - ref["loss"]) / abs(ref["loss"])
            e_grad = rel_max(grad.double().cpu().numpy(), ref["grad"])
            print(f"[compact] short runs on big segments, {cost} #{rep}: rel err loss {e_loss:.2e} grad {e_grad:.2e}")
            assert e_loss <= TOL and e_grad <= TOL, (cost, e_loss, e_grad)
    h.close()
