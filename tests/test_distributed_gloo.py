"""world_size-2 gloo test (CPU) of the time-sliced multi-GPU path.

The per-rank compute of the product is HIP-only, so here the rank-local phases are served by a
stand-in built on the CPU oracle (checker role): the test exercises what is new in the N > 1
path -- the time-slice partition, the MIN/MAX agreement on (t_min, t_max), the all-reduce of the
vote images (C1) and of the gradient (C2) in event_based_optical_flow_amd/distributed.py -- and
checks that both ranks end up with the single-process loss and gradient."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import event_based_optical_flow_amd as E
from event_based_optical_flow_amd import _lib
from event_based_optical_flow_amd.distributed import TimeSlicedObjective, time_slice_bounds
from oracle import oracle as orc


class OracleLocal:
    """Rank-local phases with the CMaxHandle interface, computed by the fp64 oracle."""

    def __init__(self, image_size, loss_jitter=0.0):
        self.size = image_size
        self.loss_jitter = loss_jitter  # relative perturbation of THIS rank's loss: the stand-in for fp64 atomics in another order

    def set_events(self, events, tmin, tmax, time_bin=0):
        self.ev = np.ascontiguousarray(events, dtype=np.float64)
        self.tmin, self.tmax = tmin, tmax

    def _dt(self, k, desc):
        frac = {_lib.REF_FIRST: 0.0, _lib.REF_LAST: 1.0}.get(desc.ref_mode[k], desc.ref_frac[k])
        return (self.ev[:, 2] - self.tmin) / (self.tmax - self.tmin) - frac  # normalize_t, global extremes

    def _warp(self, k, desc, motion):
        dt = np.ascontiguousarray(self._dt(k, desc))
        out = np.empty_like(self.ev)
        n = self.ev.shape[0]
        L = orc.lib()
        m = np.ascontiguousarray(motion, dtype=np.float64)
        if desc.model == _lib.MODEL_2DOF:
            L.orc_warp_2dof(orc._p(self.ev), ctypes.c_longlong(n), orc._p(m), orc._p(dt), orc._p(out))
        else:
            L.orc_warp_dense(orc._p(self.ev), ctypes.c_longlong(n), orc._p(m), self.size[0], self.size[1], orc._p(dt), orc._p(out))
        return out, dt

    def objective_vote(self, desc, motion):
        motion = motion.numpy()
        imgs = [orc.vote(self._warp(k, desc, motion)[0], self.size) for k in range(desc.n_ref)]
        if desc.normalized:
            imgs.append(orc.vote(self.ev, self.size))
        return torch.from_numpy(np.stack(imgs))

    def objective_finish(self, desc, motion, images, want_grad=True):
        motion = motion.numpy()
        imgs = images.numpy()
        kind = "var" if desc.cost == _lib.COST_VARIANCE else "gm"
        omit = bool(desc.omit_boundary)
        v_orig = orc._base_cost(kind, imgs[desc.n_ref], omit if kind == "gm" else False)[0] if desc.normalized else 0.0
        loss, grad = 0.0, np.zeros_like(motion)
        for k in range(desc.n_ref):
            v, G = orc._base_cost(kind, imgs[k], omit)
            if desc.normalized:
                loss += desc.mult[k] * v_orig / v
                G = desc.mult[k] * (-v_orig / v ** 2) * G
            else:
                loss += desc.mult[k] * (-v)
                G = -desc.mult[k] * G
            warped, dt = self._warp(k, desc, motion)
            gx, gy = orc.vote_bwd(warped, self.size, G)
            model = "2d-translation" if desc.model == _lib.MODEL_2DOF else "dense-flow"
            grad = grad + orc.motion_grad(self.ev, motion, model, {"dt": dt}, gx, gy)
        res = torch.zeros(8, dtype=torch.float64)
        res[0] = loss * (1.0 + self.loss_jitter)
        return res, torch.from_numpy(np.ascontiguousarray(grad))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        size = (24, 32)
        ev = E.utils.generate_events(3001, size[0], size[1], 0.0, 0.05, seed=7)  # odd count: uneven slices
        lo, hi = time_slice_bounds(len(ev), world, rank)
        obj = TimeSlicedObjective(OracleLocal(size))
        tmin, tmax = obj.set_local_events(ev[lo:hi])
        assert (tmin, tmax) == (ev[:, 2].min(), ev[:, 2].max())
        cost, model, motion = case
        desc = E.make_descriptor(cost, model, sigma=0.0)
        res, grad = obj.evaluate(desc, torch.from_numpy(motion))
        out_q.put((rank, float(res[0]), grad.numpy()))
    finally:
        dist.destroy_process_group()


CASES = [
    ("image_variance", "2d-translation", np.array([9.0, -6.0])),
    ("multi_focal_normalized_image_variance", "2d-translation", np.array([9.0, -6.0])),
    ("gradient_magnitude", "dense-flow", E.utils.generate_smooth_flow((24, 32), 8, grid=4, seed=3)),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] + "-" + c[1] for c in CASES])
def test_time_sliced_objective_world2(case):
    ctx = mp.get_context("spawn")
    for attempt in range(3):  # the free port can be taken again before rank 0's store binds it: then both ranks die at once
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, case, q)) for r in range(2)]
        for p in procs:
            p.start()
        got = []
        try:
            got = [q.get(timeout=120) for _ in range(2)]
        except Exception:  # queue.Empty: a rank did not get as far as its result
            pass
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
        if len(got) == 2 and all(p.exitcode == 0 for p in procs):
            break
        assert attempt < 2, [p.exitcode for p in procs]
    cost, model, motion = case
    size = (24, 32)
    ev = E.utils.generate_events(3001, size[0], size[1], 0.0, 0.05, seed=7)
    ref = orc.objective(ev, motion, model, size, cost=cost, sigma=0)
    for rank, loss, grad in got:
        np.testing.assert_allclose(loss, ref["loss"], rtol=1e-11)
        np.testing.assert_allclose(grad, ref["grad"], rtol=1e-8, atol=1e-13 * max(1.0, np.abs(ref["grad"]).max()))


class _SlicedLoss(torch.autograd.Function):
    """loss(theta) of a TimeSlicedObjective as an autograd node: what TorchWrapper differentiates under scipy.optimize.minimize"""

    @staticmethod
    def forward(ctx, theta, obj, desc):
        res, grad = obj.evaluate(desc, theta.detach())
        ctx.grad = grad
        return res[0].clone()

    @staticmethod
    def backward(ctx, gout):
        return ctx.grad * gout, None, None


def _minimize_worker(rank, world, port, consistent, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from event_based_optical_flow_amd.solver.scipy_autograd import minimize

        size = (24, 32)
        ev = E.utils.generate_structured_events(3001, size[0], size[1], (6.0, -4.0), n_dots=40, seed=8)
        lo, hi = time_slice_bounds(len(ev), world, rank)
        # every rank's loss carries ITS OWN last-bit noise, like statistics summed with atomics in another order on every GPU
        obj = TimeSlicedObjective(OracleLocal(size, loss_jitter=(3e-16, -5e-16)[rank]))
        if not consistent:  # negative control: the evaluation as it was before round 6 (no rank-0 scalars)
            obj._rank0_value = lambda t: t
        obj.set_local_events(ev[lo:hi])
        desc = E.make_descriptor("image_variance", "2d-translation", sigma=0.0)
        losses = []

        def fun(theta):
            loss = _SlicedLoss.apply(theta, obj, desc)
            losses.append(float(loss))
            return loss

        res = minimize(fun, np.array([4.0, -2.5]), method="BFGS", precision="float64", options={"maxiter": 12})
        out_q.put((rank, np.asarray(res.x).copy(), float(res.fun), int(res.nfev), int(res.njev), losses))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("consistent", [True, False])
def test_replicated_scipy_minimize_sees_identical_scalars(consistent):
    """every rank runs the SAME SciPy loop (src/solver/scipy_autograd/scipy_minimize.py:100-117) on a loss that each
    rank sums in its own order.  With the rank-0 scalars of round 6 both ranks must see bit-identical losses at every call, hence
    take the same decisions: identical nfev / njev / x / fun.  The negative control (the exchange switched off) shows the stand-in's
    jitter is visible to the optimiser at all -- there the recorded losses differ between the ranks."""
    ctx = mp.get_context("spawn")
    for attempt in range(3):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_minimize_worker, args=(r, 2, port, consistent, q)) for r in range(2)]
        for p in procs:
            p.start()
        got = []
        try:
            got = [q.get(timeout=240) for _ in range(2)]
        except Exception:
            pass
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
        if len(got) == 2 and all(p.exitcode == 0 for p in procs):
            break
        assert attempt < 2, [p.exitcode for p in procs]
    got.sort(key=lambda g: g[0])
    (_, x0, f0, nfev0, njev0, l0), (_, x1, f1, nfev1, njev1, l1) = got
    if consistent:
        assert nfev0 == nfev1 and njev0 == njev1 and nfev0 >= 3
        assert l0 == l1  # every loss the optimiser saw, bit for bit
        assert f0 == f1 and np.array_equal(x0, x1)
    else:
        n = min(len(l0), len(l1))
        assert any(a != b for a, b in zip(l0[:n], l1[:n]))


def test_time_slice_bounds_cover_everything():
    for n in (0, 1, 7, 1000, 1001):
        for w in (1, 2, 3, 8):
            b = [time_slice_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
