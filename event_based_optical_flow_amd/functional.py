"""Differentiable operators over the C ABI (libcmax_hip.so).

Each function takes CUDA (ROCm) torch tensors, hands their device pointers to one C entry point on
torch's current stream and wraps forward/backward in a torch.autograd.Function so that
`torch.autograd.grad(loss, x)` -- what the reference's optimiser adapter calls
(src/solver/scipy_autograd/torch_wrapper.py:40) -- flows through the HIP kernels.
PyTorch is used for device memory, streams and the autograd tape only.
"""
from typing import Optional, Tuple, Union

import torch

from . import _lib
from ._lib import check

MODEL_CODES = {
    "2d-translation": _lib.MODEL_2DOF,
    "rigid-optical-flow": _lib.MODEL_2DOF,
    "dense-flow": _lib.MODEL_DENSE,
    "dense-flow-voxel": _lib.MODEL_VOXEL,
}
SCHEME_CODES = {"burgers": _lib.SCHEME_BURGERS, "upwind": _lib.SCHEME_UPWIND}
_DIRECTION_FRAC = {"middle": 0.5, "before": -1.0, "after": 2.0}


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """Raw hipStream_t of torch's current stream on the current device (the stream every library call is enqueued on)."""
    if _raw_stream is not None:  # one C call instead of building a torch.cuda.Stream object (~3 us per library call)
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return _lib.F32
    if t.dtype == torch.float64:
        return _lib.F64
    raise TypeError(f"libcmax_hip supports float32/float64 tensors, got {t.dtype}")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _cuda(t: torch.Tensor, what: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{what} must live on the GPU (got {t.device}); there is no CPU path")
    return t.contiguous()


def direction_to_ref(direction: Union[str, float]) -> Tuple[int, float]:
    """Warp.calculate_reftime's direction argument (src/warp.py:201-233) -> (ref_mode, frac)."""
    if type(direction) is float:
        return _lib.REF_FRAC, float(direction)
    if direction == "first":
        return _lib.REF_FIRST, 0.0
    if direction == "last":
        return _lib.REF_LAST, 1.0
    if direction in _DIRECTION_FRAC:
        return _lib.REF_FRAC, _DIRECTION_FRAC[direction]
    if direction == "random":
        import numpy as np

        return _lib.REF_FRAC, float(np.random.uniform(low=0.0, high=1.0))
    raise ValueError(f"direction argument should be first, middle, last. Or float. {direction}")


# ------------------------------------------------------------------------------------------------
def tminmax(events: torch.Tensor) -> torch.Tensor:
    """(t_min, t_max) of an [n,4] event tensor as a device double[2] (no host sync)."""
    _lib.require_gpu()
    events = _cuda(events, "events")
    out = torch.empty(2, dtype=torch.float64, device=events.device)
    check(_lib.load().cmax_tminmax(_ptr(events), _code(events), events.shape[0], _ptr(out), _stream()))
    return out


class _WarpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, events, motion, model, image_size, tmm, ref_mode, frac, normalize_t):
        lib = _lib.load()
        n = events.shape[0]
        H, W = int(image_size[0]), int(image_size[1])
        T = int(motion.shape[0]) if model == _lib.MODEL_VOXEL else 0
        warped = torch.empty_like(events)
        dt = torch.empty(n, dtype=events.dtype, device=events.device)
        bins = torch.empty(n, dtype=torch.int32, device=events.device) if model == _lib.MODEL_VOXEL else None
        check(lib.cmax_warp_events(_ptr(events), _code(events), n, model, _ptr(motion), T, H, W, _ptr(tmm), ref_mode,
                                   frac, int(normalize_t), _ptr(warped), _ptr(dt), _ptr(bins), _stream()))
        ctx.save_for_backward(events, dt, bins if bins is not None else torch.empty(0))
        ctx.meta = (model, T, H, W, motion.shape)
        return warped

    @staticmethod
    def backward(ctx, gwarped):
        events, dt, bins = ctx.saved_tensors
        model, T, H, W, mshape = ctx.meta
        gwarped = gwarped.contiguous()
        gmotion = torch.empty(mshape, dtype=events.dtype, device=events.device)
        check(_lib.load().cmax_warp_events_bwd(_ptr(events), _code(events), events.shape[0], model, T, H, W, _ptr(dt),
                                               _ptr(bins) if model == _lib.MODEL_VOXEL else None, _ptr(gwarped),
                                               _ptr(gmotion), _stream()))
        return None, gmotion, None, None, None, None, None, None


def warp_events(events, motion, motion_model: str, image_size, direction="first", normalize_t=False):
    """Warp.warp_event for one un-batched [n,4] tensor (src/warp.py:156-199)."""
    _lib.require_gpu()
    if motion_model not in MODEL_CODES:
        raise KeyError(motion_model)
    events = _cuda(events, "events")
    motion = _cuda(motion, "motion").to(events.dtype)
    ref_mode, frac = direction_to_ref(direction)
    tmm = tminmax(events)
    return _WarpFn.apply(events, motion, MODEL_CODES[motion_model], tuple(image_size), tmm, ref_mode, frac, bool(normalize_t))


# ------------------------------------------------------------------------------------------------
class _VoteFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, events, weight, wscalar, Hp, Wp, ph, pw, eps, count):
        lib = _lib.load()
        n, stride = events.shape
        img = torch.empty((Hp, Wp), dtype=events.dtype, device=events.device)
        check(lib.cmax_vote(_ptr(events), _code(events), stride, n, _ptr(weight), wscalar, Hp, Wp, ph, pw, eps,
                            int(count), _ptr(img), _stream()))
        ctx.save_for_backward(events, weight if weight is not None else torch.empty(0))
        ctx.meta = (weight is not None, wscalar, Hp, Wp, ph, pw, eps, count)
        return img

    @staticmethod
    def backward(ctx, G):
        events, weight = ctx.saved_tensors
        has_w, wscalar, Hp, Wp, ph, pw, eps, count = ctx.meta
        n, stride = events.shape
        gev = torch.zeros_like(events)
        gw = None
        if not count and n > 0:
            G = G.contiguous()
            gxy = torch.empty((n, 2), dtype=events.dtype, device=events.device)
            need_gw = has_w and ctx.needs_input_grad[1]
            gw = torch.empty(n, dtype=events.dtype, device=events.device) if need_gw else None
            check(_lib.load().cmax_vote_bwd(_ptr(events), _code(events), stride, n, _ptr(weight) if has_w else None,
                                            wscalar, Hp, Wp, ph, pw, eps, _ptr(G), _ptr(gxy), _ptr(gw), _stream()))
            gev[:, :2] = gxy
        return gev, gw, None, None, None, None, None, None, None


def vote(events, image_size, outer_padding=(0, 0), weight=1.0, eps=1e-6, count=False):
    """bilinear_vote_tensor / count_event_tensor for one [n,>=2] tensor
    (src/event_image_converter.py:316-374, 209-255).  image_size is the PADDED size."""
    _lib.require_gpu()
    events = _cuda(events, "events")
    Hp, Wp = int(image_size[0]), int(image_size[1])
    ph, pw = int(outer_padding[0]), int(outer_padding[1])
    wt, wscalar = None, 1.0
    if isinstance(weight, torch.Tensor):
        assert weight.shape == events.shape[:-1]
        wt = _cuda(weight, "weight").to(events.dtype)
    else:
        wscalar = float(weight)
    return _VoteFn.apply(events, wt, wscalar, Hp, Wp, ph, pw, float(eps), bool(count))


# ------------------------------------------------------------------------------------------------
class _BlurFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, sigma):
        out = torch.empty_like(img)
        check(_lib.load().cmax_blur3(_ptr(img), _code(img), img.shape[0], img.shape[1], sigma, 0, _ptr(out), _stream()))
        ctx.sigma = sigma
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        out = torch.empty_like(g)
        check(_lib.load().cmax_blur3(_ptr(g), _code(g), g.shape[0], g.shape[1], ctx.sigma, 1, _ptr(out), _stream()))
        return out, None


def gaussian_blur3(img, sigma):
    """3-tap reflect-101 Gaussian blur of one [H,W] image (src/event_image_converter.py:153-159)."""
    _lib.require_gpu()
    return _BlurFn.apply(_cuda(img, "image"), float(sigma))


# ------------------------------------------------------------------------------------------------
class _ContrastFn(torch.autograd.Function):
    """RAW contrast (variance or gradient magnitude) of one [H,W] image."""

    @staticmethod
    def forward(ctx, img, cost, omit, ddof):
        val = torch.empty(4, dtype=torch.float64, device=img.device)  # [0] result, [1..3] scratch
        check(_lib.load().cmax_contrast(_ptr(img), _code(img), img.shape[0], img.shape[1], cost, int(omit), ddof,
                                        _ptr(val), None, None, _stream()))
        ctx.save_for_backward(img)
        ctx.meta = (cost, omit, ddof)
        return val[0].to(img.dtype)

    @staticmethod
    def backward(ctx, gout):
        (img,) = ctx.saved_tensors
        cost, omit, ddof = ctx.meta
        gs = gout.detach().to(torch.float64).reshape(1).contiguous()
        val = torch.empty(4, dtype=torch.float64, device=img.device)
        G = torch.empty_like(img)
        check(_lib.load().cmax_contrast(_ptr(img), _code(img), img.shape[0], img.shape[1], cost, int(omit), ddof,
                                        _ptr(val), _ptr(G), _ptr(gs), _stream()))
        return G, None, None, None


def contrast(img, cost: int, omit_boundary: bool, ddof: int = 1):
    _lib.require_gpu()
    img = _cuda(img, "image")
    if omit_boundary and (img.shape[0] <= 2 or img.shape[1] <= 2):
        raise ValueError("omit_boundary needs an image larger than 2x2")
    return _ContrastFn.apply(img, int(cost), bool(omit_boundary), int(ddof))


class _TVFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flow, omit):
        val = torch.empty(4, dtype=torch.float64, device=flow.device)
        check(_lib.load().cmax_total_variation(_ptr(flow), _code(flow), flow.shape[1], flow.shape[2], int(omit),
                                               _ptr(val), None, None, _stream()))
        ctx.save_for_backward(flow)
        ctx.omit = omit
        return val[0].to(flow.dtype)

    @staticmethod
    def backward(ctx, gout):
        (flow,) = ctx.saved_tensors
        gs = gout.detach().to(torch.float64).reshape(1).contiguous()
        val = torch.empty(4, dtype=torch.float64, device=flow.device)
        G = torch.empty_like(flow)
        check(_lib.load().cmax_total_variation(_ptr(flow), _code(flow), flow.shape[1], flow.shape[2], int(ctx.omit),
                                               _ptr(val), _ptr(G), _ptr(gs), _stream()))
        return G, None


def total_variation(flow, omit_boundary: bool):
    """mean |Sobel_4ch(flow)/8| of one [2,h,w] flow (src/costs/total_variation.py:60-75,110-126)."""
    _lib.require_gpu()
    return _TVFn.apply(_cuda(flow, "flow"), bool(omit_boundary))


# ------------------------------------------------------------------------------------------------
class _FlowStepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flow, dt, scheme):
        out = torch.empty_like(flow)
        check(_lib.load().cmax_flow_step(_ptr(flow), _code(flow), flow.shape[1], flow.shape[2], dt, scheme, _ptr(out), _stream()))
        ctx.save_for_backward(flow)
        ctx.meta = (dt, scheme)
        return out

    @staticmethod
    def backward(ctx, gout):
        (flow,) = ctx.saved_tensors
        dt, scheme = ctx.meta
        gout = gout.contiguous()
        gF = torch.zeros_like(flow)
        check(_lib.load().cmax_flow_step_adj(_ptr(flow), _code(flow), flow.shape[1], flow.shape[2], dt, scheme,
                                             _ptr(gout), _ptr(gF), _stream()))
        return gF, None, None


def set_leaf_deterministic(enable: bool = True) -> bool:
    """Process-wide (cmax_set_leaf_deterministic): the adjoints of `flow_step` / `construct_dense_flow_voxel` (and the second-order
    adjoint) accumulate without atomics, in a fixed order -- bit-identical gradients from run to run.  Returns the previous setting."""
    return bool(_lib.load().cmax_set_leaf_deterministic(int(bool(enable))))


def flow_step(flow, dt: float, scheme: str):
    """One Burgers / upwind step on a [2,H,W] flow (src/utils/flow_utils.py:567-639 / 439-493)."""
    _lib.require_gpu()
    return _FlowStepFn.apply(_cuda(flow, "flow"), float(dt), SCHEME_CODES[scheme])


class _VoxelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flow, T, t0, scheme):
        _, H, W = flow.shape
        V = torch.empty((T, 2, H, W), dtype=flow.dtype, device=flow.device)
        check(_lib.load().cmax_voxel_construct(_ptr(flow), _code(flow), T, t0, H, W, scheme, _ptr(V), _stream()))
        ctx.save_for_backward(V)
        ctx.meta = (T, t0, scheme)
        return V

    @staticmethod
    def backward(ctx, gV):
        (V,) = ctx.saved_tensors
        T, t0, scheme = ctx.meta
        _, _, H, W = V.shape
        gV = gV.contiguous().clone()  # clobbered by the adjoint sweep
        gF = torch.empty((2, H, W), dtype=V.dtype, device=V.device)
        check(_lib.load().cmax_voxel_construct_adj(_ptr(V), _code(V), T, t0, H, W, scheme, _ptr(gV), _ptr(gF), _stream()))
        return gF, None, None, None


def construct_dense_flow_voxel(flow, time_bin: int, scheme: str = "upwind", t0_location: str = "middle"):
    """construct_dense_flow_voxel_torch for one [2,H,W] flow (src/utils/flow_utils.py:99-161)."""
    _lib.require_gpu()
    if t0_location == "first":
        t0 = 0
    elif t0_location == "middle":
        t0 = time_bin // 2
    else:
        raise NotImplementedError(f"{t0_location =} not supported")
    if scheme not in SCHEME_CODES:
        raise NotImplementedError(f"scheme {scheme!r}: only 'burgers' and 'upwind' are built (see DESIGN.md)")
    return _VoxelFn.apply(_cuda(flow, "flow"), int(time_bin), t0, SCHEME_CODES[scheme])


def voxel_construct_tan(flow, dflow, time_bin: int, scheme: str = "burgers", t0_location: str = "middle"):
    """(V, dV): the voxel of `flow` and its directional derivative along `dflow` (cmax_voxel_construct_tan).
    Not differentiable itself -- a building block of exact Hessian-vector products."""
    _lib.require_gpu()
    flow, dflow = _cuda(flow.detach(), "flow").contiguous(), _cuda(dflow.detach(), "dflow").to(flow.dtype).contiguous()
    t0 = 0 if t0_location == "first" else time_bin // 2
    _, H, W = flow.shape
    V = torch.empty((time_bin, 2, H, W), dtype=flow.dtype, device=flow.device)
    dV = torch.empty_like(V)
    check(_lib.load().cmax_voxel_construct_tan(_ptr(flow), _ptr(dflow), _code(flow), time_bin, t0, H, W, SCHEME_CODES[scheme],
                                               _ptr(V), _ptr(dV), _stream()))
    return V, dV


def voxel_construct_adj_tan(V, dV, gV, dgV, scheme: str = "burgers", t0_location: str = "middle"):
    """(gF, dgF) = (J^T gV, J^T dgV + (dJ[dV])^T gV) of the voxel chain at V with tangent dV (cmax_voxel_construct_adj_tan)."""
    _lib.require_gpu()
    V = _cuda(V.detach(), "V").contiguous()
    T, _, H, W = V.shape
    dV = _cuda(dV.detach(), "dV").to(V.dtype).contiguous()
    gV = _cuda(gV.detach(), "gV").to(V.dtype).contiguous().clone()  # clobbered
    dgV = _cuda(dgV.detach(), "dgV").to(V.dtype).contiguous().clone()
    t0 = 0 if t0_location == "first" else T // 2
    gF = torch.empty((2, H, W), dtype=V.dtype, device=V.device)
    dgF = torch.empty_like(gF)
    check(_lib.load().cmax_voxel_construct_adj_tan(_ptr(V), _ptr(dV), _code(V), T, t0, H, W, SCHEME_CODES[scheme], _ptr(gV), _ptr(dgV),
                                                   _ptr(gF), _ptr(dgF), _stream()))
    return gF, dgF


# ------------------------------------------------------------------------------------------------
class _PatchToDenseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, motion, image_size, sliding_window, pad):
        _, ph, pw = motion.shape
        H, W = image_size
        flow = torch.empty((2, H, W), dtype=motion.dtype, device=motion.device)
        check(_lib.load().cmax_patch_to_dense(_ptr(motion), _code(motion), ph, pw, pad[0], pad[1], sliding_window[0],
                                              sliding_window[1], H, W, 0, _ptr(flow), _stream()))
        ctx.meta = (ph, pw, H, W, sliding_window, pad)
        return flow

    @staticmethod
    def backward(ctx, gflow):
        ph, pw, H, W, sliding_window, pad = ctx.meta
        gflow = gflow.contiguous()
        gm = torch.empty((2, ph, pw), dtype=gflow.dtype, device=gflow.device)
        check(_lib.load().cmax_patch_to_dense(_ptr(gflow), _code(gflow), ph, pw, pad[0], pad[1], sliding_window[0],
                                              sliding_window[1], H, W, 1, _ptr(gm), _stream()))
        return gm, None, None, None


def patch_to_dense(motion, image_size, sliding_window, pad):
    """interpolate_dense_flow_from_patch_tensor for a [2,ph,pw] patch motion (bilinear filter)
    (src/solver/patch_contrast_base.py:462-506)."""
    _lib.require_gpu()
    motion = _cuda(motion, "motion")
    return _PatchToDenseFn.apply(motion, (int(image_size[0]), int(image_size[1])),
                                 (int(sliding_window[0]), int(sliding_window[1])), (int(pad[0]), int(pad[1])))


def gaussian_filter(img, sigma):
    """scipy.ndimage.gaussian_filter of one [H,W] image (numpy branch of the reference's create_iwe,
    src/event_image_converter.py:122-124).  Not differentiable (the numpy branch never is)."""
    _lib.require_gpu()
    img = _cuda(img.detach(), "image")
    tmp, out = torch.empty_like(img), torch.empty_like(img)
    check(_lib.load().cmax_gaussian_filter(_ptr(img), _code(img), img.shape[0], img.shape[1], float(sigma), _ptr(tmp),
                                           _ptr(out), _stream()))
    return out
