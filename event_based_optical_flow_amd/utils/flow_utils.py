"""Flow helpers on the boundary.  `construct_dense_flow_voxel_torch` mirrors the reference's
function of the same name (src/utils/flow_utils.py:99-161) over the HIP kernels."""
from typing import Optional

import numpy as np
import torch

from .. import functional as F
from ..array_types import to_device_tensor


def generate_dense_optical_flow(image_size: tuple, max_val: float = 30, seed: Optional[int] = None) -> np.ndarray:
    """[2,H,W] flow ~ U(-max_val, max_val) (distribution of src/utils/flow_utils.py:20-30)."""
    return np.random.default_rng(seed).uniform(-max_val, max_val, (2,) + tuple(image_size))


def generate_smooth_flow(image_size: tuple, max_val: float = 20, grid: int = 16, seed: Optional[int] = None) -> np.ndarray:
    """Bilinear upsampling of a coarse grid x grid U(-max_val, max_val) field: the kind of flow the
    patch solver produces."""
    H, W = image_size
    g = np.random.default_rng(seed).uniform(-max_val, max_val, (2, grid, grid))
    ri = np.linspace(0, grid - 1, H)
    ci = np.linspace(0, grid - 1, W)
    r0 = np.clip(np.floor(ri).astype(int), 0, grid - 2)
    c0 = np.clip(np.floor(ci).astype(int), 0, grid - 2)
    fr = (ri - r0)[None, :, None]
    fc = (ci - c0)[None, None, :]
    g00 = g[:, r0][:, :, c0]
    g10 = g[:, r0 + 1][:, :, c0]
    g01 = g[:, r0][:, :, c0 + 1]
    g11 = g[:, r0 + 1][:, :, c0 + 1]
    return (1 - fr) * (1 - fc) * g00 + fr * (1 - fc) * g10 + (1 - fr) * fc * g01 + fr * fc * g11


def construct_dense_flow_voxel_torch(dense_flow: torch.Tensor, time_bin: int, scheme: str = "upwind",
                                     t0_location: str = "middle", clamp: Optional[int] = None) -> torch.Tensor:
    """[(b,) 2,H,W] flow at t0 -> [(b,) time_bin, 2, H, W] voxel, differentiable.

    Schemes "burgers" and "upwind" (the two the shipped configs can select); the
    nearest/griddata propagators need torch_scatter in the reference and are out of scope."""
    t = to_device_tensor(dense_flow, "dense_flow")
    if t.dim() == 4:
        v = torch.stack([F.construct_dense_flow_voxel(t[i], time_bin, scheme, t0_location) for i in range(t.shape[0])])
    else:
        v = F.construct_dense_flow_voxel(t, time_bin, scheme, t0_location)
    if clamp is not None:
        v = torch.clamp(v, -clamp, clamp)
    return v if v.device == dense_flow.device else v.to(dense_flow.device)
