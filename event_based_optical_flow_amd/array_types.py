"""Array-kind helpers of the boundary (reference: src/types/__init__.py:12-45)."""
from typing import Any, Union

import numpy as np
import torch

NUMPY_TORCH = Union[np.ndarray, torch.Tensor]
FLOAT_TORCH = Union[float, torch.Tensor]


def is_torch(arr: Any) -> bool:
    return isinstance(arr, torch.Tensor)


def is_numpy(arr: Any) -> bool:
    return isinstance(arr, np.ndarray)


def nt_max(array: NUMPY_TORCH, dim: int) -> NUMPY_TORCH:
    return array.max(axis=dim) if is_numpy(array) else torch.max(array, dim).values


def nt_min(array: NUMPY_TORCH, dim: int) -> NUMPY_TORCH:
    return array.min(axis=dim) if is_numpy(array) else torch.min(array, dim).values


def to_device_tensor(arr: NUMPY_TORCH, what: str = "array") -> torch.Tensor:
    """numpy / CPU tensor -> CUDA tensor (float32 stays float32, everything else float64).

    The kernels only run on the GPU; this is the single place host data crosses PCIe."""
    from . import _lib

    _lib.require_gpu()
    if is_numpy(arr):
        t = torch.from_numpy(np.ascontiguousarray(arr))
    elif is_torch(arr):
        t = arr
    else:
        raise RuntimeError(f"Non-supported type of {what}. {type(arr)}")
    if t.dtype not in (torch.float32, torch.float64):
        t = t.to(torch.float64)
    return t if t.is_cuda else t.to("cuda")


def like_input(result: torch.Tensor, ref: NUMPY_TORCH) -> NUMPY_TORCH:
    """Return `result` as the same kind (numpy / tensor on the same device) as `ref`."""
    if is_numpy(ref):
        return result.detach().cpu().numpy()
    return result if result.device == ref.device else result.to(ref.device)
