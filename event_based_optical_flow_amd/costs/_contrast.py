"""Shared plumbing of the image-contrast costs: numpy / torch in, HIP kernel, same kind out."""
import logging

import numpy as np
import torch

from .. import _lib
from .. import functional as F
from ..array_types import to_device_tensor

logger = logging.getLogger(__name__)


def raw_contrast(img, cost_code: int, omit_boundary: bool):
    """RAW (unsigned) contrast of one [H,W] image.

    numpy -> python float with the numpy branch's statistics (np.var is biased, image_variance.py:68);
    tensor -> 0-dim tensor on the input's device with torch.var's Bessel correction (line 55),
    differentiable through cmax_contrast."""
    if isinstance(img, torch.Tensor):
        ddof, is_np = 1, False
    elif isinstance(img, np.ndarray):
        ddof, is_np = 0, True
    else:
        e = f"Unsupported input type. {type(img)}."
        logger.error(e)
        raise NotImplementedError(e)
    t = to_device_tensor(img, "iwe")
    if t.dim() != 2:
        raise NotImplementedError("contrast costs take one [H, W] image (batched IWEs are not built)")
    v = F.contrast(t, cost_code, omit_boundary, ddof)
    if is_np:
        return float(v.item())
    return v if v.device == img.device else v.to(img.device)


VARIANCE = _lib.COST_VARIANCE
GRADMAG = _lib.COST_GRADMAG
