"""Shared plumbing of the image-contrast costs: numpy / torch in, HIP kernel, same kind out."""
import logging

import numpy as np
import torch

from .. import _lib
from .. import functional as F
from ..array_types import to_device_tensor

logger = logging.getLogger(__name__)


def raw_contrast(img, cost_code: int, omit_boundary: bool):
    """RAW (unsigned) contrast of one [H,W] image, or of a stack [..., H, W].

    numpy -> python float with the numpy branch's statistics (np.var is biased, image_variance.py:68);
    tensor -> 0-dim tensor on the input's device with torch.var's Bessel correction (line 55),
    differentiable through cmax_contrast.
    A stack is ONE sample for the reference: the variance is taken over every element of the cropped stack
    (image_variance.py:38-40, 55) and the gradient magnitude is the mean over images and pixels
    (gradient_magnitude.py:62-75)."""
    if isinstance(img, torch.Tensor):
        ddof, is_np = 1, False
    elif isinstance(img, np.ndarray):
        ddof, is_np = 0, True
    else:
        e = f"Unsupported input type. {type(img)}."
        logger.error(e)
        raise NotImplementedError(e)
    t = to_device_tensor(img, "iwe")
    if t.dim() < 2:
        raise ValueError(f"contrast costs take images [..., H, W], got shape {tuple(t.shape)}")
    if t.dim() == 2:
        v = F.contrast(t, cost_code, omit_boundary, ddof)
    elif cost_code == VARIANCE:
        # every element of the (cropped) stack is one sample: the same kernel on the stack laid out as one tall image
        x = t[..., 1:-1, 1:-1] if omit_boundary else t
        v = F.contrast(x.reshape(-1, x.shape[-1]).contiguous(), cost_code, False, ddof)
    elif is_np:
        # cv2.Sobel of the reference's numpy branch reads a third axis as channels: there is no batched meaning to mirror
        raise NotImplementedError("gradient magnitude of a numpy stack: the reference's numpy branch is not batch-aware")
    else:
        imgs = t.reshape(-1, t.shape[-2], t.shape[-1])  # equal sizes: the mean over everything = mean of the images' means
        v = torch.stack([F.contrast(im, cost_code, omit_boundary, ddof) for im in imgs]).mean()
    if is_np:
        return float(v.item())
    return v if v.device == img.device else v.to(img.device)


VARIANCE = _lib.COST_VARIANCE
GRADMAG = _lib.COST_GRADMAG
