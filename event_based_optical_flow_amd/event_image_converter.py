"""`EventImageConverter`: events -> image of (warped) events, the reference's class
(src/event_image_converter.py) over the HIP kernels cmax_vote / cmax_vote_bwd / cmax_blur3.

numpy inputs use the numpy branch's floor epsilon (1e-8, line 282) and blur
(scipy.ndimage.gaussian_filter restated as cmax_gaussian_filter, 122-124); torch inputs the torch
branch's epsilon (1e-6, line 340) and 3-tap reflect-101 Gaussian (153-159).
"""
import logging
from typing import Optional, Tuple, Union

import numpy as np
import torch

from . import functional as F
from .array_types import FLOAT_TORCH, NUMPY_TORCH, is_numpy, is_torch, like_input, to_device_tensor

logger = logging.getLogger(__name__)


class EventImageConverter(object):
    """Args:
        image_size (tuple) ... (H, W)
        outer_padding (int or tuple) ... the image grows by 2*padding per axis; event coordinates
            are shifted by the padding (src/event_image_converter.py:23-28, 344-345).
    """

    def __init__(self, image_size: tuple, outer_padding: Union[int, Tuple[int, int]] = 0):
        if isinstance(outer_padding, (int, float)):
            self.outer_padding = (int(outer_padding), int(outer_padding))
        else:
            self.outer_padding = outer_padding
        self.image_size = tuple(int(i + p * 2) for i, p in zip(image_size, self.outer_padding))

    def update_property(self, image_size: Optional[tuple] = None,
                        outer_padding: Optional[Union[int, Tuple[int, int]]] = None):
        # NB the reference adds the padding ONCE here, unlike the constructor (line 42 vs 28)
        if image_size is not None:
            self.image_size = image_size
        if outer_padding is not None:
            self.outer_padding = (outer_padding, outer_padding) if isinstance(outer_padding, int) else outer_padding
        self.image_size = tuple(i + p for i, p in zip(self.image_size, self.outer_padding))

    # -- higher layer ----------------------------------------------------------------------------
    def create_iwe(self, events: NUMPY_TORCH, method: str = "bilinear_vote", sigma: int = 1) -> NUMPY_TORCH:
        """[(b,) n, 4] events -> [(b,) H, W] image of warped events (src/event_image_converter.py:45-67)."""
        if is_numpy(events):
            return self.create_image_from_events_numpy(events, method, sigma=sigma)
        elif is_torch(events):
            return self.create_image_from_events_tensor(events, method, sigma=sigma)
        e = f"Non-supported type of events. {type(events)}"
        logger.error(e)
        raise RuntimeError(e)

    def create_eventmask(self, events: NUMPY_TORCH) -> NUMPY_TORCH:
        """[(b,) 1, H, W] boolean mask of pixels that received any vote (69-82)."""
        if is_numpy(events):
            return (0 != self.create_image_from_events_numpy(events, sigma=0))[..., None, :, :]
        elif is_torch(events):
            return (0 != self.create_image_from_events_tensor(events, sigma=0))[..., None, :, :]
        raise RuntimeError

    # -- lower layer -----------------------------------------------------------------------------
    def create_image_from_events_numpy(self, events: np.ndarray, method: str = "bilinear_vote",
                                       weight: Union[float, np.ndarray] = 1.0, sigma: int = 1) -> np.ndarray:
        if method == "count":
            image = self.count_event_numpy(events)
        elif method == "bilinear_vote":
            image = self.bilinear_vote_numpy(events, weight=weight)
        elif method == "polarity":
            pos = events[..., 3] > 0
            if is_numpy(weight):
                imgs = [self.bilinear_vote_numpy(events[m], weight=weight[m]) for m in (pos, ~pos)]
            else:
                imgs = [self.bilinear_vote_numpy(events[m], weight=weight) for m in (pos, ~pos)]
            image = np.stack(imgs, axis=-3)
        else:
            e = f"{method = } is not supported."
            logger.error(e)
            raise NotImplementedError(e)
        if sigma > 0:
            # scipy.ndimage.gaussian_filter(image, sigma) on the GPU.  The reference filters EVERY axis, so a
            # batch or the 2-channel "polarity" stack would also be blurred across images (line 123); that
            # quirk is not reproduced.
            if image.ndim != 2:
                e = "numpy-branch blur is built for one [H, W] image (the reference also blurs across the batch axis)"
                logger.error(e)
                raise NotImplementedError(e)
            image = F.gaussian_filter(to_device_tensor(image, "image"), sigma).cpu().numpy()
        return image

    def create_image_from_events_tensor(self, events: torch.Tensor, method: str = "bilinear_vote",
                                        weight: FLOAT_TORCH = 1.0, sigma: int = 0) -> torch.Tensor:
        if method == "count":
            image = self.count_event_tensor(events)
        elif method == "bilinear_vote":
            image = self.bilinear_vote_tensor(events, weight=weight)
        else:
            e = f"{method = } is not implemented"
            logger.error(e)
            raise NotImplementedError(e)
        if sigma > 0:
            dev = image.device
            img = to_device_tensor(image, "image")
            if img.dim() == 2:
                img = F.gaussian_blur3(img, sigma)
            else:
                img = torch.stack([F.gaussian_blur3(img[i], sigma) for i in range(img.shape[0])])
            image = img.to(dev)
        return torch.squeeze(image)

    # -- vote kernels ------------------------------------------------------------------------------
    def _vote(self, events: NUMPY_TORCH, weight, eps: float, count: bool) -> NUMPY_TORCH:
        ev = to_device_tensor(events, "events")
        wt = weight
        if is_numpy(weight) or is_torch(weight):
            assert tuple(weight.shape) == tuple(events.shape[:-1])
            wt = to_device_tensor(weight, "weight").to(ev.dtype)
        batched = ev.dim() == 3
        if not batched:
            ev = ev[None]
            if is_torch(wt):
                wt = wt[None]
        imgs = [F.vote(ev[i], self.image_size, self.outer_padding, wt[i] if is_torch(wt) else wt, eps, count)
                for i in range(ev.shape[0])]
        out = torch.stack(imgs).squeeze()  # the reference squeezes every singleton axis (line 374)
        return like_input(out, events)

    def bilinear_vote_numpy(self, events: np.ndarray, weight: Union[float, np.ndarray] = 1.0) -> np.ndarray:
        return self._vote(events, weight, 1e-8, False)

    def bilinear_vote_tensor(self, events: torch.Tensor, weight: FLOAT_TORCH = 1.0) -> torch.Tensor:
        return self._vote(events, weight, 1e-6, False)

    def count_event_numpy(self, events: np.ndarray) -> np.ndarray:
        return self._vote(events, 1.0, 1e-8, True)

    def count_event_tensor(self, events: torch.Tensor) -> torch.Tensor:
        # the reference's tensor count path raises on current torch (int64 values scatter-added into a
        # float image, lines 251-254); this follows the numpy path's semantics with the torch epsilon
        return self._vote(events, 1.0, 1e-6, True)
