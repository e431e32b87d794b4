"""TotalVariation regulariser of the (patch) flow -- reference: src/costs/total_variation.py.

torch branch: mean(|Sobel_4ch(flow)/8|) (lines 60-75, 110-126).  The numpy branch is 32x larger
(no /8, channels summed instead of averaged, 77-93 + 128-151): reproduced by rescaling."""
import logging

import numpy as np
import torch

from .. import functional as F
from ..array_types import to_device_tensor
from . import CostBase

logger = logging.getLogger(__name__)


class TotalVariation(CostBase):
    name = "total_variation"
    required_keys = ["flow", "omit_boundary"]

    def __init__(self, direction="minimize", store_history: bool = False, cuda_available=False, precision="32",
                 *args, **kwargs):
        super().__init__(direction=direction, store_history=store_history)

    def calculate(self, arg: dict):
        flow, omit = arg["flow"], arg["omit_boundary"]
        if isinstance(flow, torch.Tensor):
            t = to_device_tensor(flow, "flow")
            if t.dim() == 4 and t.shape[0] == 1:
                t = t[0]
            if t.dim() != 3:
                raise NotImplementedError("total_variation takes one [2, h, w] flow")
            tv = F.total_variation(t, omit)
            loss = tv if tv.device == flow.device else tv.to(flow.device)
            if self.direction == "minimize":
                return loss
            logger.warning("The loss is specified as maximize direction")
            return -loss
        elif isinstance(flow, np.ndarray):
            if flow.ndim == 4:
                raise NotImplementedError
            # numpy branch: crop rule differs (h>1 and w>1, line 143) and the value is 32x the torch one
            crop = omit and flow.shape[1] > 1 and flow.shape[2] > 1
            if crop and (flow.shape[1] <= 2 or flow.shape[2] <= 2):
                return float("nan")  # np.mean of an empty crop in the reference
            tv = F.total_variation(to_device_tensor(flow, "flow"), crop)
            loss = float(tv.item()) * 32.0
            return loss if self.direction == "minimize" else -loss
        e = f"Unsupported input type. {type(flow)}."
        logger.error(e)
        raise NotImplementedError(e)
