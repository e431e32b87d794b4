"""ImageVariance (Gallego et al. CVPR 2018) -- reference: src/costs/image_variance.py:27-71."""
from . import CostBase
from ._contrast import VARIANCE, raw_contrast


class ImageVariance(CostBase):
    name = "image_variance"
    required_keys = ["iwe", "omit_boundary"]

    def __init__(self, direction="minimize", store_history: bool = False, *args, **kwargs):
        super().__init__(direction=direction, store_history=store_history)

    def calculate(self, arg: dict):
        """arg: {"iwe": [H,W] image, "omit_boundary": bool} -> -var ('minimize') or +var."""
        var = raw_contrast(arg["iwe"], VARIANCE, arg["omit_boundary"])
        return -var if self.direction == "minimize" else var
