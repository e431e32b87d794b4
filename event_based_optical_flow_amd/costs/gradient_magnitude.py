"""GradientMagnitude (Gallego et al. CVPR 2019) -- reference: src/costs/gradient_magnitude.py:60-95.

mean(gx^2 + gy^2) of Sobel/8 with zero padding (SobelTorch, src/utils/stat_utils.py:50-83).  With
omit_boundary=True (what the solvers always pass, patch_contrast_base.py:290) the numpy branch's
cv2.Sobel border handling is invisible, so both kinds of input share one kernel."""
from . import CostBase
from ._contrast import GRADMAG, raw_contrast


class GradientMagnitude(CostBase):
    name = "gradient_magnitude"
    required_keys = ["iwe", "omit_boundary"]

    def __init__(self, direction="minimize", store_history: bool = False, cuda_available=False, precision="32",
                 *args, **kwargs):
        super().__init__(direction=direction, store_history=store_history)
        self.precision = precision

    def calculate(self, arg: dict):
        mag = self.magnitude(arg["iwe"], arg["omit_boundary"])
        return -mag if self.direction == "minimize" else mag

    def magnitude(self, iwe, omit_boundary):
        mag = raw_contrast(iwe, GRADMAG, omit_boundary)
        if self.precision == "64" and hasattr(mag, "double"):
            mag = mag.double()  # gradient_magnitude.py:65-66
        return mag
