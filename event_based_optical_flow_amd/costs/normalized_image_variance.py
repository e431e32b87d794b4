"""NormalizedImageVariance (flow warp loss, Stoffregen et al. ECCV 2020) --
reference: src/costs/normalized_image_variance.py:50-64."""
import logging

from . import CostBase
from ._contrast import VARIANCE, raw_contrast

logger = logging.getLogger(__name__)


class NormalizedImageVariance(CostBase):
    name = "normalized_image_variance"
    required_keys = ["orig_iwe", "iwe", "omit_boundary"]

    def __init__(self, direction="minimize", store_history: bool = False, *args, **kwargs):
        super().__init__(direction=direction, store_history=store_history)

    def calculate(self, arg: dict):
        return self.ratio(arg["iwe"], arg["orig_iwe"], arg["omit_boundary"])

    def ratio(self, iwe, orig_iwe, omit_boundary):
        # only `iwe` is boundary-cropped; `orig_iwe` is used whole (lines 40-41)
        v_iwe = raw_contrast(iwe, VARIANCE, omit_boundary)
        v_orig = raw_contrast(orig_iwe, VARIANCE, False)
        if self.direction == "minimize":
            return v_orig / v_iwe
        logger.warning("The loss is specified as maximize direction")
        return v_iwe / v_orig
