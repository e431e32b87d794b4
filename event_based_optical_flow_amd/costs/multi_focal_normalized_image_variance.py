"""MultiFocalNormalizedImageVariance (Shiba et al. ECCV 2022) --
reference: src/costs/multi_focal_normalized_image_variance.py:64-91."""
import logging

from . import CostBase, NormalizedImageVariance

logger = logging.getLogger(__name__)


class MultiFocalNormalizedImageVariance(CostBase):
    name = "multi_focal_normalized_image_variance"
    required_keys = ["forward_iwe", "backward_iwe", "middle_iwe", "omit_boundary", "orig_iwe"]

    def __init__(self, direction="minimize", store_history: bool = False, *args, **kwargs):
        super().__init__(direction=direction, store_history=store_history)
        self.variance_loss = NormalizedImageVariance(direction=direction)

    def calculate(self, arg: dict):
        orig, omit = arg["orig_iwe"], arg["omit_boundary"]
        loss = self.variance_loss.ratio(arg["forward_iwe"], orig, omit) + self.variance_loss.ratio(arg["backward_iwe"], orig, omit)
        if arg.get("middle_iwe", None) is not None:
            loss = loss + self.variance_loss.ratio(arg["middle_iwe"], orig, omit) * 2
        if self.direction in ["minimize", "natural"]:
            return loss
        logger.warning("The loss is specified as maximize direction")
        return -loss
