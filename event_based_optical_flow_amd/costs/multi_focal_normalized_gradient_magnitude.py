"""MultiFocalNormalizedGradientMagnitude (Shiba et al. ECCV 2022) --
reference: src/costs/multi_focal_normalized_gradient_magnitude.py:73-101."""
import logging

from . import CostBase, NormalizedGradientMagnitude

logger = logging.getLogger(__name__)


class MultiFocalNormalizedGradientMagnitude(CostBase):
    name = "multi_focal_normalized_gradient_magnitude"
    required_keys = ["forward_iwe", "backward_iwe", "middle_iwe", "omit_boundary", "orig_iwe"]

    def __init__(self, direction="minimize", store_history: bool = False, cuda_available=False, precision="32",
                 *args, **kwargs):
        super().__init__(direction=direction, store_history=store_history)
        self.gradient_loss = NormalizedGradientMagnitude(direction=direction, cuda_available=cuda_available,
                                                         precision=precision)

    def calculate(self, arg: dict):
        orig, omit = arg["orig_iwe"], arg["omit_boundary"]
        loss = self.gradient_loss.ratio(arg["forward_iwe"], orig, omit) + self.gradient_loss.ratio(arg["backward_iwe"], orig, omit)
        if arg.get("middle_iwe", None) is not None:
            loss = loss + self.gradient_loss.ratio(arg["middle_iwe"], orig, omit) * 2
        if self.direction in ["minimize", "natural"]:
            return loss
        logger.warning("The loss is specified as maximize direction")
        return -loss
