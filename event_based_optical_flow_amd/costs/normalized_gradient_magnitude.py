"""NormalizedGradientMagnitude -- reference: src/costs/normalized_gradient_magnitude.py:63-79."""
import logging

from . import CostBase, GradientMagnitude

logger = logging.getLogger(__name__)


class NormalizedGradientMagnitude(CostBase):
    name = "normalized_gradient_magnitude"
    required_keys = ["orig_iwe", "iwe", "omit_boundary"]

    def __init__(self, direction="minimize", store_history: bool = False, cuda_available=False, precision="32",
                 *args, **kwargs):
        super().__init__(direction=direction, store_history=store_history)
        self.gradient_magnitude = GradientMagnitude(direction=direction, store_history=store_history,
                                                    cuda_available=cuda_available, precision=precision)

    def calculate(self, arg: dict):
        return self.ratio(arg["iwe"], arg["orig_iwe"], arg["omit_boundary"])

    def ratio(self, iwe, orig_iwe, omit_boundary):
        m_iwe = self.gradient_magnitude.magnitude(iwe, omit_boundary)
        m_orig = self.gradient_magnitude.magnitude(orig_iwe, omit_boundary)
        if self.direction == "minimize":
            return m_orig / m_iwe
        logger.warning("The loss is specified as maximize direction")
        return m_iwe / m_orig
