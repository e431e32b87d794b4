"""Cost registry (reference: src/costs/__init__.py:23-38): `functions[name] -> class`, built by
walking CostBase's subclasses; HybridCost is imported last because it looks names up in it."""
from .base import CostBase
from .gradient_magnitude import GradientMagnitude
from .image_variance import ImageVariance
from .total_variation import TotalVariation
from .normalized_image_variance import NormalizedImageVariance
from .normalized_gradient_magnitude import NormalizedGradientMagnitude
from .multi_focal_normalized_image_variance import MultiFocalNormalizedImageVariance
from .multi_focal_normalized_gradient_magnitude import MultiFocalNormalizedGradientMagnitude


def inheritors(klass):
    found, stack = set(), [klass]
    while stack:
        for child in stack.pop().__subclasses__():
            if child not in found:
                found.add(child)
                stack.append(child)
    return found


functions = {k.name: k for k in inheritors(CostBase)}

from .hybrid import HybridCost  # noqa: E402
