"""Optimiser-side boundary of the path: `scipy_autograd.minimize` and the GPU objective for
patch-based flow.  The pyramid driver, Optuna initialisers and metrics of the reference's solver
classes are outside the hot path."""
from . import scipy_autograd
from .patch_objective import PatchFlowObjective, patch_pad
from .pyramid import PyramidalPatchContrastMaximization

# registry name of the reference (src/solver/__init__.py:14-19)
collections = {"pyramidal_patch_contrast_maximization": PyramidalPatchContrastMaximization}

__all__ = ["scipy_autograd", "PatchFlowObjective", "patch_pad", "PyramidalPatchContrastMaximization", "collections"]
