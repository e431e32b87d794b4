"""MI355X-native contrast-maximization inner loop (event warp -> IWE -> contrast + gradient)
behind the Warp / EventImageConverter / costs API of tub-rip/event_based_optical_flow."""
from . import array_types as types  # reference name: src/types
from . import costs, data_loader, event_image_converter, functional, solver, utils, warp
from .cmax import CMaxHandle, ContrastObjective, make_descriptor
from .event_image_converter import EventImageConverter
from .warp import MotionModelKeyError, Warp

__all__ = ["Warp", "MotionModelKeyError", "EventImageConverter", "costs", "CMaxHandle", "ContrastObjective",
           "make_descriptor", "functional", "solver", "data_loader", "utils", "types", "warp", "event_image_converter"]
