from .scipy_minimize import minimize
from .torch_wrapper import TorchWrapper

__all__ = ["minimize", "TorchWrapper"]
