"""Cost protocol (reference: src/costs/base.py:11-77): class attributes `name`, `required_keys`;
constructor (direction, store_history, **kw); `calculate(arg: dict)`; loss history."""
import logging
from typing import Dict, List

import torch

logger = logging.getLogger(__name__)

_DIRECTIONS = ("minimize", "maximize", "natural")


def _with_history_and_key_check(func):
    """`calculate` wrapper: log + re-raise missing dict keys, append the loss to the history when
    store_history is on (costs/base.py:29-51).  NB the history append calls .item(): a device
    sync per evaluation, so it is opt-in."""

    def wrapper(self, arg: dict):
        try:
            loss = func(self, arg)
        except KeyError as e:
            logger.error("Input for the cost needs keys of:")
            logger.error(self.required_keys)
            raise e
        if self.store_history:
            self.history["loss"].append(self.get_item(loss))
        return loss

    wrapper.__wrapped__ = func
    return wrapper


class CostBase(object):
    """Base of every cost.  direction: 'minimize' | 'maximize' | 'natural'."""

    required_keys: List[str] = []

    def __init__(self, direction="minimize", store_history: bool = False, *args, **kwargs):
        if direction not in _DIRECTIONS:
            e = f"direction should be minimize, maximize, and natural. Got {direction}."
            logger.error(e)
            raise ValueError(e)
        self.direction = direction
        self.store_history = store_history
        self.clear_history()

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        if "calculate" in cls.__dict__ and not hasattr(cls.__dict__["calculate"], "__wrapped__"):
            cls.calculate = _with_history_and_key_check(cls.__dict__["calculate"])

    def get_item(self, loss) -> float:
        return loss.item() if isinstance(loss, torch.Tensor) else loss

    def clear_history(self) -> None:
        self.history: Dict[str, list] = {"loss": []}

    def get_history(self) -> dict:
        return self.history.copy()

    def enable_history_register(self) -> None:
        self.store_history = True

    def disable_history_register(self) -> None:
        self.store_history = False

    def calculate(self, arg: dict):
        raise NotImplementedError
