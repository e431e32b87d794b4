"""HybridCost: a weighted combination of named costs -- the plugin the shipped YAMLs select with `cost: hybrid`
(reference: src/costs/hybrid.py:14-79).

Every member is a `_Member` record (cost object + weight) with one `combine` rule: a numeric weight contributes
weight * value, the string "inv" contributes 1 / value (hybrid.py:51-53).  The reference's attribute `cost_func`
(name -> {"func", "weight"}) is kept as a read-only view for callers that look into it."""
import logging
from dataclasses import dataclass
from typing import Dict, Union

from . import CostBase, functions

logger = logging.getLogger(__name__)

Weight = Union[float, int, str]


def combine(weight: Weight, value):
    """Contribution of one member to the hybrid loss."""
    return 1.0 / value if weight == "inv" else weight * value


def combine_derivatives(weight: Weight, value):
    """(phi', phi'') of the member's contribution phi(value): chain factors of the gradient and of the exact
    Hessian-vector product  H v = phi' H_c v + phi'' <grad c, v> grad c."""
    if weight == "inv":
        return -1.0 / (value * value), 2.0 / (value * value * value)
    return weight, 0.0


@dataclass
class _Member:
    cost: CostBase
    weight: Weight


class HybridCost(CostBase):
    """cost_with_weight: {cost name: weight | "inv"}."""

    name = "hybrid"

    def __init__(self, direction: str, cost_with_weight: dict, store_history: bool = False, *args, **kwargs):
        logger.info(f"Log functions are mix of {cost_with_weight}")
        self._members: Dict[str, _Member] = {}
        for cost_name, weight in cost_with_weight.items():
            member_cost = functions[cost_name](direction=direction, store_history=store_history, *args, **kwargs)
            self._members[cost_name] = _Member(member_cost, weight)
        super().__init__(direction=direction, store_history=store_history)
        self.required_keys = [key for m in self._members.values() for key in m.cost.required_keys]

    @property
    def cost_func(self) -> dict:
        """The reference's view of the members: {name: {"func": cost object, "weight": weight}}."""
        return {name: {"func": m.cost, "weight": m.weight} for name, m in self._members.items()}

    def update_weight(self, cost_with_weight):
        assert set(self._members) == set(cost_with_weight)
        for cost_name, weight in cost_with_weight.items():
            self._members[cost_name].weight = weight

    def calculate(self, arg: dict):
        return sum((combine(m.weight, m.cost.calculate(arg)) for m in self._members.values()), 0.0)

    # the members keep their own histories, reported under their names
    def clear_history(self) -> None:
        self.history = {"loss": []}
        for m in getattr(self, "_members", {}).values():
            m.cost.clear_history()

    def get_history(self) -> dict:
        report = self.history.copy()
        report.update({name: m.cost.get_history()["loss"] for name, m in self._members.items()})
        return report

    def _set_history_register(self, on: bool) -> None:
        self.store_history = on
        for m in self._members.values():
            m.cost.store_history = on

    def enable_history_register(self) -> None:
        self._set_history_register(True)

    def disable_history_register(self) -> None:
        self._set_history_register(False)
