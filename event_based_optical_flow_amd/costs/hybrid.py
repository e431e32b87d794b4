"""HybridCost: weighted sum of named costs -- reference: src/costs/hybrid.py:14-79."""
import logging

from . import CostBase, functions

logger = logging.getLogger(__name__)


class HybridCost(CostBase):
    """cost_with_weight: {cost name: weight | "inv"}; "inv" contributes 1 / cost (hybrid.py:51-53)."""

    name = "hybrid"

    def __init__(self, direction: str, cost_with_weight: dict, store_history: bool = False, *args, **kwargs):
        logger.info(f"Log functions are mix of {cost_with_weight}")
        self.cost_func = {
            key: {"func": functions[key](direction=direction, store_history=store_history, *args, **kwargs), "weight": value}
            for key, value in cost_with_weight.items()
        }
        super().__init__(direction=direction, store_history=store_history)
        self.required_keys = []
        for entry in self.cost_func.values():
            self.required_keys.extend(entry["func"].required_keys)

    def update_weight(self, cost_with_weight):
        assert set(self.cost_func.keys()) == set(cost_with_weight.keys())
        for key, value in cost_with_weight.items():
            self.cost_func[key]["weight"] = value

    def calculate(self, arg: dict):
        loss = 0.0
        for entry in self.cost_func.values():
            value = entry["func"].calculate(arg)
            loss = loss + (1.0 / value if entry["weight"] == "inv" else entry["weight"] * value)
        return loss

    def clear_history(self) -> None:
        self.history = {"loss": []}
        for entry in getattr(self, "cost_func", {}).values():
            entry["func"].clear_history()

    def get_history(self) -> dict:
        dic = self.history.copy()
        for name, entry in self.cost_func.items():
            dic.update({name: entry["func"].get_history()["loss"]})
        return dic

    def enable_history_register(self) -> None:
        self.store_history = True
        for entry in self.cost_func.values():
            entry["func"].store_history = True

    def disable_history_register(self) -> None:
        self.store_history = False
        for entry in self.cost_func.values():
            entry["func"].store_history = False
