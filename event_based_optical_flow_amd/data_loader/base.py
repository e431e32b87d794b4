"""What every dataset reader shares: the `data:` block of a YAML config and the method set the drivers call
(role of src/data_loader/base.py:13-79)."""
import os

import numpy as np


def _enabled(config: dict, key: str) -> bool:
    return bool(config.get(key, False))


def _abstract(name):
    def method(self, *args, **kwargs):
        raise NotImplementedError(f"{type(self).__name__}.{name}")

    method.__name__ = name
    return method


class DataLoaderBase:
    """config: height, width (sensor size); root (dataset directory; default <repo>/datasets); dataset (sub-directory
    name; default NAME); load_gt_flow + gt (ground-truth directory); undistort.
    A reader implements get_sequence(name) -> {file role: path}, load_event(i0, i1, cam) -> float64 [n,4]
    (x row, y column, t, p), index_to_time / time_to_index, load_optical_flow(t1, t2), load_calib()."""

    NAME = "example"

    def __init__(self, config: dict = {}):
        from . import DATASET_ROOT_DIR

        self._HEIGHT, self._WIDTH = config["height"], config["width"]
        self.root_dir = os.path.expanduser(config.get("root") or DATASET_ROOT_DIR)
        self.dataset_dir = os.path.join(self.root_dir, config.get("dataset") or self.NAME)
        self.dataset_files = {}
        self.gt_flow_available = False
        if _enabled(config, "load_gt_flow"):
            self.gt_flow_dir = os.path.expanduser(config["gt"])
            self.gt_flow_available = os.path.exists(self.gt_flow_dir)
        self.auto_undistort = _enabled(config, "undistort")

    def set_sequence(self, sequence_name: str) -> None:
        self.sequence_name = sequence_name
        self.dataset_files = self.get_sequence(sequence_name)

    get_sequence = _abstract("get_sequence")
    load_event = _abstract("load_event")
    load_calib = _abstract("load_calib")
    load_optical_flow = _abstract("load_optical_flow")
    index_to_time = _abstract("index_to_time")
    time_to_index = _abstract("time_to_index")


def empty_events() -> np.ndarray:
    return np.zeros((0, 4), dtype=np.float64)
