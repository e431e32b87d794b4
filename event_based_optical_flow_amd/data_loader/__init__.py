"""Dataset readers feeding `solver.optimize(events)`: the reference's registry
`data_loader.collections[name] -> class` (src/data_loader/__init__.py:14-30) with the MVSEC reader.  Host-side
file parsing only -- the events go to the GPU in `CMaxHandle.set_events`."""
import os

DATASET_ROOT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "datasets")

from .base import DataLoaderBase  # noqa: E402
from .mvsec import MvsecDataLoader  # noqa: E402

collections = {cls.NAME: cls for cls in (MvsecDataLoader,)}

__all__ = ["DataLoaderBase", "MvsecDataLoader", "collections", "DATASET_ROOT_DIR"]
