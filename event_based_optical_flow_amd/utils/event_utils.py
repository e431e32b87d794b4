"""Synthetic event generators (inputs of the parity tests and of bench.py).

`generate_events` follows the distribution of the reference's generator
(src/utils/event_utils.py:18-47): integer pixel coordinates stored as floats, uniformly random
timestamps sorted ascending, polarity in {0,1}.  A seeded numpy Generator replaces the global
RNG so runs are reproducible on the GPU box."""
from typing import Optional, Tuple

import numpy as np


def generate_events(n_events: int, height: int, width: int, tmin: float = 0.0, tmax: float = 0.5,
                    dist: str = "uniform", seed: Optional[int] = None, dtype=np.float64) -> np.ndarray:
    """[n_events, 4] = (x row, y col, t, p)."""
    if dist != "uniform":
        raise NotImplementedError(dist)
    rng = np.random.default_rng(seed)
    ev = np.empty((n_events, 4), dtype=dtype)
    ev[:, 0] = rng.integers(0, height, n_events)
    ev[:, 1] = rng.integers(0, width, n_events)
    ev[:, 2] = np.sort(rng.uniform(tmin, tmax, n_events))
    ev[:, 3] = rng.integers(0, 2, n_events)
    return ev


def generate_structured_events(n_events: int, height: int, width: int, velocity: Tuple[float, float],
                               n_dots: int = 400, tmin: float = 0.0, tmax: float = 0.05, jitter: float = 0.6,
                               seed: Optional[int] = None, dtype=np.float64) -> np.ndarray:
    """Events emitted by `n_dots` dots translating with `velocity` (pixel per unit NORMALISED time,
    i.e. per batch period): warping with theta = velocity (2-DoF model, x' = x + dt*theta,
    dt = t_norm - t_ref) ... sharpens the IWE.  Well-conditioned inputs for gradient parity
    (hard part 2).  Coordinates are rounded to integer pixels like a sensor."""
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(tmin, tmax, n_events))
    tn = (t - tmin) / (tmax - tmin)
    dot = rng.integers(0, n_dots, n_events)
    cx = rng.uniform(0, height, n_dots)
    cy = rng.uniform(0, width, n_dots)
    # an event warped by +tn*v lands on its dot centre  =>  emitted at centre - tn*v
    x = cx[dot] - tn * velocity[0] + rng.normal(0, jitter, n_events)
    y = cy[dot] - tn * velocity[1] + rng.normal(0, jitter, n_events)
    ev = np.empty((n_events, 4), dtype=dtype)
    ev[:, 0] = np.clip(np.round(x), 0, height - 1)
    ev[:, 1] = np.clip(np.round(y), 0, width - 1)
    ev[:, 2] = t
    ev[:, 3] = rng.integers(0, 2, n_events)
    return ev
