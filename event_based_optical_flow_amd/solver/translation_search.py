"""Grid search over 2-DoF translations on the GPU objective: K candidates per library call (cmax_objective_batch).

The reference's gradient-free paths score batches of sampled motions one objective evaluation at a time:
  * `initialize_guess_from_whole_image` / `initialize_guess_from_patch` (src/solver/patch_contrast_base.py:164-187, 126-162): every
    translation of a fixed grid -- np.arange(-150, 150, 10)^2 = 900 candidates on the whole image, np.arange(-150, 150, 30)^2 = 100 on
    one patch -- through `objective_scipy_for_patch` (244-271): motion * t_scale, the solver's own cost, first minimum wins;
  * Optuna's grid / uniform sampler over the `optimizer.parameters` box (src/solver/base.py:738-787).
Here the candidates of a grid go to the device in chunks; for the 2-DoF image-variance objective a chunk shares one launch of each
kernel (blockIdx.z = candidate), every other cost is evaluated candidate by candidate inside the same call.  Motions of this size
(150 px per unit time) want the batch in time slabs: `CMaxHandle.auto_time_slabs` is asked before the first chunk.
"""
from typing import Dict, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from ..cmax import CMaxHandle, ContrastObjective

WHOLE_IMAGE_FIELD = np.arange(-150, 150, 10)  # patch_contrast_base.py:166
ONE_PATCH_FIELD = np.arange(-150, 150, 30)    # patch_contrast_base.py:128


def candidate_losses(handle: CMaxHandle, candidates: np.ndarray, t_scale: float, cost: str = "image_variance",
                     cost_with_weight: Optional[Dict[str, Union[float, str]]] = None, sigma: float = 0.0, chunk: int = 32,
                     auto_slabs: bool = True) -> np.ndarray:
    """Loss of every candidate translation.  candidates [K, 2] in pixel per unit of the RAW timestamps (what the optimiser holds);
    the warp sees candidates * t_scale (objective_scipy_for_patch: `motion * t_scale` with normalised time).  total_variation members of
    a hybrid cost contribute exactly 0 for a single translation -- the reference hands the cost a [2, 1, 1] "flow" whose zero-padded
    Sobel response vanishes (src/costs/total_variation.py:110-126) -- and are skipped."""
    cand = np.ascontiguousarray(np.asarray(candidates, dtype=np.float64).reshape(-1, 2))
    if handle.n_events == 0:  # no events in the patch: loss 0 for every motion (patch_contrast_base.py:253-255)
        return np.zeros(cand.shape[0])
    thetas = cand * float(t_scale)
    if auto_slabs:
        handle.auto_time_slabs(float(np.abs(thetas).max()) if thetas.size else 0.0)
    obj = ContrastObjective(handle, "2d-translation", cost=cost, cost_with_weight=cost_with_weight, sigma=sigma)
    out = np.empty(cand.shape[0], dtype=np.float64)
    chunk = max(1, min(int(chunk), 64))
    for b in range(0, cand.shape[0], chunk):
        th = torch.from_numpy(thetas[b:b + chunk]).to(handle.device)  # fp64: crosses the ABI as it is (cells decided from these doubles)
        out[b:b + chunk] = obj.evaluate_candidates(th).cpu().numpy()
    return out


def grid_search_translation(handle: CMaxHandle, t_scale: float, field_x: Sequence[float] = WHOLE_IMAGE_FIELD,
                            field_y: Optional[Sequence[float]] = None, **cost_kwargs) -> Tuple[np.ndarray, np.ndarray]:
    """(best guess [2], losses [len(field_x), len(field_y)]): the double loop of initialize_guess_from_whole_image -- x outer, y inner,
    strict `<` so the FIRST minimum wins -- as one batch of candidates."""
    fx = np.asarray(field_x, dtype=np.float64)
    fy = fx if field_y is None else np.asarray(field_y, dtype=np.float64)
    gx, gy = np.meshgrid(fx, fy, indexing="ij")
    cand = np.stack([gx.reshape(-1), gy.reshape(-1)], axis=1)
    loss = candidate_losses(handle, cand, t_scale, **cost_kwargs)
    # A NaN candidate never wins: in initialize_guess_from_whole_image / _from_patch the loss is a torch tensor, calculate_cost's NaN -> 0.0
    # applies to numpy losses only (patch_contrast_base.py:283-286), and `loss < best_loss` is False for a NaN -- the candidate is skipped;
    # when every candidate is NaN the reference keeps its initial best_guess = zeros(2).  
    ranked = np.where(np.isnan(loss), np.inf, loss)
    if not np.isfinite(ranked).any():
        return np.zeros(2), loss.reshape(len(fx), len(fy))
    best = int(np.argmin(ranked))  # first occurrence, like the reference's `if loss < best_loss`
    return cand[best].copy(), loss.reshape(len(fx), len(fy))
