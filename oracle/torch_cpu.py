"""TEST / BENCH INFRASTRUCTURE ONLY -- never imported by the product package.

The reference's differentiated path as it runs on a CPU: torch tensor ops + torch.autograd, multi-threaded by torch.
A restatement (not a copy) of
    Warp.warp_event_2dof_xy / warp_event_from_optical_flow     src/warp.py:483-522, 263-313   (direction "first", normalised dt)
    EventImageConverter.bilinear_vote_tensor                   src/event_image_converter.py:316-374
    ImageVariance.calculate_torch / GradientMagnitude + Sobel  src/costs/image_variance.py:27-71, src/costs/gradient_magnitude.py:60-76,
                                                               src/utils/stat_utils.py:50-83
    gradient by torch.autograd.grad                            src/solver/scipy_autograd/torch_wrapper.py:30-49
for sigma = 0 and the two un-normalised costs, which is what bench.py's cfg2 / cfg3 / cfg5 evaluate.  Checked against the C
oracle in tests/test_oracle_golden.py; bench.py times it next to the scalar C port so that the CPU figure a reader compares
with is the reference's own kind of code on this host's cores."""
import numpy as np
import torch

_SOBEL_ROW = torch.tensor([[-1.0, -2.0, -1.0], [0.0, 0.0, 0.0], [1.0, 2.0, 1.0]], dtype=torch.float64)


def _vote(x, y, size):
    H, W = size
    fx, fy = torch.floor(x + 1e-6), torch.floor(y + 1e-6)
    a, b = x - fx, y - fy
    ix, iy = fx.long(), fy.long()
    img = torch.zeros(H * W, dtype=x.dtype)
    for dr, dc, w in ((0, 0, (1 - a) * (1 - b)), (1, 0, a * (1 - b)), (0, 1, (1 - a) * b), (1, 1, a * b)):
        r, c = ix + dr, iy + dc
        ok = (r >= 0) & (r < H) & (c >= 0) & (c < W)
        img = img.scatter_add(0, torch.where(ok, r * W + c, torch.zeros_like(r)), torch.where(ok, w, torch.zeros_like(w)))
    return img.reshape(H, W)


def value_and_grad(events, motion, motion_model, image_size, cost="image_variance", dtype=torch.float64):
    """-> (loss, gradient w.r.t. motion) for "minimize" direction, omit_boundary True, sigma 0, reference time "first"."""
    H, W = int(image_size[0]), int(image_size[1])
    ev = torch.as_tensor(np.asarray(events), dtype=dtype)
    m = torch.as_tensor(np.asarray(motion), dtype=dtype).requires_grad_()
    t = ev[:, 2]
    dt = (t - t.min()) / (t.max() - t.min())
    if motion_model == "2d-translation":
        x, y = ev[:, 0] + dt * m[0], ev[:, 1] + dt * m[1]
    elif motion_model == "dense-flow":
        src = ev[:, 0].long() * W + ev[:, 1].long()
        flat = m.reshape(2, H * W)
        x, y = ev[:, 0] - dt * flat[0][src], ev[:, 1] - dt * flat[1][src]
    else:
        raise NotImplementedError(motion_model)
    inner = _vote(x, y, (H, W))[1:-1, 1:-1]
    if cost == "image_variance":
        loss = -torch.var(inner)
    elif cost == "gradient_magnitude":
        img = _vote(x, y, (H, W))[None, None]  # Sobel on the whole image (zero padding), statistics on the cropped result
        k = torch.stack([_SOBEL_ROW, _SOBEL_ROW.t()]).to(dtype)[:, None]
        g = torch.nn.functional.conv2d(img, k, padding=1) / 8.0
        g = g[..., 1:-1, 1:-1]
        loss = -torch.mean(g[0, 0] ** 2 + g[0, 1] ** 2)
    else:
        raise NotImplementedError(cost)
    (grad,) = torch.autograd.grad(loss, m)
    return float(loss.detach()), grad.numpy()
