"""Cell-border events: the one place where an fp32 evaluation of the reference's objective cannot be compared with its
fp64 value at 1e-4.

`bilinear_vote` assigns an event to the cell floor(x' + 1e-6) (src/event_image_converter.py:340).  The IMAGE is
continuous across a cell border (the weights go to 0 / 1), so the IWE and the loss are unaffected, but the GRADIENT
is not: dL/dx' = (1-b)(G10-G00) + b(G11-G01) takes its differences of dL/dIWE from the event's own cell.  An event
whose warped coordinate lies within fp32 rounding of a border (displacement = dt * flow in fp32: ~1e-6 px absolute
at 20 px) lands in the neighbouring cell in ANY fp32 evaluation -- the reference's own fp32 path included -- and
its term of the gradient then comes from the other side of the kink.  For the 1M-5M event configurations that is a
few tens of events out of millions.

`ambiguity_bound` computes, in fp64 from the oracle's own intermediate results, how much the gradient can move if every
such event took the other cell: per source pixel (dense / voxel) or summed (2-DoF).  The parity gate of the full-size
tests is then  |g - g_ref| <= 1e-4 max|g_ref| + bound  element-wise, with the number of ambiguous events reported.
"""
import numpy as np

from oracle import oracle as orc

MARGIN = 2e-5  # > fp32 error of x' = x + dt * f at |dt * f| <= 40 px (ulp 3.8e-6, three roundings) + the floor's 1e-6


def _event_grad(Gz, fx, fy, xw, yw):
    """(dL/dx', dL/dy') of events at (xw, yw) taken from cell (fx, fy); Gz = dL/dIWE zero-padded by 2 on every side."""
    a, b = xw - fx, yw - fy
    r, c = fx.astype(np.int64) + 2, fy.astype(np.int64) + 2
    ok = (r >= 0) & (r < Gz.shape[0] - 1) & (c >= 0) & (c < Gz.shape[1] - 1)
    r, c = np.where(ok, r, 0), np.where(ok, c, 0)
    g00, g10, g01, g11 = Gz[r, c], Gz[r + 1, c], Gz[r, c + 1], Gz[r + 1, c + 1]
    gx = (1 - b) * (g10 - g00) + b * (g11 - g01)
    gy = (1 - a) * (g01 - g00) + a * (g11 - g10)
    return np.where(ok, gx, 0.0), np.where(ok, gy, 0.0)


def ambiguity_bound(events, motion, model, size, G, direction="first", margin=MARGIN):
    """-> (bound, n_ambiguous).  bound has the gradient's shape: [2] | [2,H,W] | [T,2,H,W].  G: dL/d(raw IWE) of the
    reference time `direction`, [H, W] (no padding)."""
    ev = np.asarray(events, dtype=np.float64)
    warped, aux = orc.warp_event(ev, motion, model, direction, size)
    xs, ys = warped[:, 0] + 1e-6, warped[:, 1] + 1e-6
    fx, fy = np.floor(xs), np.floor(ys)
    ax, ay = xs - fx, ys - fy
    amb_x = np.minimum(ax, 1 - ax) < margin
    amb_y = np.minimum(ay, 1 - ay) < margin
    idx = np.nonzero(amb_x | amb_y)[0]
    m = np.asarray(motion)
    bound = np.zeros(m.shape if model != "2d-translation" else 2)
    if len(idx) == 0:
        return bound, 0
    Gz = np.pad(np.asarray(G, dtype=np.float64), 2)
    xw, yw, dt = warped[idx, 0], warped[idx, 1], aux["dt"][idx]
    fx, fy, ax, ay = fx[idx], fy[idx], ax[idx], ay[idx]
    # the neighbouring cell along each ambiguous axis: below the border -> the next cell, just above it -> the previous one
    sx = np.where(amb_x[idx], np.where(ax > 0.5, 1.0, -1.0), 0.0)
    sy = np.where(amb_y[idx], np.where(ay > 0.5, 1.0, -1.0), 0.0)
    gx0, gy0 = _event_grad(Gz, fx, fy, xw, yw)
    dgx, dgy = np.zeros_like(gx0), np.zeros_like(gy0)
    for ux, uy in ((1, 0), (0, 1), (1, 1)):
        gx1, gy1 = _event_grad(Gz, fx + ux * sx, fy + uy * sy, xw, yw)
        dgx = np.maximum(dgx, np.abs(gx1 - gx0))
        dgy = np.maximum(dgy, np.abs(gy1 - gy0))
    bx, by = np.abs(dt) * dgx, np.abs(dt) * dgy
    if model == "2d-translation":
        bound[0], bound[1] = bx.sum(), by.sum()
    else:
        ix, iy = ev[idx, 0].astype(np.int64), ev[idx, 1].astype(np.int64)
        if model == "dense-flow":
            np.add.at(bound[0], (ix, iy), bx)
            np.add.at(bound[1], (ix, iy), by)
        else:
            bins = aux["bin"][idx]
            keep = bins >= 0
            np.add.at(bound[:, 0], (bins[keep], ix[keep], iy[keep]), bx[keep])
            np.add.at(bound[:, 1], (bins[keep], ix[keep], iy[keep]), by[keep])
    return bound, int(len(idx))


def raw_image_grad(ref, sigma, key="iwe"):
    """dL/d(raw votes) of reference time `key` from an orc.objective result (the blur transposed back if sigma > 0)."""
    G = 0
    for k, g in ref["image_grads"].items():
        if (("iwe" if k == "backward_iwe" else k) == key):
            G = G + g
    return orc.blur3_adj(G, sigma) if sigma > 0 else G
