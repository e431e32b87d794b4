"""GPU tests of the optimiser-side boundary: patch -> dense interpolation, the whole solver objective
(values the reference's PyramidalPatchContrastMaximization.objective_scipy produced), the
Hessian-vector product handed to Newton-CG, and an end-to-end minimisation."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import event_based_optical_flow_amd as E  # noqa: E402
from event_based_optical_flow_amd import functional as F  # noqa: E402
from event_based_optical_flow_amd.solver import PatchFlowObjective, patch_pad  # noqa: E402
from event_based_optical_flow_amd.solver.scipy_autograd import TorchWrapper, minimize  # noqa: E402
from oracle import oracle as orc  # noqa: E402

TOL = 1e-4
YAML_HYBRID = {"multi_focal_normalized_gradient_magnitude": 1.0, "total_variation": 0.01}


def rel_max(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(np.asarray(b)).max()


@pytest.mark.parametrize("tag", ["plain", "burgers"])
@pytest.mark.parametrize("scale", [1, 3])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_patch_to_dense_golden(golden, tag, scale, dtype):
    g = golden("solver_objective")
    k = f"{tag}_s{scale}"
    size = tuple(int(v) for v in g["image_size"])
    pis, ps, sw, shift = g[k + "__patch_image_size"], g[k + "__patch_size"], g[k + "__sliding_window"], g[tag + "__patch_shift"]
    pad = patch_pad(ps, sw, shift)
    m = torch.tensor(g[k + "__x"].reshape(2, *pis), dtype=dtype, device="cuda", requires_grad=True)
    dense = F.patch_to_dense(m, size, sw, pad)
    tol = 1e-12 if dtype == torch.float64 else 1e-5
    np.testing.assert_allclose(dense.detach().cpu().numpy(), g[k + "__dense"], rtol=tol, atol=tol * 300)
    cot = np.random.default_rng(0).normal(size=(2,) + size)
    (gm,) = torch.autograd.grad((dense * torch.tensor(cot, dtype=dtype, device="cuda")).sum(), m)
    ref = orc.patch_to_dense_adj(cot, pis, sw, pad)
    np.testing.assert_allclose(gm.cpu().numpy(), ref, rtol=1e-11 if dtype == torch.float64 else 1e-4,
                               atol=(1e-11 if dtype == torch.float64 else 1e-4) * np.abs(ref).max())


@pytest.mark.parametrize("tag", ["plain", "burgers"])
@pytest.mark.parametrize("scale", [1, 3])
def test_solver_objective_golden(golden, tag, scale):
    """x -> loss and d loss / d x of the shipped YAML objective, against the reference solver."""
    g = golden("solver_objective")
    k = f"{tag}_s{scale}"
    size = tuple(int(v) for v in g["image_size"])
    ev = g["events"]
    h = E.CMaxHandle(size).set_events(ev, time_bin=10 if tag == "burgers" else 0)
    t_scale = ev[:, 2].max() - ev[:, 2].min()
    obj = PatchFlowObjective(h, t_scale, g[k + "__patch_image_size"], g[k + "__patch_size"], g[k + "__sliding_window"],
                             g[tag + "__patch_shift"], cost="hybrid", cost_with_weight=YAML_HYBRID, blur_sigma=1,
                             time_aware=(tag == "burgers"), time_bin=10, flow_interpolation="burgers",
                             t0_flow_location="middle")
    x = torch.tensor(g[k + "__x"], dtype=torch.float64, device="cuda", requires_grad=True)
    loss = obj(x)
    (grad,) = torch.autograd.grad(loss, x)
    assert abs(loss.item() - g[k + "__loss"]) <= TOL * abs(g[k + "__loss"])
    assert rel_max(grad.cpu().numpy(), g[k + "__grad"]) <= TOL


def test_hvp_is_the_difference_quotient_of_the_exact_gradient(golden):
    """Newton-CG's hessp is a central difference of the analytic HIP gradient (TorchWrapper.get_hvp).
    It is checked against the same difference quotient of the fp64 oracle gradient.  It is NOT
    expected to equal the reference's autograd vhp (golden hvp.npz keeps that value for the exact
    HVP kernel of SURVEY section 8f): autograd differentiates the bilinear weights inside fixed
    pixel cells (floor has zero derivative), whereas a finite step lets events cross cell borders,
    where the gradient of the tent-kernel vote jumps."""
    g = golden("hvp")
    size = tuple(int(v) for v in g["image_size"])
    h = E.CMaxHandle(size).set_events(g["events"])
    obj = E.ContrastObjective(h, "2d-translation", cost="image_variance", sigma=1)
    w = TorchWrapper(obj, precision="float64", device="cuda")
    x = w.get_input(g["theta"])
    loss, _ = w.get_value_and_grad(x)
    assert abs(float(loss) - g["loss"]) <= TOL * abs(g["loss"])
    v = g["v"]
    hv = w.get_hvp(x, v)
    step = w.hvp_eps * (1.0 + np.abs(x).max()) / np.abs(v).max()

    def oracle_grad(theta):
        return orc.objective(g["events"], theta, "2d-translation", size, cost="image_variance", sigma=1)["grad"]

    ref = (oracle_grad(x + step * v) - oracle_grad(x - step * v)) / (2 * step)
    assert rel_max(hv, ref) <= 2e-3, (hv, ref)


@pytest.mark.parametrize("method", ["BFGS", "Newton-CG"])
def test_minimize_recovers_the_generating_velocity(method):
    """End to end through the reference's optimiser protocol: events of dots moving with a known
    2-DoF velocity; contrast maximisation must recover it."""
    size, vel = (96, 128), np.array([9.0, -6.0])
    ev = E.utils.generate_structured_events(60000, size[0], size[1], tuple(vel), n_dots=120, jitter=0.3, seed=5)
    h = E.CMaxHandle(size).set_events(ev)
    obj = E.ContrastObjective(h, "2d-translation", cost="image_variance", sigma=1)
    obj.device = torch.device("cuda")
    res = minimize(obj, vel * 0.7, method=method, precision="float64", torch_device="cuda",
                   options={"gtol": 1e-7, "maxiter": 60})
    # the optimum of the pixel-rounded, border-clipped event set sits within a fraction of a pixel of `vel`
    assert np.abs(res.x - vel).max() < 0.35, res
    assert np.linalg.norm(res.jac) < 0.05
