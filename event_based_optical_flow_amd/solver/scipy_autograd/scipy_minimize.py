"""`minimize(fun, x0, ...)`: scipy.optimize.minimize with gradients from torch autograd -- the
optimiser entry point of the reference (src/solver/scipy_autograd/scipy_minimize.py:6-19, 100-125),
same signature and return value (OptimizeResult with `.x` reshaped like x0)."""
import scipy.optimize as sopt

from .torch_wrapper import TorchWrapper

_NEEDS_HESSP = ("Newton-CG", "trust-ncg", "trust-krylov", "trust-constr")


def minimize(fun, x0, args=(), precision="float32", method=None, hvp_type=None, torch_device="cpu", bounds=None,
             constraints=None, tol=None, callback=None, options=None):
    wrapper = TorchWrapper(fun, precision=precision, hvp_type=hvp_type, device=torch_device)
    if bounds is not None:
        assert method in [None, "L-BFGS-B", "TNC", "SLSQP", "Powell", "trust-constr"], \
            "bounds are only available for L-BFGS-B, TNC, SLSQP, Powell, trust-constr"
    if constraints is not None:
        raise NotImplementedError("constraints are not built (no solver of the reference uses them)")
    if method in ("dogleg", "trust-exact"):
        raise NotImplementedError(f"{method} needs the full Hessian; only Hessian-vector products are built")
    if method == "Newton-CG" and options and "gtol" in options:
        # the reference hands Newton-CG its `gtol` (src/solver/patch_contrast_pyramid.py:300-303); SciPy's Newton-CG has no such
        # option (its tolerance is `xtol`), ignores it and warns "Unknown solver options: gtol" on every call: same run, no warning
        options = {k: v for k, v in options.items() if k != "gtol"}
    res = sopt.minimize(
        wrapper.get_value_and_grad,
        wrapper.get_input(x0),
        args=args,
        method=method,
        jac=True,
        hessp=wrapper.get_hvp if method in _NEEDS_HESSP else None,
        bounds=wrapper.get_bounds(bounds),
        tol=tol,
        callback=callback,
        options=options,
    )
    res.x = wrapper.get_output(res.x)
    if "jac" in res.keys() and len(res.jac) > 0:
        try:
            res.jac = wrapper.get_output(res.jac)
        except Exception:
            pass
    return res
