"""Import the reference (tub-rip/event_based_optical_flow) in THIS build container only.

Used by gen_golden.py to produce the fixtures under tests/golden/*.npz.  Never imported by the
tests themselves (the reference does not exist on the GPU box).  No reference source is copied:
the reference is imported in place from /root/reference.

Four third-party modules the reference imports are absent from this image (cv2, optuna, skimage,
torchvision).  They are replaced by empty stub modules, plus three behavioural shims which are the
ONLY non-reference arithmetic that can reach a fixture (recorded in every fixture's `shims` field):

  (1) torchvision.transforms.functional.gaussian_blur(img, kernel_size=3, sigma)
        -> reflect-pad(1) + depthwise conv with the outer product of the normalised taps
           exp(-0.5 (k/sigma)^2), k in {-1,0,1}   (torchvision's published algorithm)
  (2) torchvision.transforms.functional.resize(img, size, BILINEAR)
        -> F.interpolate(mode="bilinear", align_corners=False)
  (3) cv2.Sobel(img, CV_64F, dx, dy, ksize=3)
        -> separable [-1,0,1] (x) [1,2,1] correlation, BORDER_REFLECT_101 (scipy 'mirror')
"""
import sys
import types

import numpy as np
import torch

REFERENCE_ROOT = "/root/reference"
SHIMS = "gaussian_blur3(reflect101); resize(bilinear,align_corners=False); cv2.Sobel(mirror)"


def _gaussian_blur(img, kernel_size=3, sigma=None):
    if isinstance(kernel_size, (list, tuple)):
        kernel_size = kernel_size[0]
    if isinstance(sigma, (list, tuple)):
        sigma = sigma[0]
    half = (kernel_size - 1) * 0.5
    x = torch.linspace(-half, half, steps=kernel_size, dtype=img.dtype, device=img.device)
    pdf = torch.exp(-0.5 * (x / sigma) ** 2)
    k1 = pdf / pdf.sum()
    k2 = (k1[:, None] * k1[None, :])[None, None]
    pad = kernel_size // 2
    squeeze = False
    if img.dim() == 3:
        img = img[None]
        squeeze = True
    c = img.shape[1]
    out = torch.nn.functional.conv2d(
        torch.nn.functional.pad(img, (pad, pad, pad, pad), mode="reflect"), k2.expand(c, 1, -1, -1), groups=c
    )
    return out[0] if squeeze else out


def _resize(img, size, interpolation=None, **_kw):
    squeeze = img.dim() == 3
    if squeeze:
        img = img[None]
    out = torch.nn.functional.interpolate(img, size=list(size), mode="bilinear", align_corners=False)
    return out[0] if squeeze else out


def _sobel(img, ddepth, dx, dy, ksize=3):
    from scipy.ndimage import correlate1d

    img = np.asarray(img, dtype=np.float64)
    d = np.array([-1.0, 0.0, 1.0])
    s = np.array([1.0, 2.0, 1.0])
    # cv2: dx differentiates along columns (axis 1), dy along rows (axis 0)
    if dx == 1 and dy == 0:
        return correlate1d(correlate1d(img, d, axis=1, mode="mirror"), s, axis=0, mode="mirror")
    if dx == 0 and dy == 1:
        return correlate1d(correlate1d(img, d, axis=0, mode="mirror"), s, axis=1, mode="mirror")
    raise NotImplementedError


def install_stubs():
    if "cv2" not in sys.modules:
        cv2 = types.ModuleType("cv2")
        cv2.CV_64F = 6
        cv2.INTER_LINEAR = 1
        cv2.INTER_NEAREST = 0
        cv2.Sobel = _sobel
        sys.modules["cv2"] = cv2
    if "optuna" not in sys.modules:
        optuna = types.ModuleType("optuna")
        for sub in ("storages", "distributions", "study", "logging", "samplers", "trial"):
            m = types.ModuleType("optuna." + sub)
            setattr(optuna, sub, m)
            sys.modules["optuna." + sub] = m
        optuna.storages.InMemoryStorage = type("InMemoryStorage", (), {})
        optuna.distributions.BaseDistribution = type("BaseDistribution", (), {})
        optuna.study.Study = type("Study", (), {})
        optuna.trial.Trial = type("Trial", (), {})
        optuna.logging.WARNING = 30
        optuna.logging.set_verbosity = lambda *_a, **_k: None
        sys.modules["optuna"] = optuna
    if "skimage" not in sys.modules:
        skimage = types.ModuleType("skimage")
        skimage.transform = types.ModuleType("skimage.transform")
        sys.modules["skimage"] = skimage
        sys.modules["skimage.transform"] = skimage.transform
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tr = types.ModuleType("torchvision.transforms")
        fn = types.ModuleType("torchvision.transforms.functional")
        fn.gaussian_blur = _gaussian_blur
        fn.resize = _resize
        tr.functional = fn
        tr.InterpolationMode = types.SimpleNamespace(BILINEAR="bilinear", NEAREST="nearest")
        tv.transforms = tr
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.transforms"] = tr
        sys.modules["torchvision.transforms.functional"] = fn


def import_reference():
    """Returns the reference's `src` package (warp, event_image_converter, costs, utils, solver)."""
    import warnings

    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import src  # noqa: F401
        from src import costs, event_image_converter, solver, utils, warp  # noqa: F401
    return src
