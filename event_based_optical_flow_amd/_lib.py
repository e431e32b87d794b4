"""ctypes binding of libcmax_hip.so (the C ABI declared in include/cmax_hip.h).

The HIP library is the product: there is no CPU or PyTorch fallback.  If the shared object cannot
be loaded, or a compute entry point is called without a GPU, this module raises -- loudly.
"""
import ctypes
import os
import threading

from . import build as _build

c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_dbl = ctypes.c_double
c_vp = ctypes.c_void_p


class CmaxError(RuntimeError):
    """A libcmax_hip entry point returned non-zero (negative: bad argument, positive: hipError_t)."""

    def __init__(self, code, message):
        super().__init__(f"libcmax_hip error {code}: {message}")
        self.code = code


class CmaxObjective(ctypes.Structure):
    """Mirror of cmax_objective_t (include/cmax_hip.h)."""

    _fields_ = [
        ("model", ctypes.c_int32),
        ("cost", ctypes.c_int32),
        ("normalized", ctypes.c_int32),
        ("minimize", ctypes.c_int32),
        ("negate", ctypes.c_int32),
        ("omit_boundary", ctypes.c_int32),
        ("normalize_t", ctypes.c_int32),
        ("n_ref", ctypes.c_int32),
        ("ref_mode", ctypes.c_int32 * 4),
        ("ref_frac", ctypes.c_double * 4),
        ("mult", ctypes.c_double * 4),
        ("sigma", ctypes.c_double),
        ("T", ctypes.c_int32),
        ("motion_dtype", ctypes.c_int32),
    ]


class CmaxPatchObjective(ctypes.Structure):
    """Mirror of cmax_patch_objective_t (include/cmax_hip.h)."""

    _fields_ = [
        ("n_terms", ctypes.c_int32),
        ("time_aware", ctypes.c_int32),
        ("T", ctypes.c_int32),
        ("scheme", ctypes.c_int32),
        ("t0", ctypes.c_int32),
        ("H", ctypes.c_int32),
        ("W", ctypes.c_int32),
        ("ph", ctypes.c_int32),
        ("pw", ctypes.c_int32),
        ("sw_h", ctypes.c_int32),
        ("sw_w", ctypes.c_int32),
        ("pad_h", ctypes.c_int32),
        ("pad_w", ctypes.c_int32),
        ("tv_omit_boundary", ctypes.c_int32),
        ("t_scale", ctypes.c_double),
        ("weight", ctypes.c_double * 4),
        ("tv_weight", ctypes.c_double),
        ("term", CmaxObjective * 4),
    ]


# constants (include/cmax_hip.h)
F32, F64 = 0, 1
MODEL_2DOF, MODEL_DENSE, MODEL_VOXEL = 0, 1, 2
REF_FIRST, REF_LAST, REF_FRAC = 0, 1, 2
COST_VARIANCE, COST_GRADMAG = 0, 1
SCHEME_BURGERS, SCHEME_UPWIND = 0, 1
ABI_VERSION = 3
COMM_ID_BYTES = 128
RAW_LINES = 32
RAW_DOUBLES = RAW_LINES * 16
PROF_CLASSES = 6
ECOMM = -5
ENODEV = -4

# every symbol include/cmax_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "cmax_last_error": (ctypes.c_char_p, []),
    "cmax_abi_version": (c_int, []),
    "cmax_tminmax": (c_int, [c_vp, c_int, c_i64, c_vp, c_vp]),
    "cmax_warp_events": (c_int, [c_vp, c_int, c_i64, c_int, c_vp, c_int, c_int, c_int, c_vp, c_int, c_dbl, c_int,
                                 c_vp, c_vp, c_vp, c_vp]),
    "cmax_warp_events_bwd": (c_int, [c_vp, c_int, c_i64, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "cmax_vote": (c_int, [c_vp, c_int, c_i64, c_i64, c_vp, c_dbl, c_int, c_int, c_int, c_int, c_dbl, c_int, c_vp, c_vp]),
    "cmax_vote_bwd": (c_int, [c_vp, c_int, c_i64, c_i64, c_vp, c_dbl, c_int, c_int, c_int, c_int, c_dbl, c_vp, c_vp,
                              c_vp, c_vp]),
    "cmax_blur3": (c_int, [c_vp, c_int, c_int, c_int, c_dbl, c_int, c_vp, c_vp]),
    "cmax_gaussian_filter": (c_int, [c_vp, c_int, c_int, c_int, c_dbl, c_vp, c_vp, c_vp]),
    "cmax_contrast": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "cmax_total_variation": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "cmax_flow_step": (c_int, [c_vp, c_int, c_int, c_int, c_dbl, c_int, c_vp, c_vp]),
    "cmax_set_leaf_deterministic": (c_int, [c_int]),
    "cmax_flow_step_adj": (c_int, [c_vp, c_int, c_int, c_int, c_dbl, c_int, c_vp, c_vp, c_vp]),
    "cmax_voxel_construct": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "cmax_voxel_construct_adj": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "cmax_voxel_construct_tan": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "cmax_voxel_construct_adj_tan": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "cmax_patch_to_dense": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "cmax_create": (c_int, [c_int, c_int, c_int, c_int, ctypes.POINTER(c_vp)]),
    "cmax_destroy": (c_int, [c_vp]),
    "cmax_set_events": (c_int, [c_vp, c_vp, c_int, c_i64, c_int, c_dbl, c_dbl, c_int, c_vp]),
    "cmax_set_time_bins": (c_int, [c_vp, c_int, c_vp]),
    "cmax_set_time_slabs": (c_int, [c_vp, c_int, c_vp]),
    "cmax_set_keep_outside": (c_int, [c_vp, c_int]),
    "cmax_batch_outside": (c_int, [c_vp, ctypes.POINTER(ctypes.c_int64)]),
    "cmax_iwe": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_dbl, c_int, c_dbl, c_vp, c_vp]),
    "cmax_objective": (c_int, [c_vp, ctypes.POINTER(CmaxObjective), c_vp, c_vp, c_vp, c_vp]),
    "cmax_objective_host": (c_int, [c_vp, ctypes.POINTER(CmaxObjective), c_vp, c_vp, c_vp, c_vp]),
    "cmax_objective_has_raw": (c_int, [c_vp, ctypes.POINTER(CmaxObjective)]),
    "cmax_objective_raw": (c_int, [c_vp, ctypes.POINTER(CmaxObjective), c_vp, c_vp, c_vp]),
    "cmax_finalize_raw_host": (c_int, [c_vp, ctypes.POINTER(CmaxObjective), c_vp, c_vp, c_vp]),
    "cmax_objective_hvp": (c_int, [c_vp, ctypes.POINTER(CmaxObjective), c_vp, c_vp, c_vp, c_vp]),
    "cmax_objective_batch": (c_int, [c_vp, ctypes.POINTER(CmaxObjective), c_vp, c_int, c_vp, c_vp, c_vp]),
    "cmax_objective_hvp_dist": (c_int, [c_vp, ctypes.POINTER(CmaxObjective), c_vp, c_vp, c_vp, c_vp]),
    "cmax_objective_vote": (c_int, [c_vp, ctypes.POINTER(CmaxObjective), c_vp, c_vp, ctypes.POINTER(c_int), c_vp]),
    "cmax_objective_finish": (c_int, [c_vp, ctypes.POINTER(CmaxObjective), c_vp, c_vp, c_int, c_vp, c_vp, c_vp]),
    "cmax_set_deterministic": (c_int, [c_vp, c_int]),
    "cmax_get_deterministic": (c_int, [c_vp, ctypes.POINTER(c_int)]),
    "cmax_comm_available": (c_int, [ctypes.c_char_p, c_int]),
    "cmax_comm_set_c2_bands": (c_int, [c_vp, c_int]),
    "cmax_comm_unique_id": (c_int, [c_vp]),
    "cmax_comm_init": (c_int, [c_vp, c_vp, c_int, c_int]),
    "cmax_comm_destroy": (c_int, [c_vp]),
    "cmax_comm_info": (c_int, [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "cmax_comm_allreduce": (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp]),
    "cmax_objective_dist": (c_int, [c_vp, ctypes.POINTER(CmaxObjective), c_vp, c_vp, c_vp, c_vp]),
    "cmax_read_profile_all": (c_int, [c_vp, ctypes.POINTER(c_dbl), ctypes.POINTER(c_i64)]),
    "cmax_sizeof_objective": (c_int, []),
    "cmax_set_profiling": (c_int, [c_vp, c_int]),
    "cmax_read_profile": (c_int, [c_vp, ctypes.POINTER(c_dbl), ctypes.POINTER(c_i64)]),
    "cmax_copy_iwe": (c_int, [c_vp, c_int, c_vp, c_vp]),
    "cmax_handle_info": (c_int, [c_vp, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    "cmax_debug_launch_floor": (c_int, [c_vp, c_int, c_vp]),
    "cmax_debug_timeline": (c_int, [c_vp]),
    "cmax_debug_packed_events": (c_int, [c_vp, c_vp, c_vp, ctypes.POINTER(c_int), c_vp]),
    "cmax_batch_info": (c_int, [c_vp, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "cmax_work_list_info": (c_int, [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "cmax_sizeof_patch_objective": (c_int, []),
    "cmax_patch_plan_create": (c_int, [c_vp, ctypes.POINTER(CmaxPatchObjective), ctypes.POINTER(c_vp)]),
    "cmax_patch_plan_destroy": (c_int, [c_vp]),
    "cmax_patch_plan_info": (c_int, [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "cmax_patch_plan_evaluate": (c_int, [c_vp, c_vp, c_int, ctypes.POINTER(c_dbl), c_vp, c_vp]),
    "cmax_patch_plan_hvp": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp]),
    "cmax_patch_plan_set_t_scale": (c_int, [c_vp, c_dbl]),
    "cmax_patch_search": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_int, c_vp, c_dbl, c_vp, c_vp, c_vp]),
}

_lock = threading.Lock()
_lib = None


def library_path() -> str:
    return _build.LIB_PATH


def load():
    """Load (building in-tree if the .so is missing) and type the library.  Raises on failure."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        # torch first: libcmax_hip.so must bind to the libamdhip64.so.7 torch already loaded so that
        # torch's device pointers and streams are valid inside the library.
        import torch  # noqa: F401

        path = _build.LIB_PATH
        if not os.path.exists(path):
            path = _build.build_library()
        try:
            lib = ctypes.CDLL(path)
        except OSError as e:  # pragma: no cover - environment dependent
            raise RuntimeError(
                f"libcmax_hip.so could not be loaded from {path}: {e}. The HIP extension is the product; "
                "there is no CPU fallback. Build it with `python -m event_based_optical_flow_amd.build`."
            ) from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if lib.cmax_sizeof_objective() != ctypes.sizeof(CmaxObjective):
            raise RuntimeError("cmax_objective_t layout mismatch between libcmax_hip.so and the ctypes binding")
        if lib.cmax_sizeof_patch_objective() != ctypes.sizeof(CmaxPatchObjective):
            raise RuntimeError("cmax_patch_objective_t layout mismatch between libcmax_hip.so and the ctypes binding")
        if lib.cmax_abi_version() != ABI_VERSION:
            raise RuntimeError(f"libcmax_hip ABI {lib.cmax_abi_version()} != binding ABI {ABI_VERSION}")
        _lib = lib
    return _lib


def check(rc: int):
    if rc != 0:
        msg = load().cmax_last_error()
        raise CmaxError(rc, msg.decode("utf-8", "replace") if msg else "")


def require_gpu():
    """The product path needs a real device; never degrade silently."""
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError(
            "event_based_optical_flow_amd needs an AMD GPU (gfx950): torch.cuda.is_available() is False and "
            "there is no CPU fallback for the contrast-maximization kernels."
        )
