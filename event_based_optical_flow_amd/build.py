"""Builds libcmax_hip.so (gfx950) in-tree with hipcc.  No GPU needed (cross-compiles)."""
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
# CMAX_LIB: alternative output / load path (tuning experiments build variants side by side)
LIB_PATH = os.environ.get("CMAX_LIB", os.path.join(PKG_DIR, "libcmax_hip.so"))
SOURCES = ["cmax_leaf.hip", "cmax_flow.hip", "cmax_fused.hip", "cmax_solver.hip"]
HEADERS = ["cmax_common.h", "cmax_image_kernels.h", "cmax_patch_kernels.h", "cmax_flow_dual.h", "cmax_search_kernels.h", "cmax_sort_kernels.h", "cmax_event_kernels.inc", os.path.join("..", "..", "include", "cmax_hip.h")]
# -munsafe-fp-atomics: fp32/fp64 atomicAdd lower to global_atomic_add_f32/_f64 and ds_add_f32
# (hardware atomics) instead of compare-and-swap loops.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
               "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB_PATH
    extra = os.environ.get("CMAX_EXTRA_FLAGS", "").split()
    cmd = [hipcc_path()] + HIPCC_FLAGS + extra + [os.path.join(CSRC, f) for f in SOURCES] + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
