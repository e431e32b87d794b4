"""Builds libcmax_hip.so (gfx950) in-tree with hipcc.  No GPU needed (cross-compiles).

Every translation unit is compiled to its own object (in parallel, re-used while neither it nor a header
changed) and the objects are linked into the shared library.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
# CMAX_LIB: alternative output / load path (tuning experiments build variants side by side)
LIB_PATH = os.environ.get("CMAX_LIB", os.path.join(PKG_DIR, "libcmax_hip.so"))
OBJ_DIR = os.path.join(PKG_DIR, "_obj")  # objects: git-ignored (*.o) and not shipped to the GPU box (.gpurunignore)
SOURCES = ["cmax_leaf.hip", "cmax_flow.hip", "cmax_fused.hip", "cmax_solver.hip", "cmax_comm.hip"]
HEADERS = ["cmax_common.h", "cmax_image_kernels.h", "cmax_patch_kernels.h", "cmax_flow_dual.h", "cmax_search_kernels.h",
           "cmax_sort_kernels.h", "cmax_radix_sort.h", "cmax_event_kernels.inc", "cmax_comm.h", os.path.join("..", "..", "include", "cmax_hip.h")]
# -munsafe-fp-atomics: fp32/fp64 atomicAdd lower to global_atomic_add_f32/_f64 and ds_add_f32
# (hardware atomics) instead of compare-and-swap loops.
# -amdgpu-kernarg-preload-count=16: the first 16 dwords of a kernel's arguments arrive in SGPRs at wave launch (gfx940+) instead
# of through a scalar load from the argument block -- the event kernels put their work list first (cmax_event_kernels.inc).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
               "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-mllvm", "-amdgpu-kernarg-preload-count=16"]
LINK_FLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc"]
LINK_LIBS = ["-ldl"]  # cmax_comm.hip binds RCCL with dlopen (no link-time dependency on librccl)


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _deps():
    return [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in _deps() if os.path.exists(d))


def _object_path(src: str, flags) -> str:
    tag = hashlib.sha1(" ".join(flags).encode()).hexdigest()[:10]
    return os.path.join(OBJ_DIR, f"{os.path.splitext(src)[0]}.{tag}.o")


def _compile(src: str, flags, force: bool, verbose: bool) -> str:
    obj = _object_path(src, flags)
    inputs = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS]
    if not force and os.path.exists(obj) and all(os.path.getmtime(i) <= os.path.getmtime(obj) for i in inputs if os.path.exists(i)):
        return obj
    cmd = [hipcc_path()] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return obj


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    flags = HIPCC_FLAGS + os.environ.get("CMAX_EXTRA_FLAGS", "").split()
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:
        objs = list(pool.map(lambda src: _compile(src, flags, force, verbose), SOURCES))
    cmd = [hipcc_path()] + LINK_FLAGS + objs + LINK_LIBS + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
