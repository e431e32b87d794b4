#!/bin/bash
# usage: tools/resources.sh [out.txt] -- kernel-resource-usage remarks of csrc/cmax_fused.hip (device-only compile), see tools/kernel_resources.py
out=${1:-/tmp/res.txt}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fno-gpu-rdc -mllvm -amdgpu-kernarg-preload-count=16 --cuda-device-only -c \
  event_based_optical_flow_amd/csrc/cmax_fused.hip -o /tmp/x.o -Rpass-analysis=kernel-resource-usage $CMAX_EXTRA_FLAGS 2> "$out"
