// Shared helpers of libcmax_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <string>

#include "../../include/cmax_hip.h"

namespace cmax {

void set_error(const char *fmt, ...);

#define CMAX_CHECK_HIP(expr)                                                                    \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            ::cmax::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return (int)_e;                                                                     \
        }                                                                                       \
    } while (0)

#define CMAX_CHECK_LAUNCH() CMAX_CHECK_HIP(hipGetLastError())

#define CMAX_REQUIRE(cond, msg)                                            \
    do {                                                                   \
        if (!(cond)) {                                                     \
            ::cmax::set_error("%s:%d bad argument: %s", __FILE__, __LINE__, msg); \
            return CMAX_EINVAL;                                            \
        }                                                                  \
    } while (0)

constexpr int kWave = 64;  // CDNA4 wavefront

static inline int div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Grid for a streaming (grid-stride) kernel: enough workgroups to fill 256 CUs x 8, capped.
static inline int stream_grid(int64_t n, int block) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 256 * 8) g = 256 * 8;
    return (int)g;
}

// ---- wave / block reductions (64-wide) -------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// Block-wide sum of K doubles per thread; result valid in thread 0. smem: K * (blockDim/64) doubles.
template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double *smem) {
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave, nw = blockDim.x / kWave;
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) smem[k * nw + wid] = v[k];
    }
    __syncthreads();
    if (wid == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double x = lane < nw ? smem[k * nw + lane] : 0.0;
            v[k] = wave_sum(x);
        }
    }
    __syncthreads();
}

// Hardware floating-point atomics (global_atomic_add_f32 / _f64, ds_add_f32): never a CAS loop.
__device__ __forceinline__ void atomic_add(float *p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add(double *p, double v) { unsafeAtomicAdd(p, v); }

template <typename T>
__device__ __forceinline__ T floor_t(T v);
template <>
__device__ __forceinline__ float floor_t<float>(float v) { return floorf(v); }
template <>
__device__ __forceinline__ double floor_t<double>(double v) { return floor(v); }

}  // namespace cmax
