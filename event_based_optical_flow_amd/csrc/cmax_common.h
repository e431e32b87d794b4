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

// ---- library-internal view of the host-side state of a handle that changes from one evaluation to the next
// (not exported: cmax_solver.hip replays captured hipGraphs and must advance this state the way the eager
// launch sequence would have)
struct HandleEvalState {
    int cur_buf;
    unsigned zero_mask[2];
    int orig_valid, orig_cost, orig_omit;
    double orig_sigma;
    const float *last_iwe[4];
    uint64_t generation;  // bumped whenever the packed events / work list (and with them device pointers) change
    int profiling;
    int deterministic;  // (read only: part of the key of a captured launch sequence)
    int mu_buf;         // which of K1's two vote-sum buffers the next evaluation adds into (the blurred variance clears the other one)
};
// key of a captured launch sequence: everything of the state that decides which buffers / kernels an evaluation uses
__attribute__((visibility("hidden"))) uint64_t handle_state_key(const HandleEvalState &st, int kind);
__attribute__((visibility("hidden"))) void handle_get_eval_state(cmax_handle_t h, HandleEvalState *out);
__attribute__((visibility("hidden"))) void handle_set_eval_state(cmax_handle_t h, const HandleEvalState *in);
// cmax_flow.hip, for the patch plan: fp64 voxel and (when the single-launch tiled chain ran: *wrote_v32) its fp32 copy
__attribute__((visibility("hidden"))) int voxel_construct_f64_f32(const double *F, int Tn, int t0, int H, int W, int scheme, double *V,
                                                               float *V32, bool *wrote_v32, hipStream_t s);
// ... and the adjoint sweeps (gV / dgV in, gradient in bin t0 out); det: order-free step kernels (deterministic handles)
__attribute__((visibility("hidden"))) int voxel_construct_adj_f64(const double *V, int Tn, int t0, int H, int W, int scheme, double *gV, hipStream_t s,
                                                               bool det);
__attribute__((visibility("hidden"))) int voxel_construct_adj_tan_f64(const double *V, const double *dV, int Tn, int t0, int H, int W, int scheme,
                                                                   double *gV, double *dgV, hipStream_t s, bool det);
__attribute__((visibility("hidden"))) bool handle_is_deterministic(cmax_handle_t h);
// time-sliced batches (cmax_fused.hip, for the patch plan): does the handle hold a communicator; cmax_objective_dist / the product
// with the motion gradient left as this rank's share; in-place sum over the ranks on the handle's communicator (a no-op without one)
__attribute__((visibility("hidden"))) bool handle_has_comm(cmax_handle_t h);
__attribute__((visibility("hidden"))) int objective_dist_local_grad(cmax_handle_t h, const cmax_objective_t *d, const float *motion, double *result,
                                                                 void *grad, hipStream_t s);
__attribute__((visibility("hidden"))) int objective_hvp_dist_local(cmax_handle_t h, const cmax_objective_t *d, const float *motion,
                                                                const float *tangent, void *hv, hipStream_t s);
__attribute__((visibility("hidden"))) int handle_allreduce_sum(cmax_handle_t h, void *buf, size_t count, bool f64, hipStream_t s);
__attribute__((visibility("hidden"))) int handle_allreduce_sum_with_scalars(cmax_handle_t h, void *buf, size_t count, bool f64, double *scalars, int n_scalars, hipStream_t s);

// ---- wave / block reductions (64-wide) -------------------------------------------------------
// DPP on the VALU instead of ds_bpermute shuffles (~16 cycles each on gfx950): 4 row shifts, then lane 15 of
// each row into the next row and lane 31 into rows 2-3; the sum of the wave ends up in LANE 63 only.
template <int CTRL>
__device__ __forceinline__ double dpp_shift(double v) {  // lanes without a source read 0
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_lane63(double v) {
    v += dpp_shift<0x111>(v);  // row_shr:1
    v += dpp_shift<0x112>(v);  // row_shr:2
    v += dpp_shift<0x114>(v);  // row_shr:4
    v += dpp_shift<0x118>(v);  // row_shr:8
    v += dpp_shift<0x142>(v);  // row_bcast:15
    v += dpp_shift<0x143>(v);  // row_bcast:31
    return v;
}

// Block-wide sum of K doubles per thread; result valid in thread 0. smem: K * (blockDim/64) doubles.
template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double *smem) {
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave, nw = blockDim.x / kWave;
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = wave_sum_lane63(v[k]);
    if (lane == kWave - 1) {
#pragma unroll
        for (int k = 0; k < K; ++k) smem[k * nw + wid] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // <= 16 waves: a serial sum beats a second cross-lane phase
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double x = 0.0;
            for (int w = 0; w < nw; ++w) x += smem[k * nw + w];
            v[k] = x;
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
