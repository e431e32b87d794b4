// Patch grid -> dense flow and its adjoint, shared by the leaf operator (cmax_flow.hip) and the patch-objective
// plan (cmax_solver.hip).
#pragma once
#include "cmax_common.h"

namespace cmax {

// ---------------------------------------------------------------------------------------------
// Patch grid -> dense flow and its adjoint (interpolate_dense_flow_from_patch_tensor,
// src/solver/patch_contrast_base.py:462-506): negate, replicate-pad by (pad_h, pad_w), bilinear x
// (sw_h, sw_w) with align_corners=False, centre-crop to [H, W].
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void bilin_tap(int dst, int scale, int n_in, int &i0, int &i1, T &lam) {
    T src = ((T)dst + (T)0.5) / (T)scale - (T)0.5;
    if (src < (T)0) src = (T)0;
    int a = (int)floor_t<T>(src);
    if (a > n_in - 1) a = n_in - 1;
    i0 = a;
    i1 = a + 1 < n_in ? a + 1 : n_in - 1;
    lam = src - (T)a;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// TOut / scale: the plan writes the fp32 motion of the fused objective (flow * t_scale) straight from the fp64 patch grid
template <typename T, typename TOut = T>
__global__ void __launch_bounds__(256)
k_patch_to_dense(const T *__restrict__ motion, int ph, int pw, int pad_h, int pad_w, int sw_h, int sw_w, int H, int W,
                 TOut *__restrict__ flow, T scale = (T)1) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= 2 * (int64_t)H * W) return;
    const int c = (int)(p / ((int64_t)H * W));
    const int64_t q = p % ((int64_t)H * W);
    const int i = (int)(q / W), j = (int)(q % W);
    const int gh = ph + 2 * pad_h, gw = pw + 2 * pad_w;
    const int h1 = (gh * sw_h) / 2 - H / 2, w1 = (gw * sw_w) / 2 - W / 2;  // lines 501-505
    int r0, r1, c0, c1;
    T lr, lc;
    bilin_tap<T>(i + h1, sw_h, gh, r0, r1, lr);
    bilin_tap<T>(j + w1, sw_w, gw, c0, c1, lc);
    const T *m = motion + (int64_t)c * ph * pw;
    auto PM = [&](int R, int C) { return -m[(int64_t)clampi(R - pad_h, 0, ph - 1) * pw + clampi(C - pad_w, 0, pw - 1)]; };
    const T v = ((T)1 - lr) * (((T)1 - lc) * PM(r0, c0) + lc * PM(r0, c1)) + lr * (((T)1 - lc) * PM(r1, c0) + lc * PM(r1, c1));
    flow[p] = (TOut)(v * scale);
}

// adjoint, gather form: one workgroup per patch cell sums the contributions of every output pixel
// whose taps touch the cell (no atomics: the motion gradient has only 2*ph*pw entries).
template <typename T>
__global__ void __launch_bounds__(256)
k_patch_to_dense_adj(const T *__restrict__ gflow, int ph, int pw, int pad_h, int pad_w, int sw_h, int sw_w, int H, int W,
                     T *__restrict__ gmotion) {
    __shared__ double smem[4];
    const int cell = blockIdx.x;  // c * ph * pw + pr * pw + pc
    const int c = cell / (ph * pw), pr = (cell / pw) % ph, pc = cell % pw;
    const int gh = ph + 2 * pad_h, gw = pw + 2 * pad_w;
    const int h1 = (gh * sw_h) / 2 - H / 2, w1 = (gw * sw_w) / 2 - W / 2;
    // padded rows that map to patch row pr: [Rlo, Rhi]; output rows with a tap on them lie in a band around
    const int Rlo = pr == 0 ? 0 : pr + pad_h, Rhi = pr == ph - 1 ? gh - 1 : pr + pad_h;
    const int Clo = pc == 0 ? 0 : pc + pad_w, Chi = pc == pw - 1 ? gw - 1 : pc + pad_w;
    const int i_lo = max((Rlo - 1) * sw_h - h1 - 1, 0), i_hi = min((Rhi + 2) * sw_h - h1 + 1, H);
    const int j_lo = max((Clo - 1) * sw_w - w1 - 1, 0), j_hi = min((Chi + 2) * sw_w - w1 + 1, W);
    const int bw = j_hi - j_lo, bh = i_hi - i_lo;
    double acc[1] = {0.0};
    for (int t = threadIdx.x; t < bw * bh; t += blockDim.x) {
        const int i = i_lo + t / bw, j = j_lo + t % bw;
        int r0, r1, c0, c1;
        T lr, lc;
        bilin_tap<T>(i + h1, sw_h, gh, r0, r1, lr);
        bilin_tap<T>(j + w1, sw_w, gw, c0, c1, lc);
        const T wr = (clampi(r0 - pad_h, 0, ph - 1) == pr ? (T)1 - lr : (T)0) + (clampi(r1 - pad_h, 0, ph - 1) == pr ? lr : (T)0);
        const T wc = (clampi(c0 - pad_w, 0, pw - 1) == pc ? (T)1 - lc : (T)0) + (clampi(c1 - pad_w, 0, pw - 1) == pc ? lc : (T)0);
        acc[0] += (double)(-gflow[(int64_t)c * H * W + (int64_t)i * W + j] * wr * wc);
    }
    block_sum<1>(acc, smem);
    if (threadIdx.x == 0) gmotion[cell] = (T)acc[0];
}

}  // namespace cmax
