// Batched per-patch translation search: the finer-scale re-initialisation of the pyramid solver.
//
// Reference: PyramidalPatchContrastMaximization.initialize_guess_from_optuna_sampling / objective_initial /
// calculate_cost_for_small_patch (src/solver/patch_contrast_pyramid.py:320-428).  Per patch the reference crops the
// events (utils.crop_event, src/utils/event_utils.py:50-70), shifts them to the patch origin, and scores a candidate
// translation (trans_x, trans_y) with NormalizedGradientMagnitude on numpy arrays:
//     warp 2-DoF to the MIDDLE of the patch's own time span (src/warp.py:483-522, 200-234)
//     -> bilinear_vote_numpy into a patch-sized image (src/event_image_converter.py:257-314)
//     -> scipy.ndimage.gaussian_filter(sigma) (event_image_converter.py:122-124)
//     -> mean(gx^2 + gy^2), gx/gy = cv2.Sobel / 8 with OpenCV's default border (reflect-101), no boundary omitted
//        (src/costs/gradient_magnitude.py:78-95)
//     loss = GM(un-warped) / GM(warped)   (src/costs/normalized_gradient_magnitude.py:81-94).
// One trial there is a chain of numpy calls on a few thousand events; here ONE workgroup scores one
// (patch, candidate) pair entirely in LDS and a launch scores every pair of a scale.
//
// The events come from the handle's sorted arrays: the tiles a patch box overlaps are contiguous per tile row, so a
// workgroup walks [tile_start[first tile of the row], tile_start[last + 1]) and tests the box per event.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cmax_common.h"

namespace cmax {

constexpr int kSearchThreads = 256;
constexpr float kSearchFixScale = 262144.f;  // 2^18: <= 8191 unit votes per cell before the signed 32-bit range ends

struct SearchArgs {
    const uint2 *evp;  // packed events: x | y << 12 | bin << 24, tau (fp32, normalised to the batch)
    const float *rx, *ry;  // fractional parts of the source coordinates (read only when has_frac)
    const int *tile_start;  // first sorted event of every group
    int groups_per_tile, ntr, ntc, has_frac;
    int slab_major;  // the groups are ordered (tile row, time slab, tile column) -- cmax_set_time_slabs -- instead of (tile row, tile column, bin)
    const int4 *boxes;  // [n_patch] (x_min, x_max, y_min, y_max): x_min <= x < x_max, rows first
    int img_h, img_w;   // the patch image the reference's imager of this scale allocates
};

// events of tile row `tr` inside the box's tile columns: ONE contiguous range in tile-major order (a tile's time bins follow each
// other), one range per time slab `sl` in slab-major order (round 5: the solver keeps large-motion batches in slab order and
// re-initialises its finer scales from the same handle)
__device__ __forceinline__ int search_row_parts(const SearchArgs &a) { return a.slab_major ? a.groups_per_tile : 1; }
__device__ __forceinline__ void search_row_range(const SearchArgs &a, int tr, int sl, int tc0, int tc1, int &begin, int &end) {
    if (tc1 < tc0) {  // box entirely outside the sensor
        begin = end = 0;
        return;
    }
    if (a.slab_major) {
        const int row = (tr * a.groups_per_tile + sl) * a.ntc;
        begin = a.tile_start[row + tc0];
        end = a.tile_start[row + tc1 + 1];
        return;
    }
    begin = a.tile_start[(tr * a.ntc + tc0) * a.groups_per_tile];
    end = a.tile_start[(tr * a.ntc + tc1 + 1) * a.groups_per_tile];
}

// per patch: (tau_min, tau_max) of the events inside the box and their number
__global__ void __launch_bounds__(kSearchThreads) k_search_range(SearchArgs a, float2 *__restrict__ range, int *__restrict__ count) {
    __shared__ float s_lo[kSearchThreads / kWave], s_hi[kSearchThreads / kWave];
    __shared__ int s_n[kSearchThreads / kWave];
    const int4 box = a.boxes[blockIdx.x];
    float lo = 3.0e38f, hi = -3.0e38f;
    int cnt = 0;
    if (box.y > box.x && box.w > box.z) {
        const int tr0 = max(box.x, 0) >> 4, tr1 = min((box.y - 1) >> 4, a.ntr - 1), tc0 = max(box.z, 0) >> 4, tc1 = min((box.w - 1) >> 4, a.ntc - 1);
        for (int part = 0; part < (tr1 - tr0 + 1) * search_row_parts(a); ++part) {
            int begin, end;
            search_row_range(a, tr0 + part / search_row_parts(a), part % search_row_parts(a), tc0, tc1, begin, end);
            for (int i = begin + (int)threadIdx.x; i < end; i += kSearchThreads) {
                const uint2 e = a.evp[i];
                const int x = (int)(e.x & 0xFFFu), y = (int)((e.x >> 12) & 0xFFFu);
                if (x >= box.x && x < box.y && y >= box.z && y < box.w) {
                    const float tau = __uint_as_float(e.y);
                    lo = fminf(lo, tau);
                    hi = fmaxf(hi, tau);
                    ++cnt;
                }
            }
        }
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o, kWave));
        hi = fmaxf(hi, __shfl_xor(hi, o, kWave));
        cnt += __shfl_xor(cnt, o, kWave);
    }
    const int wid = threadIdx.x / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) {
        s_lo[wid] = lo;
        s_hi[wid] = hi;
        s_n[wid] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kSearchThreads / kWave; ++w) {
            lo = fminf(lo, s_lo[w]);
            hi = fmaxf(hi, s_hi[w]);
            cnt += s_n[w];
        }
        range[blockIdx.x] = make_float2(lo, hi);
        count[blockIdx.x] = cnt;
    }
}

__device__ __forceinline__ int search_refl_dup(int i, int n) {  // scipy 'reflect': d c b a | a b c d | d c b a
    const int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

__device__ __forceinline__ int search_refl_101(int i, int n) {  // OpenCV BORDER_REFLECT_101 for a one-pixel apron
    if (n == 1) return 0;
    if (i < 0) return -i;
    if (i >= n) return 2 * n - 2 - i;
    return i;
}

// grid (n_patch, n_cand + 1): blockIdx.y == n_cand scores the un-warped events (the numerator of the loss).
// cand [n_patch][n_cand] translations in pixel per time unit of the RAW timestamps; period = t_max - t_min of the batch.
// gm [n_patch][n_cand + 1] mean squared Sobel magnitude of the blurred patch image.
// dynamic LDS: 2 * img_h * img_w words.
__global__ void __launch_bounds__(kSearchThreads)
k_patch_search(SearchArgs a, const float2 *__restrict__ range, int n_cand, const float2 *__restrict__ cand, float period, float sigma,
               int radius, float *__restrict__ gm) {
    extern __shared__ float s_dyn[];
    __shared__ double s_red[kSearchThreads / kWave];
    const int np = a.img_h * a.img_w;
    float *img = s_dyn, *tmp = s_dyn + np;
    int *votes = reinterpret_cast<int *>(s_dyn);
    const int patch = blockIdx.x, c = blockIdx.y;
    const int4 box = a.boxes[patch];
    const float2 tr = range[patch];
    float2 th = make_float2(0.f, 0.f);
    if (c < n_cand) th = cand[(int64_t)patch * n_cand + c];
    // displacement = theta * (t - t_mid) = theta * period * (tau - tau_mid)
    const float tau_mid = tr.x + 0.5f * (tr.y - tr.x);
    const float kx = th.x * period, ky = th.y * period;

    for (int i = threadIdx.x; i < np; i += kSearchThreads) votes[i] = 0;
    __syncthreads();
    // a patch whose events share one timestamp: the reference normalises dt by a zero span, every warped
    // coordinate is NaN and no vote lands (src/warp.py:254-258) -- the warped image stays empty
    const bool zero_span = c < n_cand && !(tr.y > tr.x);
    if (box.y > box.x && box.w > box.z && !zero_span) {
        const int tr0 = max(box.x, 0) >> 4, tr1 = min((box.y - 1) >> 4, a.ntr - 1), tc0 = max(box.z, 0) >> 4, tc1 = min((box.w - 1) >> 4, a.ntc - 1);
        for (int part = 0; part < (tr1 - tr0 + 1) * search_row_parts(a); ++part) {
            int begin, end;
            search_row_range(a, tr0 + part / search_row_parts(a), part % search_row_parts(a), tc0, tc1, begin, end);
            for (int i = begin + (int)threadIdx.x; i < end; i += kSearchThreads) {
                const uint2 e = a.evp[i];
                const int x = (int)(e.x & 0xFFFu), y = (int)((e.x >> 12) & 0xFFFu);
                if (!(x >= box.x && x < box.y && y >= box.z && y < box.w)) continue;
                const float dt = __uint_as_float(e.y) - tau_mid;
                float ox = kx * dt, oy = ky * dt;
                if (a.has_frac) {
                    ox += a.rx[i];
                    oy += a.ry[i];
                }
                // integer source pixel + small offset: the floor is taken on the offset alone (full fp32 resolution)
                const float bx = floorf(ox), by = floorf(oy);
                const float fx = ox - bx, fy = oy - by;
                const int r0 = x - box.x + (int)bx, c0 = y - box.z + (int)by;
                const bool r_in0 = r0 >= 0 && r0 < a.img_h, r_in1 = r0 + 1 >= 0 && r0 + 1 < a.img_h;
                const bool c_in0 = c0 >= 0 && c0 < a.img_w, c_in1 = c0 + 1 >= 0 && c0 + 1 < a.img_w;
                const float wx0 = 1.f - fx, wy0 = 1.f - fy;
                if (r_in0 && c_in0) atomicAdd(&votes[r0 * a.img_w + c0], (int)rintf(wx0 * wy0 * kSearchFixScale));
                if (r_in1 && c_in0) atomicAdd(&votes[(r0 + 1) * a.img_w + c0], (int)rintf(fx * wy0 * kSearchFixScale));
                if (r_in0 && c_in1) atomicAdd(&votes[r0 * a.img_w + c0 + 1], (int)rintf(wx0 * fy * kSearchFixScale));
                if (r_in1 && c_in1) atomicAdd(&votes[(r0 + 1) * a.img_w + c0 + 1], (int)rintf(fx * fy * kSearchFixScale));
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < np; i += kSearchThreads) img[i] = (float)votes[i] * (1.f / kSearchFixScale);
    __syncthreads();
    if (radius > 0) {
        // scipy.ndimage.gaussian_filter: taps exp(-t^2 / (2 sigma^2)) / sum over |t| <= radius, axis 0 then axis 1
        const float inv2s2 = 0.5f / (sigma * sigma);
        float norm = 0.f;
        for (int t = -radius; t <= radius; ++t) norm += __expf(-(float)(t * t) * inv2s2);
        const float inv_norm = 1.f / norm;
        for (int i = threadIdx.x; i < np; i += kSearchThreads) {
            const int r = i / a.img_w, q = i - r * a.img_w;
            float acc = 0.f;
            for (int t = -radius; t <= radius; ++t)
                acc += __expf(-(float)(t * t) * inv2s2) * img[search_refl_dup(r + t, a.img_h) * a.img_w + q];
            tmp[i] = acc * inv_norm;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < np; i += kSearchThreads) {
            const int r = i / a.img_w, q = i - r * a.img_w;
            float acc = 0.f;
            for (int t = -radius; t <= radius; ++t)
                acc += __expf(-(float)(t * t) * inv2s2) * tmp[r * a.img_w + search_refl_dup(q + t, a.img_w)];
            img[i] = acc * inv_norm;
        }
        __syncthreads();
    }
    double sum = 0.0;
    for (int i = threadIdx.x; i < np; i += kSearchThreads) {
        const int r = i / a.img_w, q = i - r * a.img_w;
        const int rm = search_refl_101(r - 1, a.img_h) * a.img_w, rz = r * a.img_w, rp = search_refl_101(r + 1, a.img_h) * a.img_w;
        const int qm = search_refl_101(q - 1, a.img_w), qp = search_refl_101(q + 1, a.img_w);
        const float a00 = img[rm + qm], a01 = img[rm + q], a02 = img[rm + qp];
        const float a10 = img[rz + qm], a12 = img[rz + qp];
        const float a20 = img[rp + qm], a21 = img[rp + q], a22 = img[rp + qp];
        // cv2.Sobel dx=1: derivative along columns, smoothing along rows (and the transpose for dy=1), / 8
        const float gc = ((a02 - a00) + 2.f * (a12 - a10) + (a22 - a20)) * 0.125f;
        const float gr = ((a20 - a00) + 2.f * (a21 - a01) + (a22 - a02)) * 0.125f;
        sum += (double)(gc * gc + gr * gr);
    }
    double v[1] = {sum};
    block_sum<1>(v, s_red);
    if (threadIdx.x == 0) gm[(int64_t)patch * (n_cand + 1) + c] = (float)(v[0] / (double)np);
}

}  // namespace cmax
