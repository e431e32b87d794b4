// Image-space kernels shared by the leaf operators and the fused objective (header-only
// templates): 3-tap reflect-101 Gaussian blur + transpose, Sobel/8 with zero padding.
#pragma once
#include "cmax_common.h"

namespace cmax {

__device__ __forceinline__ int refl101(int i, int n) {
    if (n == 1) return 0;
    if (i < 0) return -i;
    if (i >= n) return 2 * n - 2 - i;
    return i;
}

// torchvision gaussian_blur(kernel_size=3, sigma) taps (src/event_image_converter.py:158)
static inline void blur_taps(double sigma, double &k0, double &k1) {
    double e = exp(-0.5 / (sigma * sigma));
    double s = 1.0 + 2.0 * e;
    k0 = 1.0 / s;
    k1 = e / s;
}

// forward: out[i,j] = sum_{a,b} k[a] k[b] in[refl(i+a), refl(j+b)]
// bs: element stride between the images of a batch (blockIdx.y), 0 for a single image
template <typename T>
__global__ void __launch_bounds__(256) k_blur3(const T *__restrict__ in, int H, int W, T k0, T k1, T *__restrict__ out, int64_t bs = 0) {
    in += blockIdx.y * bs;
    out += blockIdx.y * bs;
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (int64_t)H * W) return;
    int i = (int)(p / W), j = (int)(p % W);
    int im = refl101(i - 1, H), ip = refl101(i + 1, H), jm = refl101(j - 1, W), jp = refl101(j + 1, W);
    auto row = [&](int r) { return k1 * in[(int64_t)r * W + jm] + k0 * in[(int64_t)r * W + j] + k1 * in[(int64_t)r * W + jp]; };
    out[p] = k1 * row(im) + k0 * row(i) + k1 * row(ip);
}

// 1-D transposed operator: (K^T g)[p] = k0 g[p] + k1 (g[p-1] + g[p+1]) (in range) + the reflected
// taps folded back: index 1 also receives k1 g[0], index n-2 also receives k1 g[n-1].
template <typename T, typename F>
__device__ __forceinline__ T blur_adj_1d(int p, int n, T k0, T k1, F g) {
    if (n == 1) return (k0 + (T)2 * k1) * g(0);
    T s = k0 * g(p);
    if (p - 1 >= 0) s += k1 * g(p - 1);
    if (p + 1 < n) s += k1 * g(p + 1);
    if (p == 1) s += k1 * g(0);
    if (p == n - 2) s += k1 * g(n - 1);
    return s;
}

template <typename T>
__global__ void __launch_bounds__(256) k_blur3_adj(const T *__restrict__ g, int H, int W, T k0, T k1, T *__restrict__ out, int64_t bs = 0) {
    g += blockIdx.y * bs;
    out += blockIdx.y * bs;
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (int64_t)H * W) return;
    int i = (int)(p / W), j = (int)(p % W);
    out[p] = blur_adj_1d<T>(i, H, k0, k1, [&](int r) {
        return blur_adj_1d<T>(j, W, k0, k1, [&](int c) { return g[(int64_t)r * W + c]; });
    });
}

// Sobel / 8 with zero padding (SobelTorch, src/utils/stat_utils.py:50-62,80-83); fp64 arithmetic
// like the solvers' precision="64" (src/solver/base.py:180, gradient_magnitude.py:65-66).
template <typename T>
__device__ __forceinline__ void sobel8(const T *__restrict__ img, int H, int W, int i, int j, double &gx, double &gy) {
    auto at = [&](int r, int c) -> double {
        return (r < 0 || r >= H || c < 0 || c >= W) ? 0.0 : (double)img[(int64_t)r * W + c];
    };
    double a00 = at(i - 1, j - 1), a01 = at(i - 1, j), a02 = at(i - 1, j + 1);
    double a10 = at(i, j - 1), a12 = at(i, j + 1);
    double a20 = at(i + 1, j - 1), a21 = at(i + 1, j), a22 = at(i + 1, j + 1);
    gx = ((a20 + 2.0 * a21 + a22) - (a00 + 2.0 * a01 + a02)) / 8.0;  // row derivative
    gy = ((a02 + 2.0 * a12 + a22) - (a00 + 2.0 * a10 + a20)) / 8.0;  // column derivative
}

// d mean(gx^2+gy^2) / d img[p] * n / 2 * 8 : sum over the output pixels q in Omega that read p.
template <typename T>
__device__ __forceinline__ double sobel8_adj(const T *__restrict__ img, int H, int W, int i0, int i, int j) {
    const double SX[3][3] = {{-1, -2, -1}, {0, 0, 0}, {1, 2, 1}};
    const double SY[3][3] = {{-1, 0, 1}, {-2, 0, 2}, {-1, 0, 1}};
    double s = 0.0;
#pragma unroll
    for (int a = -1; a <= 1; ++a)
#pragma unroll
        for (int b = -1; b <= 1; ++b) {
            int qi = i - a, qj = j - b;  // output pixel q read p with tap (a, b)
            if (qi < i0 || qi >= H - i0 || qj < i0 || qj >= W - i0) continue;
            double gx, gy;
            sobel8<T>(img, H, W, qi, qj, gx, gy);
            s += gx * SX[a + 1][b + 1] + gy * SY[a + 1][b + 1];
        }
    return s;
}

// fp32 variants for the fused path (the IWE is fp32; sums of squares are accumulated in fp64 by the caller)
__device__ __forceinline__ void sobel8_f32(const float *__restrict__ img, int H, int W, int i, int j, float &gx, float &gy) {
    auto at = [&](int r, int c) -> float {
        return (r < 0 || r >= H || c < 0 || c >= W) ? 0.f : img[(int64_t)r * W + c];
    };
    const float a00 = at(i - 1, j - 1), a01 = at(i - 1, j), a02 = at(i - 1, j + 1);
    const float a10 = at(i, j - 1), a12 = at(i, j + 1);
    const float a20 = at(i + 1, j - 1), a21 = at(i + 1, j), a22 = at(i + 1, j + 1);
    gx = ((a20 + 2.f * a21 + a22) - (a00 + 2.f * a01 + a02)) * 0.125f;
    gy = ((a02 + 2.f * a12 + a22) - (a00 + 2.f * a10 + a20)) * 0.125f;
}

// sum_{q in Omega reading p} gx(q) SX + gy(q) SY from one 5x5 neighbourhood held in registers
// (25 loads per pixel instead of 9 x 8); gx, gy include the /8.
// gx0, gy0 (optional): the Sobel/8 response AT (i, j) from the same registers
__device__ __forceinline__ float sobel8_adj_f32(const float *__restrict__ img, int H, int W, int i0, int i, int j,
                                                float *gx0 = nullptr, float *gy0 = nullptr) {
    float v[5][5];
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            const int r = i + a - 2, c = j + b - 2;
            v[a][b] = (r < 0 || r >= H || c < 0 || c >= W) ? 0.f : img[(int64_t)r * W + c];
        }
    if (gx0) {
        *gx0 = ((v[3][1] + 2.f * v[3][2] + v[3][3]) - (v[1][1] + 2.f * v[1][2] + v[1][3])) * 0.125f;
        *gy0 = ((v[1][3] + 2.f * v[2][3] + v[3][3]) - (v[1][1] + 2.f * v[2][1] + v[3][1])) * 0.125f;
    }
    float s = 0.f;
#pragma unroll
    for (int a = -1; a <= 1; ++a)
#pragma unroll
        for (int b = -1; b <= 1; ++b) {
            const int qi = i - a, qj = j - b;  // output pixel q read p with tap (a, b)
            if (qi < i0 || qi >= H - i0 || qj < i0 || qj >= W - i0) continue;
            const int ci = 2 - a, cj = 2 - b;  // q inside the 5x5 block
            const float gx = ((v[ci + 1][cj - 1] + 2.f * v[ci + 1][cj] + v[ci + 1][cj + 1]) -
                              (v[ci - 1][cj - 1] + 2.f * v[ci - 1][cj] + v[ci - 1][cj + 1])) * 0.125f;
            const float gy = ((v[ci - 1][cj + 1] + 2.f * v[ci][cj + 1] + v[ci + 1][cj + 1]) -
                              (v[ci - 1][cj - 1] + 2.f * v[ci][cj - 1] + v[ci + 1][cj - 1])) * 0.125f;
            const float sx = (float)a * (b == 0 ? 2.f : 1.f);  // SX[a+1][b+1] = a * (2 - |b|)
            const float sy = (float)b * (a == 0 ? 2.f : 1.f);  // SY[a+1][b+1] = b * (2 - |a|)
            s += gx * sx + gy * sy;
        }
    return s;
}

}  // namespace cmax
