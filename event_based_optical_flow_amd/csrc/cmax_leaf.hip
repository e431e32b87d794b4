// Leaf operators of libcmax_hip.so: one HIP kernel (family) per reference leaf function.
// dtype-generic (fp32 / fp64) because the reference's outputs follow the events' dtype
// (src/event_image_converter.py:338); the fused hot path lives in cmax_fused.hip.
#include <stdarg.h>

#include "cmax_common.h"
#include "cmax_image_kernels.h"

namespace cmax {

static thread_local std::string g_last_error;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

// ---------------------------------------------------------------------------------------------
// t_min / t_max  (Warp.calculate_reftime, src/warp.py:216-224)
// doubles are mapped to order-preserving uint64 keys so hardware u64 atomics can be used.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long f64_key(double v) {
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double f64_unkey(unsigned long long k) {
    unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

__global__ void k_tmm_init(unsigned long long *keys) {
    keys[0] = ~0ull;  // running min
    keys[1] = 0ull;   // running max
}

template <typename T>
__global__ void __launch_bounds__(256) k_tmm_reduce(const T *ev, int64_t n, unsigned long long *keys) {
    double lo = INFINITY, hi = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double t = (double)ev[4 * i + 2];
        lo = fmin(lo, t);
        hi = fmax(hi, t);
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        lo = fmin(lo, __shfl_xor(lo, o, kWave));
        hi = fmax(hi, __shfl_xor(hi, o, kWave));
    }
    // one pair of atomics per WORKGROUP: same-address atomics serialise at ~12 ns each (two per wave of a
    // 4096-workgroup grid: 190 us for 1M events)
    __shared__ double s_lo[4], s_hi[4];
    if ((threadIdx.x & (kWave - 1)) == 0) {
        s_lo[threadIdx.x / kWave] = lo;
        s_hi[threadIdx.x / kWave] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = fmin(fmin(s_lo[0], s_lo[1]), fmin(s_lo[2], s_lo[3]));
        hi = fmax(fmax(s_hi[0], s_hi[1]), fmax(s_hi[2], s_hi[3]));
        if (lo <= hi) {
            atomicMin(&keys[0], f64_key(lo));
            atomicMax(&keys[1], f64_key(hi));
        }
    }
}

__global__ void k_tmm_final(unsigned long long *keys) {
    double lo = f64_unkey(keys[0]), hi = f64_unkey(keys[1]);
    reinterpret_cast<double *>(keys)[0] = lo;
    reinterpret_cast<double *>(keys)[1] = hi;
}

template <typename T>
static int tminmax_impl(const void *events, int64_t n, double *tminmax, hipStream_t s) {
    auto *keys = reinterpret_cast<unsigned long long *>(tminmax);
    hipLaunchKernelGGL(k_tmm_init, dim3(1), dim3(1), 0, s, keys);
    if (n > 0) hipLaunchKernelGGL(k_tmm_reduce<T>, dim3(std::min(stream_grid(n, 256), 512)), dim3(256), 0, s, (const T *)events, n, keys);
    hipLaunchKernelGGL(k_tmm_final, dim3(1), dim3(1), 0, s, keys);
    CMAX_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Warp.warp_event  (src/warp.py:156-199; a4 483-522, a5 263-313, a6 315-396)
// ---------------------------------------------------------------------------------------------
template <typename T>
struct RefTime {
    T tref, period;       // dt = (t - tref) / period   (period = 1 when not normalising)
    double dtmin, dtmax;  // extremes of dt (voxel bin edges, src/warp.py:342-345)
};

template <typename T>
__device__ __forceinline__ RefTime<T> make_reftime(const double *tmm, int ref_mode, double frac, int normalize) {
    RefTime<T> r;
    T tmin = (T)tmm[0], tmax = (T)tmm[1];
    if (ref_mode == CMAX_REF_FIRST) r.tref = tmin;                 // warp.py:219-220
    else if (ref_mode == CMAX_REF_LAST) r.tref = tmax;             // warp.py:223-224
    else r.tref = tmin + (tmax - tmin) * (T)frac;                  // warp.py:216-218
    T dlo = tmin - r.tref, dhi = tmax - r.tref;
    r.period = normalize ? (dhi - dlo) : (T)1;                     // warp.py:254-259
    r.dtmin = (double)(normalize ? dlo / r.period : dlo);
    r.dtmax = (double)(normalize ? dhi / r.period : dhi);
    return r;
}

template <typename T>
__global__ void __launch_bounds__(256)
k_warp(const T *__restrict__ ev, int64_t n, int model, const T *__restrict__ motion, int Tn, int H, int W,
       const double *__restrict__ tmm, int ref_mode, double frac, int normalize, T *__restrict__ out,
       T *__restrict__ dt_out, int32_t *__restrict__ bin_out) {
    const RefTime<T> rt = make_reftime<T>(tmm, ref_mode, frac, normalize);
    const int64_t hw = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        T x = ev[4 * i + 0], y = ev[4 * i + 1], t = ev[4 * i + 2], p = ev[4 * i + 3];
        T dt = t - rt.tref;
        if (normalize) dt = dt / rt.period;
        T xo = x, yo = y;
        int bin = -1;
        if (model == CMAX_MODEL_2DOF) {
            xo = x + dt * motion[0];  // warp.py:506-515 (plus sign)
            yo = y + dt * motion[1];
        } else {
            const T *f = motion;
            bool hit = true;
            if (model == CMAX_MODEL_VOXEL) {
                hit = false;
                for (int k = 0; k < Tn; ++k) {  // warp.py:344-361
                    double e0 = ((double)k / (double)Tn) * (rt.dtmax - rt.dtmin) + rt.dtmin;
                    double e1 = (k + 1 < Tn) ? ((double)(k + 1) / (double)Tn) * (rt.dtmax - rt.dtmin) + rt.dtmin : rt.dtmax + 1e3;
                    if ((T)e0 <= dt && dt < (T)e1) {
                        bin = k;
                        hit = true;
                    }
                }
                if (hit) f = motion + (int64_t)bin * 2 * hw;
            }
            if (hit) {
                int64_t ind = (int64_t)x * W + (int64_t)y;  // trunc like .long(), warp.py:304
                xo = x - dt * f[ind];                        // minus sign, warp.py:305-306
                yo = y - dt * f[hw + ind];
            }
        }
        out[4 * i + 0] = xo;
        out[4 * i + 1] = yo;
        out[4 * i + 2] = dt;
        out[4 * i + 3] = p;
        if (dt_out) dt_out[i] = dt;
        if (bin_out) bin_out[i] = bin;
    }
}

// adjoint w.r.t. the motion
template <typename T>
__global__ void __launch_bounds__(256)
k_warp_bwd_2dof(int64_t n, const T *__restrict__ dt, const T *__restrict__ gw, double *__restrict__ acc) {
    __shared__ double smem[2 * 4];
    double v[2] = {0.0, 0.0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double d = (double)dt[i];
        v[0] += d * (double)gw[4 * i + 0];
        v[1] += d * (double)gw[4 * i + 1];
    }
    block_sum<2>(v, smem);
    if (threadIdx.x == 0) {
        atomic_add(&acc[0], v[0]);
        atomic_add(&acc[1], v[1]);
    }
}

template <typename T>
__global__ void k_cast2(const double *acc, T *out) {
    out[0] = (T)acc[0];
    out[1] = (T)acc[1];
}

template <typename T>
__global__ void __launch_bounds__(256)
k_warp_bwd_flow(const T *__restrict__ ev, int64_t n, int H, int W, const T *__restrict__ dt,
                const int32_t *__restrict__ bin, const T *__restrict__ gw, T *__restrict__ gflow) {
    const int64_t hw = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int b = bin ? bin[i] : 0;
        if (b < 0) continue;
        int64_t ind = (int64_t)ev[4 * i + 0] * W + (int64_t)ev[4 * i + 1];
        T *g = gflow + (int64_t)b * 2 * hw;
        T d = dt[i];
        atomic_add(&g[ind], -d * gw[4 * i + 0]);
        atomic_add(&g[hw + ind], -d * gw[4 * i + 1]);
    }
}

// ---------------------------------------------------------------------------------------------
// bilinear vote / count  (src/event_image_converter.py:316-374, 209-255) and its adjoint
// ---------------------------------------------------------------------------------------------
template <typename T>
struct Corners {
    int64_t i00, i10, i01, i11;  // linear indices (row y1, col x1), (y1+1, x1), (y1, x1+1), (y1+1, x1+1)
    bool m00, m10, m01, m11;     // in-bounds masks (355-363)
    T a, b;                      // row / column fractions (341)
};

template <typename T>
__device__ __forceinline__ Corners<T> corners(T x, T y, int Hp, int Wp, int ph, int pw, T eps) {
    Corners<T> c;
    T fx = floor_t<T>(x + eps), fy = floor_t<T>(y + eps);  // 340
    c.a = x - fx;
    c.b = y - fy;
    // clamp before the int conversion so wild coordinates cannot overflow (they are masked anyway)
    fx = fx < (T)-4e6 ? (T)-4e6 : (fx > (T)4e6 ? (T)4e6 : fx);
    fy = fy < (T)-4e6 ? (T)-4e6 : (fy > (T)4e6 ? (T)4e6 : fy);
    int64_t y1 = (int64_t)fx + ph, x1 = (int64_t)fy + pw;  // 344-345
    bool xin0 = (0 <= x1) && (x1 < Wp), xin1 = (0 <= x1 + 1) && (x1 + 1 < Wp);
    bool yin0 = (0 <= y1) && (y1 < Hp), yin1 = (0 <= y1 + 1) && (y1 + 1 < Hp);
    c.m00 = xin0 && yin0;
    c.m10 = xin0 && yin1;
    c.m01 = xin1 && yin0;
    c.m11 = xin1 && yin1;
    c.i00 = y1 * Wp + x1;
    c.i10 = c.i00 + Wp;
    c.i01 = c.i00 + 1;
    c.i11 = c.i00 + Wp + 1;
    return c;
}

template <typename T>
__global__ void __launch_bounds__(256)
k_vote(const T *__restrict__ xy, int64_t stride, int64_t n, const T *__restrict__ weight, T wscalar, int Hp,
       int Wp, int ph, int pw, T eps, int count, T *__restrict__ img) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        T x = xy[i * stride], y = xy[i * stride + 1];
        if (!(x == x) || !(y == y)) continue;  // NaN coordinates vote nowhere
        Corners<T> c = corners<T>(x, y, Hp, Wp, ph, pw, eps);
        T w = weight ? weight[i] : wscalar;
        T v00, v10, v01, v11;
        if (count) {
            v00 = v10 = v01 = v11 = (T)1;
        } else {
            v00 = ((T)1 - c.a) * ((T)1 - c.b) * w;  // 365-368
            v10 = c.a * ((T)1 - c.b) * w;
            v01 = ((T)1 - c.a) * c.b * w;
            v11 = c.a * c.b * w;
        }
        if (c.m00) atomic_add(&img[c.i00], v00);
        if (c.m10) atomic_add(&img[c.i10], v10);
        if (c.m01) atomic_add(&img[c.i01], v01);
        if (c.m11) atomic_add(&img[c.i11], v11);
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
k_vote_bwd(const T *__restrict__ xy, int64_t stride, int64_t n, const T *__restrict__ weight, T wscalar,
           int Hp, int Wp, int ph, int pw, T eps, const T *__restrict__ G, T *__restrict__ gxy,
           T *__restrict__ gw) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        T x = xy[i * stride], y = xy[i * stride + 1];
        T gx = 0, gy = 0, gwv = 0;
        if ((x == x) && (y == y)) {
            Corners<T> c = corners<T>(x, y, Hp, Wp, ph, pw, eps);
            T w = weight ? weight[i] : wscalar;
            T g00 = c.m00 ? G[c.i00] : (T)0, g10 = c.m10 ? G[c.i10] : (T)0;
            T g01 = c.m01 ? G[c.i01] : (T)0, g11 = c.m11 ? G[c.i11] : (T)0;
            gx = w * (((T)1 - c.b) * (g10 - g00) + c.b * (g11 - g01));
            gy = w * (((T)1 - c.a) * (g01 - g00) + c.a * (g11 - g10));
            gwv = ((T)1 - c.a) * ((T)1 - c.b) * g00 + c.a * ((T)1 - c.b) * g10 + ((T)1 - c.a) * c.b * g01 + c.a * c.b * g11;
        }
        gxy[2 * i + 0] = gx;
        gxy[2 * i + 1] = gy;
        if (gw) gw[i] = gwv;
    }
}

// ---------------------------------------------------------------------------------------------
// 3-tap Gaussian blur, reflect-101  (src/event_image_converter.py:153-159) and its transpose
// ---------------------------------------------------------------------------------------------
// (kernels k_blur3 / k_blur3_adj live in cmax_image_kernels.h, shared with the fused path)

// ---------------------------------------------------------------------------------------------
// contrast functions
// ---------------------------------------------------------------------------------------------
// variance: acc[1] += sum x, acc[2] += sum x^2 over the (cropped) region   (image_variance.py:38-55)
template <typename T>
__global__ void __launch_bounds__(256) k_var_sums(const T *__restrict__ img, int H, int W, int omit, double *acc) {
    __shared__ double smem[2 * 4];
    const int i0 = omit ? 1 : 0, h = H - 2 * i0, w = W - 2 * i0;
    const int64_t n = (int64_t)(h > 0 ? h : 0) * (w > 0 ? w : 0);
    double v[2] = {0.0, 0.0};
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
        int i = (int)(q / w) + i0, j = (int)(q % w) + i0;
        double x = (double)img[(int64_t)i * W + j];
        v[0] += x;
        v[1] += x * x;
    }
    block_sum<2>(v, smem);
    if (threadIdx.x == 0) {
        atomic_add(&acc[1], v[0]);
        atomic_add(&acc[2], v[1]);
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
k_var_final(const T *__restrict__ img, int H, int W, int omit, int ddof, double *acc, T *__restrict__ G,
            const double *__restrict__ gscale) {
    const int i0 = omit ? 1 : 0, h = H - 2 * i0, w = W - 2 * i0;
    const double n = (double)(h > 0 ? h : 0) * (double)(w > 0 ? w : 0);
    const double mu = acc[1] / n;
    const double var = (acc[2] - acc[1] * mu) / (n - ddof);
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p == 0) acc[0] = var;
    if (!G || p >= (int64_t)H * W) return;
    int i = (int)(p / W), j = (int)(p % W);
    bool in = (i >= i0) && (i < H - i0) && (j >= i0) && (j < W - i0);
    double gs = gscale ? *gscale : 1.0;
    G[p] = in ? (T)(gs * 2.0 * ((double)img[p] - mu) / (n - ddof)) : (T)0;
}

template <typename T>
__global__ void __launch_bounds__(256) k_gm_sums(const T *__restrict__ img, int H, int W, int omit, double *acc) {
    __shared__ double smem[1 * 4];
    const int i0 = omit ? 1 : 0, h = H - 2 * i0, w = W - 2 * i0;
    const int64_t n = (int64_t)(h > 0 ? h : 0) * (w > 0 ? w : 0);
    double v[1] = {0.0};
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
        int i = (int)(q / w) + i0, j = (int)(q % w) + i0;
        double gx, gy;
        sobel8<T>(img, H, W, i, j, gx, gy);
        v[0] += gx * gx + gy * gy;  // gradient_magnitude.py:73
    }
    block_sum<1>(v, smem);
    if (threadIdx.x == 0) atomic_add(&acc[1], v[0]);
}

// G[p] = (2/n)/8 * sum over q in Omega, q = p - (a,b):  gx(q) SX[a][b] + gy(q) SY[a][b]
template <typename T>
__global__ void __launch_bounds__(256)
k_gm_final(const T *__restrict__ img, int H, int W, int omit, double *acc, T *__restrict__ G,
           const double *__restrict__ gscale) {
    const int i0 = omit ? 1 : 0, h = H - 2 * i0, w = W - 2 * i0;
    const double n = (double)(h > 0 ? h : 0) * (double)(w > 0 ? w : 0);
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p == 0) acc[0] = acc[1] / n;
    if (!G || p >= (int64_t)H * W) return;
    int i = (int)(p / W), j = (int)(p % W);
    double s = sobel8_adj<T>(img, H, W, i0, i, j);
    double gs = gscale ? *gscale : 1.0;
    G[p] = (T)(gs * (2.0 / n) * s / 8.0);
}

// total variation of a [2,h,w] flow  (src/costs/total_variation.py:60-75,110-126)
template <typename T>
__global__ void __launch_bounds__(256) k_tv_sums(const T *__restrict__ flow, int h, int w, int crop, double *acc) {
    __shared__ double smem[1 * 4];
    const int i0 = crop ? 1 : 0, hh = h - 2 * i0, ww = w - 2 * i0;
    const int64_t n = (int64_t)hh * ww;
    double v[1] = {0.0};
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < 2 * n; q += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)(q / n);
        int64_t r = q % n;
        int i = (int)(r / ww) + i0, j = (int)(r % ww) + i0;
        double sx, sy;
        sobel8<T>(flow + (int64_t)c * h * w, h, w, i, j, sx, sy);
        v[0] += fabs(sx) + fabs(sy);
    }
    block_sum<1>(v, smem);
    if (threadIdx.x == 0) atomic_add(&acc[1], v[0]);
}

template <typename T>
__global__ void __launch_bounds__(256)
k_tv_final(const T *__restrict__ flow, int h, int w, int crop, double *acc, T *__restrict__ G,
           const double *__restrict__ gscale) {
    const int i0 = crop ? 1 : 0, hh = h - 2 * i0, ww = w - 2 * i0;
    const double n = 4.0 * (double)hh * (double)ww;
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p == 0) acc[0] = acc[1] / n;
    if (!G || p >= 2 * (int64_t)h * w) return;
    int c = (int)(p / ((int64_t)h * w));
    int64_t r = p % ((int64_t)h * w);
    int i = (int)(r / w), j = (int)(r % w);
    const T *f = flow + (int64_t)c * h * w;
    const double SX[3][3] = {{-1, -2, -1}, {0, 0, 0}, {1, 2, 1}};
    const double SY[3][3] = {{-1, 0, 1}, {-2, 0, 2}, {-1, 0, 1}};
    double s = 0.0;
    for (int a = -1; a <= 1; ++a)
        for (int b = -1; b <= 1; ++b) {
            int qi = i - a, qj = j - b;
            if (qi < i0 || qi >= h - i0 || qj < i0 || qj >= w - i0) continue;
            double sx, sy;
            sobel8<T>(f, h, w, qi, qj, sx, sy);
            double gsx = (double)((sx > 0) - (sx < 0)), gsy = (double)((sy > 0) - (sy < 0));
            s += gsx * SX[a + 1][b + 1] + gsy * SY[a + 1][b + 1];
        }
    double gs = gscale ? *gscale : 1.0;
    G[p] = (T)(gs * s / 8.0 / n);
}

// ---------------------------------------------------------------------------------------------
// dispatch helpers
// ---------------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------------
// numpy-branch blur: scipy.ndimage.gaussian_filter(image, sigma) (src/event_image_converter.py:122-124)
// radius = int(4 sigma + 0.5), weights exp(-t^2 / (2 sigma^2)) normalised, 'reflect' (edge-duplicating)
// boundary, one 1-D pass per axis (axis 0 first).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int refl_dup(int i, int n) {
    while (i < 0 || i >= n) {
        if (i < 0) i = -i - 1;
        if (i >= n) i = 2 * n - 1 - i;
    }
    return i;
}

template <typename T, int AXIS>
__global__ void __launch_bounds__(256) k_gauss1d(const T *__restrict__ in, int H, int W, T inv2s2, int r, T *__restrict__ out) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (int64_t)H * W) return;
    const int i = (int)(p / W), j = (int)(p % W);
    T acc = 0, norm = 0;
    for (int t = -r; t <= r; ++t) {
        const T w = (T)exp(-(double)((T)(t * t) * inv2s2));
        norm += w;
        const int q = AXIS == 0 ? refl_dup(i + t, H) * W + j : i * W + refl_dup(j + t, W);
        acc += w * in[q];
    }
    out[p] = acc / norm;
}

}  // namespace cmax

using namespace cmax;

#define DISPATCH_DTYPE(dtype, CALL)                         \
    do {                                                    \
        if ((dtype) == CMAX_F32) { using T = float; CALL; } \
        else if ((dtype) == CMAX_F64) { using T = double; CALL; } \
        else { set_error("unknown dtype %d", (int)(dtype)); return CMAX_EINVAL; } \
    } while (0)

extern "C" {

const char *cmax_last_error(void) { return g_last_error.c_str(); }
int cmax_abi_version(void) { return CMAX_ABI_VERSION; }

int cmax_tminmax(const void *events, int dtype, int64_t n, double *tminmax, cmax_stream_t stream) {
    CMAX_REQUIRE(tminmax != nullptr && n >= 0 && (events != nullptr || n == 0), "tminmax");
    DISPATCH_DTYPE(dtype, return tminmax_impl<T>(events, n, tminmax, (hipStream_t)stream));
}

int cmax_warp_events(const void *events, int dtype, int64_t n, int model, const void *motion, int Tn, int H, int W,
                     const double *tminmax, int ref_mode, double ref_frac, int normalize_t, void *warped,
                     void *dt_out, int32_t *bin_out, cmax_stream_t stream) {
    CMAX_REQUIRE(n >= 0 && (n == 0 || (events && warped)) && motion && tminmax, "warp_events: null pointer");
    CMAX_REQUIRE(model == CMAX_MODEL_2DOF || model == CMAX_MODEL_DENSE || model == CMAX_MODEL_VOXEL, "warp_events: model");
    CMAX_REQUIRE(ref_mode >= CMAX_REF_FIRST && ref_mode <= CMAX_REF_FRAC, "warp_events: ref_mode");
    CMAX_REQUIRE(model == CMAX_MODEL_2DOF || (H > 0 && W > 0), "warp_events: image size");
    CMAX_REQUIRE(model != CMAX_MODEL_VOXEL || Tn > 0, "warp_events: T");
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_warp<T>, dim3(stream_grid(n, 256)), dim3(256), 0, s, (const T *)events, n, model,
                                             (const T *)motion, Tn, H, W, tminmax, ref_mode, ref_frac, normalize_t,
                                             (T *)warped, (T *)dt_out, bin_out));
    CMAX_CHECK_LAUNCH();
    return 0;
}

int cmax_warp_events_bwd(const void *events, int dtype, int64_t n, int model, int Tn, int H, int W, const void *dt,
                         const int32_t *bin, const void *gwarped, void *gmotion, cmax_stream_t stream) {
    CMAX_REQUIRE(n >= 0 && gmotion && (n == 0 || (dt && gwarped)), "warp_events_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const size_t esz = dtype == CMAX_F64 ? 8 : 4;
    if (model == CMAX_MODEL_2DOF) {
        // fp64 accumulators live behind the output for the duration of the call: use a small
        // stream-ordered temporary instead (hipMallocAsync keeps the op stateless).
        double *acc = nullptr;
        CMAX_CHECK_HIP(hipMallocAsync((void **)&acc, 2 * sizeof(double), s));
        CMAX_CHECK_HIP(hipMemsetAsync(acc, 0, 2 * sizeof(double), s));
        if (n > 0) DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_warp_bwd_2dof<T>, dim3(stream_grid(n, 256)), dim3(256), 0, s, n, (const T *)dt, (const T *)gwarped, acc));
        DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_cast2<T>, dim3(1), dim3(1), 0, s, acc, (T *)gmotion));
        CMAX_CHECK_LAUNCH();
        CMAX_CHECK_HIP(hipFreeAsync(acc, s));
        return 0;
    }
    CMAX_REQUIRE(model == CMAX_MODEL_DENSE || model == CMAX_MODEL_VOXEL, "warp_events_bwd: model");
    CMAX_REQUIRE(H > 0 && W > 0 && events, "warp_events_bwd: image size / events");
    CMAX_REQUIRE(model != CMAX_MODEL_VOXEL || (Tn > 0 && bin), "warp_events_bwd: voxel needs T and bin");
    const int64_t cnt = (int64_t)(model == CMAX_MODEL_VOXEL ? Tn : 1) * 2 * H * W;
    CMAX_CHECK_HIP(hipMemsetAsync(gmotion, 0, cnt * esz, s));
    if (n == 0) return 0;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_warp_bwd_flow<T>, dim3(stream_grid(n, 256)), dim3(256), 0, s, (const T *)events, n, H, W,
                                             (const T *)dt, model == CMAX_MODEL_VOXEL ? bin : nullptr, (const T *)gwarped, (T *)gmotion));
    CMAX_CHECK_LAUNCH();
    return 0;
}

int cmax_vote(const void *xy, int dtype, int64_t stride, int64_t n, const void *weight, double wscalar, int Hp, int Wp,
              int ph, int pw, double eps, int count, void *img, cmax_stream_t stream) {
    CMAX_REQUIRE(img && Hp > 0 && Wp > 0 && n >= 0 && (n == 0 || xy) && stride >= 2, "vote");
    hipStream_t s = (hipStream_t)stream;
    const size_t esz = dtype == CMAX_F64 ? 8 : 4;
    CMAX_REQUIRE(dtype == CMAX_F32 || dtype == CMAX_F64, "vote: dtype");
    CMAX_CHECK_HIP(hipMemsetAsync(img, 0, (size_t)Hp * Wp * esz, s));
    if (n == 0) return 0;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_vote<T>, dim3(stream_grid(n, 256)), dim3(256), 0, s, (const T *)xy, stride, n,
                                             (const T *)weight, (T)wscalar, Hp, Wp, ph, pw, (T)eps, count, (T *)img));
    CMAX_CHECK_LAUNCH();
    return 0;
}

int cmax_vote_bwd(const void *xy, int dtype, int64_t stride, int64_t n, const void *weight, double wscalar, int Hp,
                  int Wp, int ph, int pw, double eps, const void *G, void *gxy, void *gw, cmax_stream_t stream) {
    CMAX_REQUIRE(G && Hp > 0 && Wp > 0 && n >= 0 && (n == 0 || (xy && gxy)) && stride >= 2, "vote_bwd");
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_vote_bwd<T>, dim3(stream_grid(n, 256)), dim3(256), 0, s, (const T *)xy, stride, n,
                                             (const T *)weight, (T)wscalar, Hp, Wp, ph, pw, (T)eps, (const T *)G, (T *)gxy, (T *)gw));
    CMAX_CHECK_LAUNCH();
    return 0;
}

int cmax_blur3(const void *in, int dtype, int H, int W, double sigma, int adjoint, void *out, cmax_stream_t stream) {
    CMAX_REQUIRE(in && out && in != out && H > 0 && W > 0 && sigma > 0, "blur3");
    double k0, k1;
    blur_taps(sigma, k0, k1);
    hipStream_t s = (hipStream_t)stream;
    const int grid = div_up((int64_t)H * W, 256);
    if (adjoint) DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_blur3_adj<T>, dim3(grid), dim3(256), 0, s, (const T *)in, H, W, (T)k0, (T)k1, (T *)out));
    else DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_blur3<T>, dim3(grid), dim3(256), 0, s, (const T *)in, H, W, (T)k0, (T)k1, (T *)out));
    CMAX_CHECK_LAUNCH();
    return 0;
}

int cmax_contrast(const void *img, int dtype, int H, int W, int cost, int omit_boundary, int ddof, double *value,
                  void *G, const double *gscale, cmax_stream_t stream) {
    CMAX_REQUIRE(img && value && H > 0 && W > 0, "contrast");
    CMAX_REQUIRE(cost == CMAX_COST_VARIANCE || cost == CMAX_COST_GRADMAG, "contrast: cost");
    CMAX_REQUIRE(!omit_boundary || (H > 2 && W > 2), "contrast: image too small for omit_boundary");
    hipStream_t s = (hipStream_t)stream;
    CMAX_CHECK_HIP(hipMemsetAsync(value, 0, 4 * sizeof(double), s));
    const int64_t npix = (int64_t)H * W;
    const int rgrid = stream_grid(npix, 256), fgrid = div_up(npix, 256);
    if (cost == CMAX_COST_VARIANCE) {
        DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_var_sums<T>, dim3(rgrid), dim3(256), 0, s, (const T *)img, H, W, omit_boundary, value));
        DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_var_final<T>, dim3(G ? fgrid : 1), dim3(256), 0, s, (const T *)img, H, W, omit_boundary, ddof, value, (T *)G, gscale));
    } else {
        DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_gm_sums<T>, dim3(rgrid), dim3(256), 0, s, (const T *)img, H, W, omit_boundary, value));
        DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_gm_final<T>, dim3(G ? fgrid : 1), dim3(256), 0, s, (const T *)img, H, W, omit_boundary, value, (T *)G, gscale));
    }
    CMAX_CHECK_LAUNCH();
    return 0;
}

int cmax_total_variation(const void *flow, int dtype, int h, int w, int omit_boundary, double *value, void *G,
                         const double *gscale, cmax_stream_t stream) {
    CMAX_REQUIRE(flow && value && h > 0 && w > 0, "total_variation");
    hipStream_t s = (hipStream_t)stream;
    const int crop = omit_boundary && h > 2 && w > 2;  // total_variation.py:123-125
    CMAX_CHECK_HIP(hipMemsetAsync(value, 0, 4 * sizeof(double), s));
    const int64_t npix = 2 * (int64_t)h * w;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_tv_sums<T>, dim3(stream_grid(npix, 256)), dim3(256), 0, s, (const T *)flow, h, w, crop, value));
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(k_tv_final<T>, dim3(G ? div_up(npix, 256) : 1), dim3(256), 0, s, (const T *)flow, h, w, crop, value, (T *)G, gscale));
    CMAX_CHECK_LAUNCH();
    return 0;
}

int cmax_gaussian_filter(const void *in, int dtype, int H, int W, double sigma, void *tmp, void *out, cmax_stream_t stream) {
    CMAX_REQUIRE(in && tmp && out && in != tmp && tmp != out && H > 0 && W > 0 && sigma > 0, "gaussian_filter");
    const int r = (int)(4.0 * sigma + 0.5);
    CMAX_REQUIRE(r <= 64, "gaussian_filter: sigma too large (radius > 64)");
    hipStream_t s = (hipStream_t)stream;
    const int grid = div_up((int64_t)H * W, 256);
    const double inv2s2 = 0.5 / (sigma * sigma);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((k_gauss1d<T, 0>), dim3(grid), dim3(256), 0, s, (const T *)in, H, W, (T)inv2s2, r, (T *)tmp));
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((k_gauss1d<T, 1>), dim3(grid), dim3(256), 0, s, (const T *)tmp, H, W, (T)inv2s2, r, (T *)out));
    CMAX_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
