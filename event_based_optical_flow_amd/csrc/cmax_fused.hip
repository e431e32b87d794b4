// Fused contrast-maximization objective for MI355X (gfx950).
//
// One evaluation = what PatchContrastMaximization.get_arg_for_cost + cost.calculate +
// torch.autograd.grad compute in the reference (src/solver/patch_contrast_base.py:273-352,
// src/solver/scipy_autograd/torch_wrapper.py:30-49), without ever materialising warped events:
//
//   set_events (once per batch)  pack to 8 B/event (12-bit row | 12-bit col | 8-bit time bin, fp32
//                                normalised time) and counting-sort by source-pixel tile
//   K1  k_vote      per event: warp (2-DoF / dense / voxel) + bilinear vote  -> IWE   (atomics)
//   K2  k_stats     IWE (blurred if sigma>0) -> fp64 sums of the contrast function
//   K2b k_gimage    G = dL/dIWE (chain factor from the device-side stats), blur transpose
//   K3  k_grad      per event: re-warp, gather G at the 4 corners -> dL/d(x',y') -> motion gradient
//
// fp32 per event with the integer source pixel split from the fp32 displacement (keeps the
// bilinear fractions accurate to ulp(displacement) instead of ulp(coordinate)); fp64 reductions.
#include <vector>

#include "cmax_common.h"
#include "cmax_image_kernels.h"

namespace cmax {

constexpr int kTile = 16;  // source-pixel tile edge of the counting sort
constexpr uint32_t kDropped = 0xFFFFFFFFu;

struct EvView {
    const uint32_t *xyb;  // row | col << 12 | bin << 24
    const float *tau;     // (t - tmin) / (tmax - tmin)
    const float *rx;      // fractional residual of the source coordinate (nullptr if integral)
    const float *ry;
    int64_t n;
};

struct WarpParams {
    int H, W, Hp, Wp, ph, pw;  // un-padded sensor, padded image, padding
    int T;                     // voxel bins
    float d;                   // reference time as a fraction of the batch period
    int normalize;             // normalize_t
    const double *tmm;         // device (tmin, tmax)
    const float *motion;       // theta[2] | flow[2,H,W] | voxel[T,2,H,W]
};

}  // namespace cmax

struct cmax_handle_s {
    int H = 0, W = 0, ph = 0, pw = 0, Hp = 0, Wp = 0;
    int device = 0;
    int64_t n = 0, cap = 0;
    bool has_frac = false;
    int n_time_bin = 0;
    // packed, sorted events
    uint32_t *xyb = nullptr;
    float *tau = nullptr, *rx = nullptr, *ry = nullptr;
    double *tau64 = nullptr;
    // sort scratch
    uint32_t *key_tmp = nullptr;
    int *counts = nullptr;  // [nkeys + 1] -> offsets after the scan
    int *cursor = nullptr;  // [nkeys]
    int nkeys = 0, ntr = 0, ntc = 0;
    int *d_flags = nullptr;  // [0] any fractional source coordinate, [1] dropped events
    // images
    float *imgs = nullptr;                                  // [5, Hp, Wp] raw votes: one per reference time + un-warped
    float *iweb[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // blurred copies
    float *G = nullptr, *Gt = nullptr;
    const float *last_iwe[4] = {nullptr, nullptr, nullptr, nullptr};
    // device scalars
    double *d_tmm = nullptr;  // [2]
    double *d_acc = nullptr;  // [16] accumulators: slot k -> [2k, 2k+1]; orig -> [8, 9]
    // orig-IWE cache key
    bool orig_valid = false;
    double orig_sigma = -1;
    int orig_cost = -1, orig_omit = -1;
    int64_t bytes = 0;
    // optional per-kernel-class timing with HIP events (cmax_set_profiling)
    bool profiling = false;
    std::vector<hipEvent_t> prof_ev[4];  // class -> [start0, stop0, start1, stop1, ...]
};

namespace cmax {

// kernel classes for cmax_set_profiling / cmax_read_profile
enum { kProfVote = 0, kProfStats = 1, kProfGimage = 2, kProfGrad = 3 };
constexpr size_t kProfMaxPairs = 16384;

// RAII: records a HIP event on the launch stream before and after the enclosed launch
struct ProfScope {
    cmax_handle_s *h;
    int cls;
    hipStream_t s;
    hipEvent_t stop = nullptr;
    ProfScope(cmax_handle_s *h_, int cls_, hipStream_t s_) : h(h_), cls(cls_), s(s_) {
        if (!h->profiling || h->prof_ev[cls].size() >= 2 * kProfMaxPairs) return;
        hipEvent_t start = nullptr;
        if (hipEventCreate(&start) != hipSuccess || hipEventCreate(&stop) != hipSuccess) {
            stop = nullptr;
            return;
        }
        (void)hipEventRecord(start, s);
        h->prof_ev[cls].push_back(start);
        h->prof_ev[cls].push_back(stop);
    }
    ~ProfScope() {
        if (stop) (void)hipEventRecord(stop, s);
    }
};

// ---------------------------------------------------------------------------------------------
// memory helpers
// ---------------------------------------------------------------------------------------------
template <typename T>
static int dev_alloc(cmax_handle_s *h, T **p, int64_t count) {
    if (count <= 0) count = 1;
    hipError_t e = hipMalloc((void **)p, (size_t)count * sizeof(T));
    if (e != hipSuccess) {
        set_error("hipMalloc(%lld bytes) failed: %s", (long long)(count * sizeof(T)), hipGetErrorString(e));
        return CMAX_ENOMEM;
    }
    h->bytes += count * (int64_t)sizeof(T);
    return 0;
}
template <typename T>
static void dev_free(T **p) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
}

// ---------------------------------------------------------------------------------------------
// set_events: pack + counting sort by source tile
// ---------------------------------------------------------------------------------------------
__global__ void k_tmm_set(double *tmm, double lo, double hi) {
    tmm[0] = lo;
    tmm[1] = hi;
}

__device__ __forceinline__ int voxel_bin(double tau, int T) {
    // reference edges for direction "first": e_k = k/T * (dtmax - dtmin) + dtmin with dt in [0,1]
    // (src/warp.py:342-345); the event belongs to the last k with e_k <= dt.
    int k = (int)(tau * (double)T);
    if (k > T - 1) k = T - 1;
    if (k < 0) k = 0;
    while (k > 0 && ((double)k / (double)T) > tau) --k;
    while (k + 1 < T && ((double)(k + 1) / (double)T) <= tau) ++k;
    return k;
}

// pass 1: sort key per event + histogram
template <typename T>
__global__ void __launch_bounds__(256)
k_pack_hist(const T *__restrict__ ev, int64_t n, int H, int W, int ntc, uint32_t *__restrict__ key, int *__restrict__ counts,
            int *__restrict__ flags) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        T x = ev[4 * i + 0], y = ev[4 * i + 1];
        T fx = floor_t<T>(x), fy = floor_t<T>(y);
        uint32_t k = kDropped;
        if (fx >= (T)0 && fx < (T)H && fy >= (T)0 && fy < (T)W) {  // NaN fails every comparison -> dropped
            int ix = (int)fx, iy = (int)fy;
            k = (uint32_t)(((ix / kTile) * ntc + (iy / kTile)) * (kTile * kTile) + (ix % kTile) * kTile + (iy % kTile));
            atomicAdd(&counts[k], 1);
            if (x != fx || y != fy) flags[0] = 1;
        } else {
            atomicAdd(&flags[1], 1);
        }
        key[i] = k;
    }
}

// single-workgroup exclusive scan of counts[0..m) in place; counts[m] = total
__global__ void __launch_bounds__(1024) k_scan(int *__restrict__ counts, int m) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int chunk = (m + 1023) / 1024;
    const int b = t * chunk, e = min(b + chunk, m);
    int s = 0;
    for (int i = b; i < e; ++i) s += counts[i];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan of the per-thread totals
        int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;
    for (int i = b; i < e; ++i) {
        int c = counts[i];
        counts[i] = run;
        run += c;
    }
    if (t == 1023) counts[m] = part[1023];
}

// pass 2: scatter into the sorted, packed SoA
template <typename T>
__global__ void __launch_bounds__(256)
k_scatter(const T *__restrict__ ev, int64_t n, const uint32_t *__restrict__ key, const int *__restrict__ offsets,
          int *__restrict__ cursor, const double *__restrict__ tmm, int n_time_bin, uint32_t *__restrict__ xyb,
          float *__restrict__ tau, float *__restrict__ rx, float *__restrict__ ry, double *__restrict__ tau64) {
    const double tmin = tmm[0], per = tmm[1] - tmm[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t k = key[i];
        if (k == kDropped) continue;
        int pos = offsets[k] + atomicAdd(&cursor[k], 1);
        T x = ev[4 * i + 0], y = ev[4 * i + 1];
        T fx = floor_t<T>(x), fy = floor_t<T>(y);
        double tn = per > 0 ? ((double)ev[4 * i + 2] - tmin) / per : 0.0;
        uint32_t bin = n_time_bin > 0 ? (uint32_t)voxel_bin(tn, n_time_bin) : 0u;
        xyb[pos] = (uint32_t)(int)fx | ((uint32_t)(int)fy << 12) | (bin << 24);
        tau[pos] = (float)tn;
        rx[pos] = (float)(x - fx);
        ry[pos] = (float)(y - fy);
        tau64[pos] = tn;
    }
}

__global__ void __launch_bounds__(256) k_rebin(int64_t n, const double *__restrict__ tau64, int n_time_bin, uint32_t *__restrict__ xyb) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t bin = n_time_bin > 0 ? (uint32_t)voxel_bin(tau64[i], n_time_bin) : 0u;
        xyb[i] = (xyb[i] & 0x00FFFFFFu) | (bin << 24);
    }
}

// ---------------------------------------------------------------------------------------------
// per-event warp shared by K1 and K3
// ---------------------------------------------------------------------------------------------
struct Warped {
    int row, col;  // top-left corner in the padded image
    float a, b;    // row / column fractions
    float dt;
    int src;       // source pixel linear index (un-padded), + bin * 2HW for voxel
};

// MODEL: -1 none (orig_iwe), 0 2-DoF, 1 dense, 2 voxel
template <int MODEL, bool FRAC>
__device__ __forceinline__ Warped warp_one(const EvView &ev, int64_t i, const WarpParams &wp, float tscale, float th0, float th1) {
    Warped w;
    const uint32_t pk = ev.xyb[i];
    const int ix = (int)(pk & 0xFFFu), iy = (int)((pk >> 12) & 0xFFFu);
    w.dt = (ev.tau[i] - wp.d) * tscale;  // calculate_dt, src/warp.py:254-259
    float dx = FRAC ? ev.rx[i] : 0.f, dy = FRAC ? ev.ry[i] : 0.f;
    w.src = ix * wp.W + iy;
    if (MODEL == CMAX_MODEL_2DOF) {
        dx = fmaf(w.dt, th0, dx);  // x' = x + dt*theta0, src/warp.py:506-515
        dy = fmaf(w.dt, th1, dy);
    } else if (MODEL == CMAX_MODEL_DENSE || MODEL == CMAX_MODEL_VOXEL) {
        const int hw = wp.H * wp.W;
        if (MODEL == CMAX_MODEL_VOXEL) w.src += (int)(pk >> 24) * 2 * hw;
        dx = fmaf(-w.dt, wp.motion[w.src], dx);  // x' = x - dt*F[0,ix,iy], src/warp.py:305-306
        dy = fmaf(-w.dt, wp.motion[w.src + hw], dy);
    }
    // floor(x' + 1e-6) = ix + floor(dx + 1e-6) exactly because ix is an integer
    // (bilinear_vote_tensor, src/event_image_converter.py:340-345)
    const float fx = floorf(dx + 1e-6f), fy = floorf(dy + 1e-6f);
    w.a = dx - fx;
    w.b = dy - fy;
    const float cx = fminf(fmaxf(fx, -8192.f), 8192.f), cy = fminf(fmaxf(fy, -8192.f), 8192.f);
    w.row = ix + (int)cx + wp.ph;
    w.col = iy + (int)cy + wp.pw;
    return w;
}

__device__ __forceinline__ float time_scale(const WarpParams &wp) {
    return wp.normalize ? 1.0f : (float)(wp.tmm[1] - wp.tmm[0]);
}

// contiguous chunk of the sorted stream per workgroup (spatially coherent votes / gathers)
__device__ __forceinline__ void chunk_range(int64_t n, int64_t &b, int64_t &e) {
    int64_t chunk = (n + gridDim.x - 1) / gridDim.x;
    chunk = (chunk + 63) & ~(int64_t)63;
    b = (int64_t)blockIdx.x * chunk;
    e = b + chunk < n ? b + chunk : n;
}

// ---------------------------------------------------------------------------------------------
// K1: warp + bilinear vote (global fp32 atomics)
// ---------------------------------------------------------------------------------------------
template <int MODEL, bool FRAC>
__global__ void __launch_bounds__(256) k_vote(EvView ev, WarpParams wp, float *__restrict__ iwe) {
    const float tscale = time_scale(wp);
    float th0 = 0.f, th1 = 0.f;
    if (MODEL == CMAX_MODEL_2DOF) {
        th0 = wp.motion[0];
        th1 = wp.motion[1];
    }
    int64_t b, e;
    chunk_range(ev.n, b, e);
    for (int64_t i = b + threadIdx.x; i < e; i += blockDim.x) {
        const Warped w = warp_one<MODEL, FRAC>(ev, i, wp, tscale, th0, th1);
        const bool r0 = (unsigned)w.row < (unsigned)wp.Hp, r1 = (unsigned)(w.row + 1) < (unsigned)wp.Hp;
        const bool c0 = (unsigned)w.col < (unsigned)wp.Wp, c1 = (unsigned)(w.col + 1) < (unsigned)wp.Wp;
        float *p = iwe + (int64_t)w.row * wp.Wp + w.col;
        const float na = 1.f - w.a, nb = 1.f - w.b;
        if (r0 && c0) atomic_add(p, na * nb);             // w_pos0, event_image_converter.py:365
        if (r1 && c0) atomic_add(p + wp.Wp, w.a * nb);    // w_pos1
        if (r0 && c1) atomic_add(p + 1, na * w.b);        // w_pos2
        if (r1 && c1) atomic_add(p + wp.Wp + 1, w.a * w.b);  // w_pos3
    }
}

// ---------------------------------------------------------------------------------------------
// K2: contrast statistics of one image.  acc[0] += sum x (variance) or sum gx^2+gy^2 (grad-mag),
//     acc[1] += sum x^2.
// ---------------------------------------------------------------------------------------------
template <int COST>
__global__ void __launch_bounds__(256) k_stats(const float *__restrict__ img, int H, int W, int omit, double *__restrict__ acc) {
    __shared__ double smem[2 * 4];
    const int i0 = omit ? 1 : 0, h = H - 2 * i0, w = W - 2 * i0;
    const int64_t n = (int64_t)h * w;
    double v[2] = {0.0, 0.0};
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(q / w) + i0, j = (int)(q % w) + i0;
        if (COST == CMAX_COST_VARIANCE) {
            const double x = (double)img[(int64_t)i * W + j];
            v[0] += x;
            v[1] += x * x;
        } else {
            double gx, gy;
            sobel8<float>(img, H, W, i, j, gx, gy);
            v[0] += gx * gx + gy * gy;
        }
    }
    block_sum<2>(v, smem);
    if (threadIdx.x == 0) {
        atomic_add(&acc[0], v[0]);
        if (COST == CMAX_COST_VARIANCE) atomic_add(&acc[1], v[1]);
    }
}

struct ObjParams {
    int cost, normalized, minimize, negate, omit, n_ref;
    double mult[4];
    int H, W;  // padded image
};

// raw contrast of slot k from its accumulators (variance: unbiased, torch.var, image_variance.py:55)
__device__ __forceinline__ double contrast_value(int cost, const double *acc, double npix, double *mu_out) {
    if (cost == CMAX_COST_VARIANCE) {
        const double mu = acc[0] / npix;
        if (mu_out) *mu_out = mu;
        return (acc[1] - acc[0] * mu) / (npix - 1.0);
    }
    return acc[0] / npix;
}

__device__ __forceinline__ double region_pixels(int H, int W, int omit) {
    const int i0 = omit ? 1 : 0;
    return (double)(H - 2 * i0) * (double)(W - 2 * i0);
}

// loss + per-slot values from the accumulators -> result[0..5]
__device__ void finalize_result(const ObjParams &op, const double *__restrict__ acc, double *__restrict__ result) {
    const double npix = region_pixels(op.H, op.W, op.omit);
    double v_orig = 0.0;
    if (op.normalized) {
        // orig_iwe is NOT cropped for the variance (normalized_image_variance.py:40-41)
        const int omit_o = op.cost == CMAX_COST_VARIANCE ? 0 : op.omit;
        v_orig = contrast_value(op.cost, acc + 8, region_pixels(op.H, op.W, omit_o), nullptr);
    }
    double loss = 0.0;
    for (int k = 0; k < op.n_ref; ++k) {
        const double v = contrast_value(op.cost, acc + 2 * k, npix, nullptr);
        result[1 + k] = v;
        if (!op.normalized) loss += op.mult[k] * (op.minimize ? -v : v);
        else loss += op.mult[k] * (op.minimize ? v_orig / v : v / v_orig);
    }
    if (op.negate) loss = -loss;
    result[0] = loss;
    result[5] = v_orig;
}

__global__ void k_finalize(ObjParams op, const double *__restrict__ acc, double *__restrict__ result) {
    if (threadIdx.x == 0 && blockIdx.x == 0) finalize_result(op, acc, result);
}

// K2b: G[p] = dL/dv_k * dv_k/dI[p].  The last slot's launch also writes the loss.
template <int COST>
__global__ void __launch_bounds__(256)
k_gimage(const float *__restrict__ img, ObjParams op, int k, const double *__restrict__ acc, float *__restrict__ G,
         double *__restrict__ result, int write_result) {
    const int H = op.H, W = op.W;
    const int i0 = op.omit ? 1 : 0;
    const double npix = region_pixels(H, W, op.omit);
    double mu = 0.0;
    const double v = contrast_value(COST, acc + 2 * k, npix, &mu);
    double coef;
    if (!op.normalized) {
        coef = op.mult[k] * (op.minimize ? -1.0 : 1.0);
    } else {
        const int omit_o = COST == CMAX_COST_VARIANCE ? 0 : op.omit;
        const double v_orig = contrast_value(COST, acc + 8, region_pixels(H, W, omit_o), nullptr);
        coef = op.mult[k] * (op.minimize ? -v_orig / (v * v) : 1.0 / v_orig);
    }
    if (op.negate) coef = -coef;
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p == 0 && write_result) finalize_result(op, acc, result);
    if (p >= (int64_t)H * W) return;
    const int i = (int)(p / W), j = (int)(p % W);
    if (COST == CMAX_COST_VARIANCE) {
        const bool in = (i >= i0) && (i < H - i0) && (j >= i0) && (j < W - i0);
        G[p] = in ? (float)(coef * 2.0 * ((double)img[p] - mu) / (npix - 1.0)) : 0.f;
    } else {
        G[p] = (float)(coef * (2.0 / npix) * sobel8_adj<float>(img, H, W, i0, i, j) / 8.0);
    }
}

// ---------------------------------------------------------------------------------------------
// K3: per-event gradient.  dL/dx' = (1-b)(G10-G00) + b(G11-G01), dL/dy' = (1-a)(G01-G00) + a(G11-G10)
//     2-DoF : gtheta += dt * (gx, gy)        fp64 block reduction + one fp64 atomic per workgroup
//     dense : gflow[c, src] += -dt * g_c     fp32 atomics (events of one source pixel are adjacent)
// ---------------------------------------------------------------------------------------------
template <int MODEL, bool FRAC>
__global__ void __launch_bounds__(256) k_grad(EvView ev, WarpParams wp, const float *__restrict__ G, void *__restrict__ grad) {
    __shared__ double smem[2 * 4];
    const float tscale = time_scale(wp);
    float th0 = 0.f, th1 = 0.f;
    if (MODEL == CMAX_MODEL_2DOF) {
        th0 = wp.motion[0];
        th1 = wp.motion[1];
    }
    const int hw = wp.H * wp.W;
    double acc[2] = {0.0, 0.0};
    int64_t b, e;
    chunk_range(ev.n, b, e);
    for (int64_t i = b + threadIdx.x; i < e; i += blockDim.x) {
        const Warped w = warp_one<MODEL, FRAC>(ev, i, wp, tscale, th0, th1);
        const bool r0 = (unsigned)w.row < (unsigned)wp.Hp, r1 = (unsigned)(w.row + 1) < (unsigned)wp.Hp;
        const bool c0 = (unsigned)w.col < (unsigned)wp.Wp, c1 = (unsigned)(w.col + 1) < (unsigned)wp.Wp;
        const float *p = G + (int64_t)w.row * wp.Wp + w.col;
        const float g00 = (r0 && c0) ? p[0] : 0.f, g10 = (r1 && c0) ? p[wp.Wp] : 0.f;
        const float g01 = (r0 && c1) ? p[1] : 0.f, g11 = (r1 && c1) ? p[wp.Wp + 1] : 0.f;
        const float gx = (1.f - w.b) * (g10 - g00) + w.b * (g11 - g01);
        const float gy = (1.f - w.a) * (g01 - g00) + w.a * (g11 - g10);
        if (MODEL == CMAX_MODEL_2DOF) {
            acc[0] += (double)(w.dt * gx);
            acc[1] += (double)(w.dt * gy);
        } else {
            float *gf = reinterpret_cast<float *>(grad);
            atomic_add(&gf[w.src], -w.dt * gx);
            atomic_add(&gf[w.src + hw], -w.dt * gy);
        }
    }
    if (MODEL == CMAX_MODEL_2DOF) {
        block_sum<2>(acc, smem);
        if (threadIdx.x == 0) {
            double *gt = reinterpret_cast<double *>(grad);
            atomic_add(&gt[0], acc[0]);
            atomic_add(&gt[1], acc[1]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host-side orchestration
// ---------------------------------------------------------------------------------------------
static int event_grid(int64_t n) {
    // >= 2 workgroups per CU on 256 CUs, at least ~512 events per workgroup
    int64_t g = (n + 511) / 512;
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    return (int)g;
}

static float ref_fraction(int ref_mode, double frac) {
    if (ref_mode == CMAX_REF_FIRST) return 0.f;
    if (ref_mode == CMAX_REF_LAST) return 1.f;
    return (float)frac;
}

template <int MODEL>
static void launch_vote(cmax_handle_s *h, const EvView &ev, const WarpParams &wp, float *img, hipStream_t s) {
    const int grid = event_grid(ev.n);
    ProfScope prof(h, kProfVote, s);
    if (h->has_frac) hipLaunchKernelGGL((k_vote<MODEL, true>), dim3(grid), dim3(256), 0, s, ev, wp, img);
    else hipLaunchKernelGGL((k_vote<MODEL, false>), dim3(grid), dim3(256), 0, s, ev, wp, img);
}

template <int MODEL>
static void launch_grad(cmax_handle_s *h, const EvView &ev, const WarpParams &wp, const float *G, void *grad, hipStream_t s) {
    const int grid = event_grid(ev.n);
    ProfScope prof(h, kProfGrad, s);
    if (h->has_frac) hipLaunchKernelGGL((k_grad<MODEL, true>), dim3(grid), dim3(256), 0, s, ev, wp, G, grad);
    else hipLaunchKernelGGL((k_grad<MODEL, false>), dim3(grid), dim3(256), 0, s, ev, wp, G, grad);
}

static EvView ev_view(const cmax_handle_s *h) {
    EvView ev;
    ev.xyb = h->xyb;
    ev.tau = h->tau;
    ev.rx = h->rx;
    ev.ry = h->ry;
    ev.n = h->n;
    return ev;
}

static WarpParams warp_params(const cmax_handle_s *h, const float *motion, int T, int ref_mode, double frac, int normalize) {
    WarpParams wp;
    wp.H = h->H;
    wp.W = h->W;
    wp.Hp = h->Hp;
    wp.Wp = h->Wp;
    wp.ph = h->ph;
    wp.pw = h->pw;
    wp.T = T;
    wp.d = ref_fraction(ref_mode, frac);
    wp.normalize = normalize;
    wp.tmm = h->d_tmm;
    wp.motion = motion;
    return wp;
}

// raw votes of one reference time into `raw` (zeroed here)
static int vote_image(cmax_handle_s *h, int model, const float *motion, int T, int ref_mode, double frac, int normalize,
                      float *raw, hipStream_t s) {
    const int64_t npix = (int64_t)h->Hp * h->Wp;
    CMAX_CHECK_HIP(hipMemsetAsync(raw, 0, npix * sizeof(float), s));
    if (h->n == 0) return 0;
    const EvView ev = ev_view(h);
    const WarpParams wp = warp_params(h, motion, T, ref_mode, frac, normalize);
    switch (model) {
        case CMAX_MODEL_2DOF: launch_vote<CMAX_MODEL_2DOF>(h, ev, wp, raw, s); break;
        case CMAX_MODEL_DENSE: launch_vote<CMAX_MODEL_DENSE>(h, ev, wp, raw, s); break;
        case CMAX_MODEL_VOXEL: launch_vote<CMAX_MODEL_VOXEL>(h, ev, wp, raw, s); break;
        default: launch_vote<-1>(h, ev, wp, raw, s); break;
    }
    CMAX_CHECK_LAUNCH();
    return 0;
}

// the image the contrast is evaluated on: `raw`, or its blurred copy in `blur` when sigma > 0
static int blur_image(cmax_handle_s *h, double sigma, const float *raw, float *blur, const float **out, hipStream_t s) {
    *out = raw;
    if (sigma > 0) {
        const int64_t npix = (int64_t)h->Hp * h->Wp;
        double k0, k1;
        blur_taps(sigma, k0, k1);
        hipLaunchKernelGGL(k_blur3<float>, dim3(div_up(npix, 256)), dim3(256), 0, s, raw, h->Hp, h->Wp, (float)k0, (float)k1, blur);
        CMAX_CHECK_LAUNCH();
        *out = blur;
    }
    return 0;
}

static int launch_stats(cmax_handle_s *h, int cost, const float *img, int Hp, int Wp, int omit, double *acc, hipStream_t s) {
    const int grid = stream_grid((int64_t)Hp * Wp, 256);
    ProfScope prof(h, kProfStats, s);
    if (cost == CMAX_COST_VARIANCE) hipLaunchKernelGGL(k_stats<CMAX_COST_VARIANCE>, dim3(grid), dim3(256), 0, s, img, Hp, Wp, omit, acc);
    else hipLaunchKernelGGL(k_stats<CMAX_COST_GRADMAG>, dim3(grid), dim3(256), 0, s, img, Hp, Wp, omit, acc);
    CMAX_CHECK_LAUNCH();
    return 0;
}

}  // namespace cmax

using namespace cmax;

extern "C" {

int cmax_create(int H, int W, int ph, int pw, cmax_handle_t *out) {
    CMAX_REQUIRE(out != nullptr, "create: out");
    CMAX_REQUIRE(H > 0 && W > 0 && H <= 4096 && W <= 4096 && ph >= 0 && pw >= 0, "create: image size must be in 1..4096");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device visible");
        return CMAX_ENODEV;
    }
    cmax_handle_s *h = new cmax_handle_s();
    h->H = H;
    h->W = W;
    h->ph = ph;
    h->pw = pw;
    h->Hp = H + 2 * ph;
    h->Wp = W + 2 * pw;
    CMAX_CHECK_HIP(hipGetDevice(&h->device));
    h->ntr = div_up(H, kTile);
    h->ntc = div_up(W, kTile);
    h->nkeys = h->ntr * h->ntc * kTile * kTile;
    const int64_t npix = (int64_t)h->Hp * h->Wp;
    int rc = dev_alloc(h, &h->imgs, 5 * npix);
    for (int k = 0; k < 5 && !rc; ++k) rc = dev_alloc(h, &h->iweb[k], npix);
    if (!rc) rc = dev_alloc(h, &h->G, npix);
    if (!rc) rc = dev_alloc(h, &h->Gt, npix);
    if (!rc) rc = dev_alloc(h, &h->d_tmm, 2);
    if (!rc) rc = dev_alloc(h, &h->d_acc, 16);
    if (!rc) rc = dev_alloc(h, &h->counts, h->nkeys + 1);
    if (!rc) rc = dev_alloc(h, &h->cursor, h->nkeys);
    if (!rc) rc = dev_alloc(h, &h->d_flags, 2);
    if (rc) {
        cmax_destroy(h);
        return rc;
    }
    *out = h;
    return 0;
}

int cmax_destroy(cmax_handle_t h) {
    if (!h) return 0;
    dev_free(&h->imgs);
    for (int k = 0; k < 5; ++k) dev_free(&h->iweb[k]);
    dev_free(&h->G);
    dev_free(&h->Gt);
    dev_free(&h->d_tmm);
    dev_free(&h->d_acc);
    dev_free(&h->counts);
    dev_free(&h->cursor);
    dev_free(&h->d_flags);
    dev_free(&h->xyb);
    dev_free(&h->tau);
    dev_free(&h->rx);
    dev_free(&h->ry);
    dev_free(&h->tau64);
    dev_free(&h->key_tmp);
    for (int c = 0; c < 4; ++c)
        for (hipEvent_t e : h->prof_ev[c]) (void)hipEventDestroy(e);
    delete h;
    return 0;
}

int cmax_set_events(cmax_handle_t h, const void *events, int dtype, int64_t n, int have_tminmax, double tmin, double tmax,
                    int n_time_bin, cmax_stream_t stream) {
    CMAX_REQUIRE(h != nullptr, "set_events: handle");
    CMAX_REQUIRE(n >= 0 && n < (int64_t)2147483647 && (n == 0 || events), "set_events: n / events");
    CMAX_REQUIRE(dtype == CMAX_F32 || dtype == CMAX_F64, "set_events: dtype");
    CMAX_REQUIRE(n_time_bin >= 0 && n_time_bin <= 255, "set_events: n_time_bin must be in 0..255");
    hipStream_t s = (hipStream_t)stream;
    h->orig_valid = false;
    h->n = 0;
    if (n > h->cap) {
        // the old buffers may still be in use by work queued on the stream
        CMAX_CHECK_HIP(hipStreamSynchronize(s));
        dev_free(&h->xyb);
        dev_free(&h->tau);
        dev_free(&h->rx);
        dev_free(&h->ry);
        dev_free(&h->tau64);
        dev_free(&h->key_tmp);
        int rc = dev_alloc(h, &h->xyb, n);
        if (!rc) rc = dev_alloc(h, &h->tau, n);
        if (!rc) rc = dev_alloc(h, &h->rx, n);
        if (!rc) rc = dev_alloc(h, &h->ry, n);
        if (!rc) rc = dev_alloc(h, &h->tau64, n);
        if (!rc) rc = dev_alloc(h, &h->key_tmp, n);
        if (rc) return rc;
        h->cap = n;
    }
    // global time extremes
    if (have_tminmax) {
        CMAX_REQUIRE(tmax >= tmin, "set_events: tmax < tmin");
        hipLaunchKernelGGL(k_tmm_set, dim3(1), dim3(1), 0, s, h->d_tmm, tmin, tmax);
    } else {
        int rc = cmax_tminmax(events, dtype, n, h->d_tmm, stream);  // leaf reduction (cmax_leaf.hip)
        if (rc) return rc;
    }
    CMAX_CHECK_LAUNCH();
    CMAX_CHECK_HIP(hipMemsetAsync(h->counts, 0, (size_t)(h->nkeys + 1) * sizeof(int), s));
    CMAX_CHECK_HIP(hipMemsetAsync(h->cursor, 0, (size_t)h->nkeys * sizeof(int), s));
    CMAX_CHECK_HIP(hipMemsetAsync(h->d_flags, 0, 2 * sizeof(int), s));
    h->n_time_bin = n_time_bin;
    if (n == 0) {
        h->has_frac = false;
        return 0;
    }
    const int grid = stream_grid(n, 256);
    if (dtype == CMAX_F32) hipLaunchKernelGGL(k_pack_hist<float>, dim3(grid), dim3(256), 0, s, (const float *)events, n, h->H, h->W, h->ntc, h->key_tmp, h->counts, h->d_flags);
    else hipLaunchKernelGGL(k_pack_hist<double>, dim3(grid), dim3(256), 0, s, (const double *)events, n, h->H, h->W, h->ntc, h->key_tmp, h->counts, h->d_flags);
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, h->counts, h->nkeys);
    if (dtype == CMAX_F32) hipLaunchKernelGGL(k_scatter<float>, dim3(grid), dim3(256), 0, s, (const float *)events, n, h->key_tmp, h->counts, h->cursor, h->d_tmm, n_time_bin, h->xyb, h->tau, h->rx, h->ry, h->tau64);
    else hipLaunchKernelGGL(k_scatter<double>, dim3(grid), dim3(256), 0, s, (const double *)events, n, h->key_tmp, h->counts, h->cursor, h->d_tmm, n_time_bin, h->xyb, h->tau, h->rx, h->ry, h->tau64);
    CMAX_CHECK_LAUNCH();
    // once per batch: how many events survived, and whether any source coordinate is fractional
    int flags[2] = {0, 0};
    CMAX_CHECK_HIP(hipMemcpyAsync(flags, h->d_flags, sizeof(flags), hipMemcpyDeviceToHost, s));
    CMAX_CHECK_HIP(hipStreamSynchronize(s));
    h->has_frac = flags[0] != 0;
    h->n = n - flags[1];
    return 0;
}

int cmax_set_time_bins(cmax_handle_t h, int n_time_bin, cmax_stream_t stream) {
    CMAX_REQUIRE(h != nullptr, "set_time_bins: handle");
    CMAX_REQUIRE(n_time_bin >= 0 && n_time_bin <= 255, "set_time_bins: n_time_bin must be in 0..255");
    if (n_time_bin == h->n_time_bin) return 0;
    h->n_time_bin = n_time_bin;
    if (h->n == 0) return 0;
    hipLaunchKernelGGL(k_rebin, dim3(stream_grid(h->n, 256)), dim3(256), 0, (hipStream_t)stream, h->n, h->tau64, n_time_bin, h->xyb);
    CMAX_CHECK_LAUNCH();
    return 0;
}

int cmax_iwe(cmax_handle_t h, int model, const float *motion, int T, int ref_mode, double ref_frac, int normalize_t,
             double sigma, float *iwe_out, cmax_stream_t stream) {
    CMAX_REQUIRE(h != nullptr && iwe_out != nullptr, "iwe: handle / output");
    CMAX_REQUIRE(model < 0 || motion != nullptr, "iwe: motion");
    CMAX_REQUIRE(model <= CMAX_MODEL_VOXEL, "iwe: model");
    CMAX_REQUIRE(model != CMAX_MODEL_VOXEL || (T > 0 && T == h->n_time_bin), "iwe: voxel T must match the handle's time bins");
    hipStream_t s = (hipStream_t)stream;
    float *raw = sigma > 0 ? h->imgs : iwe_out;
    int rc = vote_image(h, model, motion, T, ref_mode, ref_frac, normalize_t, raw, s);
    if (rc) return rc;
    const float *img = nullptr;
    return blur_image(h, sigma, raw, iwe_out, &img, s);
}

static int check_objective_args(cmax_handle_t h, const cmax_objective_t *d, const float *motion) {
    CMAX_REQUIRE(h && d && motion, "objective: null pointer");
    CMAX_REQUIRE(d->model >= CMAX_MODEL_2DOF && d->model <= CMAX_MODEL_VOXEL, "objective: model");
    CMAX_REQUIRE(d->cost == CMAX_COST_VARIANCE || d->cost == CMAX_COST_GRADMAG, "objective: cost");
    CMAX_REQUIRE(d->n_ref >= 1 && d->n_ref <= 4, "objective: n_ref");
    CMAX_REQUIRE(d->model != CMAX_MODEL_VOXEL || (d->T > 0 && d->T == h->n_time_bin), "objective: voxel T must match the handle's time bins");
    CMAX_REQUIRE(!d->omit_boundary || (h->Hp > 2 && h->Wp > 2), "objective: image too small for omit_boundary");
    return 0;
}

static bool orig_cache_hit(const cmax_handle_s *h, const cmax_objective_t *d) {
    return h->orig_valid && h->orig_sigma == d->sigma && h->orig_cost == d->cost && h->orig_omit == d->omit_boundary;
}

int cmax_objective_vote(cmax_handle_t h, const cmax_objective_t *d, const float *motion, float *images, int *n_images_host,
                        cmax_stream_t stream) {
    int rc = check_objective_args(h, d, motion);
    if (rc) return rc;
    CMAX_REQUIRE(images && n_images_host, "objective_vote: images / n_images_host");
    hipStream_t s = (hipStream_t)stream;
    const int64_t npix = (int64_t)h->Hp * h->Wp;
    for (int k = 0; k < d->n_ref; ++k) {
        rc = vote_image(h, d->model, motion, d->T, d->ref_mode[k], d->ref_frac[k], d->normalize_t, images + k * npix, s);
        if (rc) return rc;
    }
    int n_images = d->n_ref;
    if (d->normalized && !orig_cache_hit(h, d)) {  // un-warped image, once per batch (patch_contrast_base.py:295-301)
        rc = vote_image(h, -1, nullptr, 0, CMAX_REF_FIRST, 0.0, 1, images + (int64_t)d->n_ref * npix, s);
        if (rc) return rc;
        ++n_images;
    }
    *n_images_host = n_images;
    return 0;
}

int cmax_objective_finish(cmax_handle_t h, const cmax_objective_t *d, const float *motion, const float *images, int n_images,
                          double *result, void *grad, cmax_stream_t stream) {
    int rc = check_objective_args(h, d, motion);
    if (rc) return rc;
    CMAX_REQUIRE(images && result, "objective_finish: images / result");
    CMAX_REQUIRE(n_images == d->n_ref || n_images == d->n_ref + 1, "objective_finish: n_images");
    hipStream_t s = (hipStream_t)stream;
    const int Hp = h->Hp, Wp = h->Wp;
    const int64_t npix = (int64_t)Hp * Wp;
    const int64_t gcount = d->model == CMAX_MODEL_2DOF ? 2 : (int64_t)(d->model == CMAX_MODEL_VOXEL ? d->T : 1) * 2 * h->H * h->W;
    const size_t gbytes = d->model == CMAX_MODEL_2DOF ? 2 * sizeof(double) : (size_t)gcount * sizeof(float);

    ObjParams op;
    op.cost = d->cost;
    op.normalized = d->normalized;
    op.minimize = d->minimize;
    op.negate = d->negate;
    op.omit = d->omit_boundary;
    op.n_ref = d->n_ref;
    for (int k = 0; k < 4; ++k) op.mult[k] = d->mult[k];
    op.H = Hp;
    op.W = Wp;

    // accumulators: slots [0..7]; the cached statistics of the un-warped image [8..9] survive
    CMAX_CHECK_HIP(hipMemsetAsync(h->d_acc, 0, 8 * sizeof(double), s));
    if (d->normalized) {
        if (n_images == d->n_ref + 1) {
            const float *img = nullptr;
            rc = blur_image(h, d->sigma, images + (int64_t)d->n_ref * npix, h->iweb[4], &img, s);
            if (rc) return rc;
            CMAX_CHECK_HIP(hipMemsetAsync(h->d_acc + 8, 0, 2 * sizeof(double), s));
            // orig_iwe is NOT boundary-cropped for the variance (normalized_image_variance.py:40-41)
            const int omit_o = d->cost == CMAX_COST_VARIANCE ? 0 : d->omit_boundary;
            rc = launch_stats(h, d->cost, img, Hp, Wp, omit_o, h->d_acc + 8, s);
            if (rc) return rc;
            h->orig_valid = true;
            h->orig_sigma = d->sigma;
            h->orig_cost = d->cost;
            h->orig_omit = d->omit_boundary;
        } else if (!orig_cache_hit(h, d)) {
            set_error("objective_finish: normalised cost needs the un-warped image (n_images == n_ref + 1)");
            return CMAX_ESTATE;
        }
    }

    // contrast statistics per reference time
    for (int k = 0; k < d->n_ref; ++k) {
        const float *img = nullptr;
        rc = blur_image(h, d->sigma, images + k * npix, h->iweb[k], &img, s);
        if (rc) return rc;
        h->last_iwe[k] = img;
        rc = launch_stats(h, d->cost, img, Hp, Wp, d->omit_boundary, h->d_acc + 2 * k, s);
        if (rc) return rc;
    }

    if (!grad) {
        hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, s, op, h->d_acc, result);
        CMAX_CHECK_LAUNCH();
        return 0;
    }

    // backward: G image (+ blur transpose) and the per-event gather, accumulated over reference times
    CMAX_CHECK_HIP(hipMemsetAsync(grad, 0, gbytes, s));
    const EvView ev = ev_view(h);
    double k0 = 0, k1 = 0;
    if (d->sigma > 0) blur_taps(d->sigma, k0, k1);
    for (int k = 0; k < d->n_ref; ++k) {
        const int last = k == d->n_ref - 1;
        float *Gk = d->sigma > 0 ? h->Gt : h->G;
        {
        ProfScope prof(h, kProfGimage, s);
        if (d->cost == CMAX_COST_VARIANCE)
            hipLaunchKernelGGL(k_gimage<CMAX_COST_VARIANCE>, dim3(div_up(npix, 256)), dim3(256), 0, s, h->last_iwe[k], op, k, h->d_acc, Gk, result, last);
        else
            hipLaunchKernelGGL(k_gimage<CMAX_COST_GRADMAG>, dim3(div_up(npix, 256)), dim3(256), 0, s, h->last_iwe[k], op, k, h->d_acc, Gk, result, last);
        }
        if (d->sigma > 0)
            hipLaunchKernelGGL(k_blur3_adj<float>, dim3(div_up(npix, 256)), dim3(256), 0, s, Gk, Hp, Wp, (float)k0, (float)k1, h->G);
        CMAX_CHECK_LAUNCH();
        if (h->n == 0) continue;
        const WarpParams wp = warp_params(h, motion, d->T, d->ref_mode[k], d->ref_frac[k], d->normalize_t);
        switch (d->model) {
            case CMAX_MODEL_2DOF: launch_grad<CMAX_MODEL_2DOF>(h, ev, wp, h->G, grad, s); break;
            case CMAX_MODEL_DENSE: launch_grad<CMAX_MODEL_DENSE>(h, ev, wp, h->G, grad, s); break;
            default: launch_grad<CMAX_MODEL_VOXEL>(h, ev, wp, h->G, grad, s); break;
        }
        CMAX_CHECK_LAUNCH();
    }
    return 0;
}

int cmax_objective(cmax_handle_t h, const cmax_objective_t *d, const float *motion, double *result, void *grad,
                   cmax_stream_t stream) {
    int rc = check_objective_args(h, d, motion);
    if (rc) return rc;
    CMAX_REQUIRE(result, "objective: result");
    // empty batch: loss 0, zero gradient (patch_contrast_base.py:253-255)
    if (h->n == 0) {
        hipStream_t s = (hipStream_t)stream;
        const int64_t gcount = d->model == CMAX_MODEL_2DOF ? 2 : (int64_t)(d->model == CMAX_MODEL_VOXEL ? d->T : 1) * 2 * h->H * h->W;
        const size_t gbytes = d->model == CMAX_MODEL_2DOF ? 2 * sizeof(double) : (size_t)gcount * sizeof(float);
        CMAX_CHECK_HIP(hipMemsetAsync(result, 0, 8 * sizeof(double), s));
        if (grad) CMAX_CHECK_HIP(hipMemsetAsync(grad, 0, gbytes, s));
        return 0;
    }
    int n_images = 0;
    rc = cmax_objective_vote(h, d, motion, h->imgs, &n_images, stream);
    if (rc) return rc;
    return cmax_objective_finish(h, d, motion, h->imgs, n_images, result, grad, stream);
}

int cmax_sizeof_objective(void) { return (int)sizeof(cmax_objective_t); }

static void prof_clear(cmax_handle_s *h) {
    for (int c = 0; c < 4; ++c) {
        for (hipEvent_t e : h->prof_ev[c]) (void)hipEventDestroy(e);
        h->prof_ev[c].clear();
    }
}

int cmax_set_profiling(cmax_handle_t h, int enable) {
    CMAX_REQUIRE(h != nullptr, "set_profiling");
    prof_clear(h);
    h->profiling = enable != 0;
    return 0;
}

int cmax_read_profile(cmax_handle_t h, double *total_ms_host, int64_t *count_host) {
    CMAX_REQUIRE(h && total_ms_host && count_host, "read_profile");
    for (int c = 0; c < 4; ++c) {
        double tot = 0.0;
        const size_t np = h->prof_ev[c].size() / 2;
        for (size_t i = 0; i < np; ++i) {
            CMAX_CHECK_HIP(hipEventSynchronize(h->prof_ev[c][2 * i + 1]));
            float ms = 0.f;
            CMAX_CHECK_HIP(hipEventElapsedTime(&ms, h->prof_ev[c][2 * i], h->prof_ev[c][2 * i + 1]));
            tot += (double)ms;
        }
        total_ms_host[c] = tot;
        count_host[c] = (int64_t)np;
    }
    prof_clear(h);
    return 0;
}

int cmax_copy_iwe(cmax_handle_t h, int k, float *iwe_out, cmax_stream_t stream) {
    CMAX_REQUIRE(h && iwe_out && k >= 0 && k < 4, "copy_iwe");
    if (!h->last_iwe[k]) {
        set_error("copy_iwe: no objective evaluated yet for slot %d", k);
        return CMAX_ESTATE;
    }
    CMAX_CHECK_HIP(hipMemcpyAsync(iwe_out, h->last_iwe[k], (size_t)h->Hp * h->Wp * sizeof(float), hipMemcpyDeviceToDevice,
                                  (hipStream_t)stream));
    return 0;
}

int cmax_handle_info(cmax_handle_t h, int64_t *n_events, int64_t *workspace_bytes) {
    CMAX_REQUIRE(h != nullptr, "handle_info");
    if (n_events) *n_events = h->n;
    if (workspace_bytes) *workspace_bytes = h->bytes;
    return 0;
}

}  // extern "C"
