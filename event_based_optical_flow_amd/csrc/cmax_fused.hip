// Fused contrast-maximization objective for MI355X (gfx950).
//
// One evaluation = what PatchContrastMaximization.get_arg_for_cost + cost.calculate +
// torch.autograd.grad compute in the reference (src/solver/patch_contrast_base.py:273-352,
// src/solver/scipy_autograd/torch_wrapper.py:30-49), without ever materialising warped events:
//
//   set_events (once per batch)  pack to 8 B/event (12-bit row | 12-bit col | 8-bit time bin, fp32
//                                normalised time) and counting-sort by source-pixel tile
//   K1  k_vote      one workgroup per SEGMENT (<= 2040 consecutive sorted events = a few neighbouring
//                   source tiles): warp, bounding box of the targets, votes accumulated in an LDS
//                   window as 12.20 fixed point with ds_add_u32 (fp32 LDS atomics run at 1/12 of the
//                   integer rate on gfx950, profiles/r01_microbench.txt), one coalesced global
//                   atomicAdd per touched window pixel
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
constexpr int kEPT = 8;                       // events per thread (cached in registers between the two phases)
constexpr int kSegMax = 2040;                 // events per segment: |sum of votes| < 2040 * 2^20 < 2^31
constexpr int kWinCap = 8192;                 // LDS window capacity in 32-bit words (32 KiB)
constexpr int kWinMaxW = 128;                 // widest window when the bounding box has to be clipped
constexpr float kFix = 1048576.f;             // 2^20: votes are accumulated as signed 12.20 fixed point
constexpr float kInvFix = 1.f / 1048576.f;

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
    int *d_tile_start = nullptr;  // [ntiles + 1] first sorted event of every source tile
    int2 *d_segs = nullptr;       // [nseg] (begin, count) work items of the event kernels
    int nseg = 0, seg_cap = 0;
    // images
    float *imgs = nullptr;                                  // [5, Hp, Wp] raw votes: one per reference time + un-warped
    float *iweb[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // blurred copies
    float *G = nullptr, *Gt = nullptr;
    const float *last_iwe[4] = {nullptr, nullptr, nullptr, nullptr};
    // device scalars
    double *d_tmm = nullptr;  // [2]
    double *d_part = nullptr;   // [5 slots][256 workgroups][2] contrast-statistics partials (slot 4 = un-warped image)
    double *d_gpart = nullptr;  // [4 reference times][nseg][2] per-segment 2-DoF gradient partials
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

// first sorted event of every source tile (offsets[] holds one entry per source PIXEL key)
__global__ void __launch_bounds__(256) k_tile_starts(const int *__restrict__ offsets, int ntiles, int *__restrict__ tile_start) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t <= ntiles) tile_start[t] = offsets[t * (kTile * kTile)];
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

// Workgroup -> segment, XCD-aware: the dispatcher places block b on XCD b % 8, so giving XCD x the
// contiguous range [x * per, (x+1) * per) keeps neighbouring tiles (shared halo rows of the IWE / G
// windows, neighbouring flow pixels) in one XCD's L2.  Placement only affects speed.
__device__ __forceinline__ int segment_of_block(int nseg) {
    const int per = (nseg + 7) >> 3;
    return (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
}

struct Window {
    int r0, c0, h, w;  // top-left corner in the padded image, extent (clipped to the image and to LDS)
};

// Phase A of both event kernels: warp this thread's <= kEPT events (kept in registers), reduce the
// bounding box of their 2x2 vote footprints over the workgroup and derive the LDS window.
// rc[j] packs (row + 16384) << 16 | (col + 16384); 0 marks an empty slot.
template <int MODEL, bool FRAC, bool WANT_DT>
__device__ __forceinline__ Window phase_warp(const EvView &ev, const WarpParams &wp, int2 sg, unsigned (&rc)[kEPT],
                                              float (&fa)[kEPT], float (&fb)[kEPT], float (&fdt)[kEPT], int (&fsrc)[kEPT],
                                              int *s_box) {
    const float tscale = time_scale(wp);
    float th0 = 0.f, th1 = 0.f;
    if (MODEL == CMAX_MODEL_2DOF) {
        th0 = wp.motion[0];
        th1 = wp.motion[1];
    }
    if (threadIdx.x == 0) {
        s_box[0] = 0x7fffffff;  // min row
        s_box[1] = -0x7fffffff; // max row
        s_box[2] = 0x7fffffff;  // min col
        s_box[3] = -0x7fffffff; // max col
    }
    int mnr = 0x7fffffff, mxr = -0x7fffffff, mnc = 0x7fffffff, mxc = -0x7fffffff;
#pragma unroll
    for (int j = 0; j < kEPT; ++j) {
        const int i = (int)threadIdx.x + j * 256;
        rc[j] = 0u;
        if (i < sg.y) {
            const Warped w = warp_one<MODEL, FRAC>(ev, (int64_t)sg.x + i, wp, tscale, th0, th1);
            rc[j] = ((unsigned)(w.row + 16384) << 16) | (unsigned)(w.col + 16384);
            fa[j] = w.a;
            fb[j] = w.b;
            if (WANT_DT) {
                fdt[j] = w.dt;
                fsrc[j] = w.src;
            }
            mnr = min(mnr, w.row);
            mxr = max(mxr, w.row);
            mnc = min(mnc, w.col);
            mxc = max(mxc, w.col);
        }
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        mnr = min(mnr, __shfl_xor(mnr, o, kWave));
        mxr = max(mxr, __shfl_xor(mxr, o, kWave));
        mnc = min(mnc, __shfl_xor(mnc, o, kWave));
        mxc = max(mxc, __shfl_xor(mxc, o, kWave));
    }
    __syncthreads();  // s_box initialised
    if ((threadIdx.x & (kWave - 1)) == 0) {
        atomicMin(&s_box[0], mnr);
        atomicMax(&s_box[1], mxr);
        atomicMin(&s_box[2], mnc);
        atomicMax(&s_box[3], mxc);
    }
    __syncthreads();
    Window win;
    // footprint rows [min, max + 1], clipped to the image
    int r0 = max(s_box[0], 0), r1 = min(s_box[1] + 2, wp.Hp);
    int c0 = max(s_box[2], 0), c1 = min(s_box[3] + 2, wp.Wp);
    int h = max(r1 - r0, 0), w = max(c1 - c0, 0);
    if (h * w > kWinCap) {  // rare (very large displacements): keep the centre, the rest goes to global atomics
        if (w > kWinMaxW) {
            c0 += (w - kWinMaxW) / 2;
            w = kWinMaxW;
        }
        const int hmax = kWinCap / w;
        if (h > hmax) {
            r0 += (h - hmax) / 2;
            h = hmax;
        }
    }
    win.r0 = r0;
    win.c0 = c0;
    win.h = h;
    win.w = w;
    return win;
}

// ---------------------------------------------------------------------------------------------
// K1: warp + bilinear vote.  LDS window in signed 12.20 fixed point (ds_add_u32), coalesced flush.
// ---------------------------------------------------------------------------------------------
template <int MODEL, bool FRAC>
__global__ void __launch_bounds__(256) k_vote(EvView ev, WarpParams wp, const int2 *__restrict__ segs, int nseg,
                                              float *__restrict__ iwe) {
    __shared__ int s_win[kWinCap];
    __shared__ int s_box[4];
    const int sidx = segment_of_block(nseg);
    if (sidx >= nseg) return;
    const int2 sg = segs[sidx];
    unsigned rc[kEPT];
    float fa[kEPT], fb[kEPT], fdt[kEPT];
    int fsrc[kEPT];
    const Window win = phase_warp<MODEL, FRAC, false>(ev, wp, sg, rc, fa, fb, fdt, fsrc, s_box);
    const int wn = win.h * win.w;
    for (int i = threadIdx.x; i < wn; i += 256) s_win[i] = 0;
    __syncthreads();
    // phase B: 4 votes per event
#pragma unroll
    for (int j = 0; j < kEPT; ++j) {
        if (rc[j] == 0u) continue;
        const int row = (int)(rc[j] >> 16) - 16384, col = (int)(rc[j] & 0xFFFFu) - 16384;
        const float a = fa[j], b = fb[j], na = 1.f - a, nb = 1.f - b;
        const float wv[4] = {na * nb, a * nb, na * b, a * b};  // w_pos0..3, event_image_converter.py:365-368
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = row + (q & 1), c = col + (q >> 1);
            const int lr = r - win.r0, lc = c - win.c0;
            if ((unsigned)lr < (unsigned)win.h && (unsigned)lc < (unsigned)win.w) {
                atomicAdd(&s_win[lr * win.w + lc], __float2int_rn(wv[q] * kFix));
            } else if ((unsigned)r < (unsigned)wp.Hp && (unsigned)c < (unsigned)wp.Wp) {
                atomic_add(&iwe[(int64_t)r * wp.Wp + c], wv[q]);  // outside the LDS window, inside the image
            }
        }
    }
    __syncthreads();
    // flush: one coalesced global atomic per touched window pixel
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (win.w >= 48) {
        for (int r = wave; r < win.h; r += 4) {
            float *dst = iwe + (int64_t)(win.r0 + r) * wp.Wp + win.c0;
            for (int c = lane; c < win.w; c += kWave) {
                const int v = s_win[r * win.w + c];
                if (v != 0) atomic_add(&dst[c], (float)v * kInvFix);
            }
        }
    } else {
        for (int i = threadIdx.x; i < wn; i += 256) {
            const int v = s_win[i];
            if (v != 0) {
                const int r = i / win.w, c = i - r * win.w;
                atomic_add(&iwe[(int64_t)(win.r0 + r) * wp.Wp + win.c0 + c], (float)v * kInvFix);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K2: contrast statistics of one image.  No atomics: workgroup b writes its partial sums
//     part[2b] = sum x (variance) or sum gx^2+gy^2 (grad-mag), part[2b+1] = sum x^2; consumers add
//     the <= kStatBlocksMax partials themselves (same-address fp64 atomics serialise at ~12 ns each).
// ---------------------------------------------------------------------------------------------
constexpr int kStatBlocksMax = 256;
constexpr int kStatSlots = 5;  // reference times 0..3, slot 4 = un-warped image

template <int COST>
__global__ void __launch_bounds__(256) k_stats(const float *__restrict__ img, int H, int W, int omit, double *__restrict__ part) {
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
        part[2 * blockIdx.x] = v[0];
        part[2 * blockIdx.x + 1] = v[1];
    }
}

struct ObjParams {
    int cost, normalized, minimize, negate, omit, n_ref;
    double mult[4];
    int H, W;      // padded image
    int nblk;      // workgroups of k_stats (partials per slot)
};

// Sum the partials of every slot the objective uses into LDS: s_acc[2*slot + {0,1}].
// Called by all threads of the workgroup (contains barriers).
__device__ __forceinline__ void load_stats(const ObjParams &op, const double *__restrict__ part, double *s_acc) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, nw = blockDim.x / kWave;
    for (int slot = wave; slot < kStatSlots; slot += nw) {
        const bool used = slot < op.n_ref || (slot == 4 && op.normalized);
        double a0 = 0.0, a1 = 0.0;
        if (used) {
            const double *p = part + (int64_t)slot * 2 * kStatBlocksMax;
            for (int b = lane; b < op.nblk; b += kWave) {
                a0 += p[2 * b];
                a1 += p[2 * b + 1];
            }
        }
        a0 = wave_sum(a0);
        a1 = wave_sum(a1);
        if (lane == 0) {
            s_acc[2 * slot] = a0;
            s_acc[2 * slot + 1] = a1;
        }
    }
    __syncthreads();
}

// raw contrast from the summed accumulators (variance: unbiased like torch.var, image_variance.py:55)
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

__device__ __forceinline__ double orig_value(const ObjParams &op, const double *s_acc) {
    // orig_iwe is NOT boundary-cropped for the variance (normalized_image_variance.py:40-41)
    const int omit_o = op.cost == CMAX_COST_VARIANCE ? 0 : op.omit;
    return contrast_value(op.cost, s_acc + 8, region_pixels(op.H, op.W, omit_o), nullptr);
}

// dL/dv_k: chain factor of reference time k, and the mean of its image (variance)
__device__ __forceinline__ double chain_coef(const ObjParams &op, const double *s_acc, int k, double *mu_out) {
    const double npix = region_pixels(op.H, op.W, op.omit);
    const double v = contrast_value(op.cost, s_acc + 2 * k, npix, mu_out);
    double coef;
    if (!op.normalized) coef = op.mult[k] * (op.minimize ? -1.0 : 1.0);
    else {
        const double v_orig = orig_value(op, s_acc);
        coef = op.mult[k] * (op.minimize ? -v_orig / (v * v) : 1.0 / v_orig);
    }
    return op.negate ? -coef : coef;
}

// K2b: G[p] = dL/dv_k * dv_k/dI[p]   (needed when a blur transpose follows or for the grad-mag cost;
//      the plain-variance gradient is folded into K3 instead)
template <int COST>
__global__ void __launch_bounds__(256)
k_gimage(const float *__restrict__ img, ObjParams op, int k, const double *__restrict__ part, float *__restrict__ G) {
    __shared__ double s_acc[2 * kStatSlots];
    load_stats(op, part, s_acc);
    const int H = op.H, W = op.W;
    const int i0 = op.omit ? 1 : 0;
    const double npix = region_pixels(H, W, op.omit);
    double mu = 0.0;
    const double coef = chain_coef(op, s_acc, k, &mu);
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
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
// K3: per-event gradient, one workgroup per segment (same work list and window as K1).
//     The dL/dIWE window is staged in LDS; per event
//        dL/dx' = (1-b)(G10-G00) + b(G11-G01),  dL/dy' = (1-a)(G01-G00) + a(G11-G10)
//     2-DoF : per-workgroup fp64 partial of sum dt*(gx, gy) -> gpart[block] (summed by k_finish)
//     dense : -dt*g reduced over runs of equal source pixel inside the wave (events are sorted by
//             pixel), one fp32 atomic per run and channel
//     FOLD  : G is not materialised: G = c2 * (IWE - mu) on the cropped interior (variance, no blur)
// ---------------------------------------------------------------------------------------------
template <int MODEL, bool FRAC, bool FOLD>
__global__ void __launch_bounds__(256)
k_grad(EvView ev, WarpParams wp, const int2 *__restrict__ segs, int nseg, const float *__restrict__ img, ObjParams op, int k,
       const double *__restrict__ part, double *__restrict__ gpart, float *__restrict__ gflow) {
    __shared__ float s_win[kWinCap];
    __shared__ int s_box[4];
    __shared__ double s_acc[2 * kStatSlots];
    __shared__ double s_red[2 * 4];
    const int sidx = segment_of_block(nseg);
    if (sidx >= nseg) return;
    const int2 sg = segs[sidx];
    float c2 = 0.f, mu = 0.f;
    if (FOLD) {
        load_stats(op, part, s_acc);
        double mud = 0.0;
        const double coef = chain_coef(op, s_acc, k, &mud);
        c2 = (float)(coef * 2.0 / (region_pixels(op.H, op.W, op.omit) - 1.0));
        mu = (float)mud;
    }
    const int i0 = op.omit ? 1 : 0;
    auto g_at = [&](int r, int c) -> float {  // dL/dIWE at an in-image pixel
        const float x = img[(int64_t)r * wp.Wp + c];
        if (!FOLD) return x;
        const bool in = (r >= i0) && (r < wp.Hp - i0) && (c >= i0) && (c < wp.Wp - i0);
        return in ? c2 * (x - mu) : 0.f;
    };
    unsigned rc[kEPT];
    float fa[kEPT], fb[kEPT], fdt[kEPT];
    int fsrc[kEPT];
    const Window win = phase_warp<MODEL, FRAC, true>(ev, wp, sg, rc, fa, fb, fdt, fsrc, s_box);
    const int wn = win.h * win.w;
    for (int i = threadIdx.x; i < wn; i += 256) {
        const int r = i / win.w, c = i - r * win.w;
        s_win[i] = g_at(win.r0 + r, win.c0 + c);
    }
    __syncthreads();
    const int hw = wp.H * wp.W;
    const int lane = threadIdx.x & (kWave - 1);
    double acc[2] = {0.0, 0.0};
#pragma unroll
    for (int j = 0; j < kEPT; ++j) {
        float gx = 0.f, gy = 0.f, dt = 0.f;
        int key = -1 - lane;  // unique per lane: empty slots never merge
        if (rc[j] != 0u) {
            const int row = (int)(rc[j] >> 16) - 16384, col = (int)(rc[j] & 0xFFFFu) - 16384;
            float g[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = row + (q & 1), c = col + (q >> 1);
                const int lr = r - win.r0, lc = c - win.c0;
                if ((unsigned)lr < (unsigned)win.h && (unsigned)lc < (unsigned)win.w) g[q] = s_win[lr * win.w + lc];
                else if ((unsigned)r < (unsigned)wp.Hp && (unsigned)c < (unsigned)wp.Wp) g[q] = g_at(r, c);
                else g[q] = 0.f;  // corner outside the image: masked vote, zero gradient
            }
            // g[0] = G00 (row, col), g[1] = G10 (row+1, col), g[2] = G01 (row, col+1), g[3] = G11
            const float a = fa[j], b = fb[j];
            gx = (1.f - b) * (g[1] - g[0]) + b * (g[3] - g[2]);
            gy = (1.f - a) * (g[2] - g[0]) + a * (g[3] - g[1]);
            dt = fdt[j];
            key = fsrc[j];
        }
        if (MODEL == CMAX_MODEL_2DOF) {
            acc[0] += (double)(dt * gx);
            acc[1] += (double)(dt * gy);
        } else {
            // segmented inclusive scan over the wave: lanes hold consecutive sorted events, a run =
            // adjacent lanes with the same key (voxel keys of one pixel interleave time bins, so the
            // scan carries head flags instead of comparing keys at a distance)
            float vx = -dt * gx, vy = -dt * gy;
            const int kprev = __shfl_up(key, 1, kWave);
            int head = (lane == 0 || kprev != key) ? 1 : 0;
            const int hnext = __shfl_down(head, 1, kWave);  // evaluated by all lanes (no short-circuit around a shuffle)
            const int tail = (lane == kWave - 1) || (hnext != 0);
#pragma unroll
            for (int o = 1; o < kWave; o <<= 1) {
                const int h2 = __shfl_up(head, o, kWave);
                const float x2 = __shfl_up(vx, o, kWave), y2 = __shfl_up(vy, o, kWave);
                if (lane >= o && !head) {
                    vx += x2;
                    vy += y2;
                    head |= h2;
                }
            }
            if (key >= 0 && tail) {  // last lane of its run holds the run's sum
                atomic_add(&gflow[key], vx);
                atomic_add(&gflow[key + hw], vy);
            }
        }
    }
    if (MODEL == CMAX_MODEL_2DOF) {
        block_sum<2>(acc, s_red);
        if (threadIdx.x == 0) {
            gpart[2 * sidx] = acc[0];
            gpart[2 * sidx + 1] = acc[1];
        }
    }
}

// Last kernel of an evaluation (one workgroup): loss and per-slot values from the statistics
// partials; 2-DoF gradient = sum of the per-segment partials of every K3 launch.
__global__ void __launch_bounds__(256)
k_finish(ObjParams op, const double *__restrict__ part, const double *__restrict__ gpart, int n_gpart,
         double *__restrict__ result, double *__restrict__ gtheta) {
    __shared__ double s_acc[2 * kStatSlots];
    __shared__ double s_red[2 * 4];
    load_stats(op, part, s_acc);
    if (threadIdx.x == 0) {
        const double npix = region_pixels(op.H, op.W, op.omit);
        const double v_orig = op.normalized ? orig_value(op, s_acc) : 0.0;
        double loss = 0.0;
        for (int k = 0; k < op.n_ref; ++k) {
            const double v = contrast_value(op.cost, s_acc + 2 * k, npix, nullptr);
            result[1 + k] = v;
            if (!op.normalized) loss += op.mult[k] * (op.minimize ? -v : v);
            else loss += op.mult[k] * (op.minimize ? v_orig / v : v / v_orig);
        }
        result[0] = op.negate ? -loss : loss;
        result[5] = v_orig;
    }
    if (gtheta) {
        double acc[2] = {0.0, 0.0};
        for (int i = threadIdx.x; i < n_gpart; i += blockDim.x) {
            acc[0] += gpart[2 * i];
            acc[1] += gpart[2 * i + 1];
        }
        block_sum<2>(acc, s_red);
        if (threadIdx.x == 0) {
            gtheta[0] = acc[0];
            gtheta[1] = acc[1];
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
    const int grid = 8 * ((h->nseg + 7) / 8);
    ProfScope prof(h, kProfVote, s);
    if (h->has_frac) hipLaunchKernelGGL((k_vote<MODEL, true>), dim3(grid), dim3(256), 0, s, ev, wp, h->d_segs, h->nseg, img);
    else hipLaunchKernelGGL((k_vote<MODEL, false>), dim3(grid), dim3(256), 0, s, ev, wp, h->d_segs, h->nseg, img);
}

template <int MODEL>
static void launch_grad(cmax_handle_s *h, const EvView &ev, const WarpParams &wp, const float *img, bool fold,
                        const ObjParams &op, int k, double *gpart, float *gflow, hipStream_t s) {
    const int grid = 8 * ((h->nseg + 7) / 8);
    ProfScope prof(h, kProfGrad, s);
#define CMAX_LAUNCH_GRAD(FRAC, FOLD) \
    hipLaunchKernelGGL((k_grad<MODEL, FRAC, FOLD>), dim3(grid), dim3(256), 0, s, ev, wp, h->d_segs, h->nseg, img, op, k, h->d_part, gpart, gflow)
    if (h->has_frac) {
        if (fold) CMAX_LAUNCH_GRAD(true, true);
        else CMAX_LAUNCH_GRAD(true, false);
    } else {
        if (fold) CMAX_LAUNCH_GRAD(false, true);
        else CMAX_LAUNCH_GRAD(false, false);
    }
#undef CMAX_LAUNCH_GRAD
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

static int stat_blocks(const cmax_handle_s *h) {
    // ~8 pixels per thread, between 32 and kStatBlocksMax workgroups
    int64_t b = ((int64_t)h->Hp * h->Wp + 2047) / 2048;
    if (b < 32) b = 32;
    if (b > kStatBlocksMax) b = kStatBlocksMax;
    return (int)b;
}

static int launch_stats(cmax_handle_s *h, int cost, const float *img, int Hp, int Wp, int omit, int slot, hipStream_t s) {
    const int grid = stat_blocks(h);
    double *part = h->d_part + (int64_t)slot * 2 * kStatBlocksMax;
    ProfScope prof(h, kProfStats, s);
    if (cost == CMAX_COST_VARIANCE) hipLaunchKernelGGL(k_stats<CMAX_COST_VARIANCE>, dim3(grid), dim3(256), 0, s, img, Hp, Wp, omit, part);
    else hipLaunchKernelGGL(k_stats<CMAX_COST_GRADMAG>, dim3(grid), dim3(256), 0, s, img, Hp, Wp, omit, part);
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
    if (!rc) rc = dev_alloc(h, &h->d_part, kStatSlots * kStatBlocksMax * 2);
    if (!rc) rc = dev_alloc(h, &h->counts, h->nkeys + 1);
    if (!rc) rc = dev_alloc(h, &h->cursor, h->nkeys);
    if (!rc) rc = dev_alloc(h, &h->d_flags, 2);
    if (!rc) rc = dev_alloc(h, &h->d_tile_start, h->ntr * h->ntc + 1);
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
    dev_free(&h->d_part);
    dev_free(&h->d_gpart);
    dev_free(&h->counts);
    dev_free(&h->cursor);
    dev_free(&h->d_flags);
    dev_free(&h->d_tile_start);
    dev_free(&h->d_segs);
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
        h->nseg = 0;
        return 0;
    }
    const int grid = stream_grid(n, 256);
    if (dtype == CMAX_F32) hipLaunchKernelGGL(k_pack_hist<float>, dim3(grid), dim3(256), 0, s, (const float *)events, n, h->H, h->W, h->ntc, h->key_tmp, h->counts, h->d_flags);
    else hipLaunchKernelGGL(k_pack_hist<double>, dim3(grid), dim3(256), 0, s, (const double *)events, n, h->H, h->W, h->ntc, h->key_tmp, h->counts, h->d_flags);
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, h->counts, h->nkeys);
    if (dtype == CMAX_F32) hipLaunchKernelGGL(k_scatter<float>, dim3(grid), dim3(256), 0, s, (const float *)events, n, h->key_tmp, h->counts, h->cursor, h->d_tmm, n_time_bin, h->xyb, h->tau, h->rx, h->ry, h->tau64);
    else hipLaunchKernelGGL(k_scatter<double>, dim3(grid), dim3(256), 0, s, (const double *)events, n, h->key_tmp, h->counts, h->cursor, h->d_tmm, n_time_bin, h->xyb, h->tau, h->rx, h->ry, h->tau64);
    CMAX_CHECK_LAUNCH();
    // once per batch: how many events survived, whether any source coordinate is fractional, and the
    // per-tile event ranges from which the segment work list is cut
    const int ntiles = h->ntr * h->ntc;
    hipLaunchKernelGGL(k_tile_starts, dim3(div_up(ntiles + 1, 256)), dim3(256), 0, s, h->counts, ntiles, h->d_tile_start);
    CMAX_CHECK_LAUNCH();
    int flags[2] = {0, 0};
    std::vector<int> tile_start((size_t)ntiles + 1);
    CMAX_CHECK_HIP(hipMemcpyAsync(flags, h->d_flags, sizeof(flags), hipMemcpyDeviceToHost, s));
    CMAX_CHECK_HIP(hipMemcpyAsync(tile_start.data(), h->d_tile_start, tile_start.size() * sizeof(int), hipMemcpyDeviceToHost, s));
    CMAX_CHECK_HIP(hipStreamSynchronize(s));
    h->has_frac = flags[0] != 0;
    h->n = n - flags[1];
    // Segments: consecutive tiles of one tile row are merged while they fit (sparse batches), a dense
    // tile is split into several segments; never more than kSegMax events (fixed-point range).
    std::vector<int2> segs;
    int begin = 0, count = 0, row_of_begin = -1;
    auto close = [&]() {
        if (count > 0) segs.push_back(make_int2(begin, count));
        count = 0;
    };
    for (int t = 0; t < ntiles; ++t) {
        int b = tile_start[t], c = tile_start[t + 1] - tile_start[t];
        const int trow = t / h->ntc;
        if (c == 0) continue;
        if (count > 0 && (trow != row_of_begin || count + c > kSegMax)) close();
        while (c > 0) {
            if (count == 0) {
                begin = b;
                row_of_begin = trow;
            }
            const int take = c < kSegMax - count ? c : kSegMax - count;
            count += take;
            b += take;
            c -= take;
            if (count == kSegMax) close();
        }
    }
    close();
    h->nseg = (int)segs.size();
    if (h->nseg > h->seg_cap) {
        dev_free(&h->d_segs);
        dev_free(&h->d_gpart);
        int rc = dev_alloc(h, &h->d_segs, h->nseg);
        if (!rc) rc = dev_alloc(h, &h->d_gpart, (int64_t)4 * h->nseg * 2);
        if (rc) return rc;
        h->seg_cap = h->nseg;
    }
    if (h->nseg > 0) {
        CMAX_CHECK_HIP(hipMemcpyAsync(h->d_segs, segs.data(), segs.size() * sizeof(int2), hipMemcpyHostToDevice, s));
        CMAX_CHECK_HIP(hipStreamSynchronize(s));  // `segs` is a host temporary
    }
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
    op.nblk = stat_blocks(h);

    // statistics of the un-warped image (slot 4) are cached per batch
    if (d->normalized) {
        if (n_images == d->n_ref + 1) {
            const float *img = nullptr;
            rc = blur_image(h, d->sigma, images + (int64_t)d->n_ref * npix, h->iweb[4], &img, s);
            if (rc) return rc;
            // orig_iwe is NOT boundary-cropped for the variance (normalized_image_variance.py:40-41)
            const int omit_o = d->cost == CMAX_COST_VARIANCE ? 0 : d->omit_boundary;
            rc = launch_stats(h, d->cost, img, Hp, Wp, omit_o, 4, s);
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
        rc = launch_stats(h, d->cost, img, Hp, Wp, d->omit_boundary, k, s);
        if (rc) return rc;
    }

    const bool two_dof = d->model == CMAX_MODEL_2DOF;
    int n_gpart = 0;
    if (grad && h->n > 0) {
        // backward: dL/dIWE (folded into K3 for the plain variance; otherwise G image + blur transpose)
        // and the per-event gather, accumulated over the reference times
        if (!two_dof) CMAX_CHECK_HIP(hipMemsetAsync(grad, 0, gbytes, s));
        const EvView ev = ev_view(h);
        const bool fold = d->cost == CMAX_COST_VARIANCE && !(d->sigma > 0);
        double k0 = 0, k1 = 0;
        if (d->sigma > 0) blur_taps(d->sigma, k0, k1);
        for (int k = 0; k < d->n_ref; ++k) {
            const float *gsrc = h->last_iwe[k];
            if (!fold) {
                float *Gk = d->sigma > 0 ? h->Gt : h->G;
                {
                    ProfScope prof(h, kProfGimage, s);
                    if (d->cost == CMAX_COST_VARIANCE)
                        hipLaunchKernelGGL(k_gimage<CMAX_COST_VARIANCE>, dim3(div_up(npix, 256)), dim3(256), 0, s, h->last_iwe[k], op, k, h->d_part, Gk);
                    else
                        hipLaunchKernelGGL(k_gimage<CMAX_COST_GRADMAG>, dim3(div_up(npix, 256)), dim3(256), 0, s, h->last_iwe[k], op, k, h->d_part, Gk);
                }
                if (d->sigma > 0)
                    hipLaunchKernelGGL(k_blur3_adj<float>, dim3(div_up(npix, 256)), dim3(256), 0, s, Gk, Hp, Wp, (float)k0, (float)k1, h->G);
                CMAX_CHECK_LAUNCH();
                gsrc = h->G;
            }
            const WarpParams wp = warp_params(h, motion, d->T, d->ref_mode[k], d->ref_frac[k], d->normalize_t);
            double *gpart = h->d_gpart + (int64_t)k * h->nseg * 2;
            switch (d->model) {
                case CMAX_MODEL_2DOF: launch_grad<CMAX_MODEL_2DOF>(h, ev, wp, gsrc, fold, op, k, gpart, nullptr, s); break;
                case CMAX_MODEL_DENSE: launch_grad<CMAX_MODEL_DENSE>(h, ev, wp, gsrc, fold, op, k, nullptr, (float *)grad, s); break;
                default: launch_grad<CMAX_MODEL_VOXEL>(h, ev, wp, gsrc, fold, op, k, nullptr, (float *)grad, s); break;
            }
            CMAX_CHECK_LAUNCH();
        }
        n_gpart = two_dof ? d->n_ref * h->nseg : 0;
    } else if (grad) {
        CMAX_CHECK_HIP(hipMemsetAsync(grad, 0, gbytes, s));  // this rank holds no events of the batch
    }
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(256), 0, s, op, h->d_part, h->d_gpart, n_gpart, result,
                       (grad && two_dof && h->n > 0) ? (double *)grad : nullptr);
    CMAX_CHECK_LAUNCH();
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
