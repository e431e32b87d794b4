// Per-batch preparation: pack the events and order them for the event kernels.
//
// Order: source tile (16 x 16 pixels) major; inside a tile by pixel (un-binned handle -- the dense model's run
// reduction wants equal pixels adjacent) or by time bin (binned handle -- the voxel model's LDS accumulators want
// few bins per workgroup).  Sensor events arrive time-sorted, i.e. random in space, so a counting sort on the full
// key costs one scattered global atomic per event and pass (1M events: 101 us histogram + 70 us scatter).  Two
// levels instead:
//   S1 k_bucket_hist     4096 events per workgroup, tile histogram in LDS, one global atomic per (workgroup, tile)
//   S2 scan of the tile counts (the caller's scan kernels)
//   S3 k_bucket_scatter  same chunks: a range per (workgroup, tile) is reserved with one returning global atomic, the
//                        slot inside it comes from an LDS atomic; events land in their tile's bucket of a staging
//                        SoA in runs of ~chunk / tiles elements
//   S4 k_tile_sort       one workgroup per tile: LDS counting sort of the bucket by pixel-in-tile / time bin into
//                        the final SoA; writes the group starts (tile, or (tile, bin)) and counts the active pixels
// Sensors with more than kSortLdsTiles tiles (beyond 1024 x 1024) take the same path with the per-event atomics
// on the global tile counters.
// The same pipeline re-bins a batch in place (cmax_set_time_bins): the source is then the packed SoA itself.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cmax_common.h"

namespace cmax {

constexpr int kSortThreads = 1024;   // S1 / S3: 4 events per thread, held in registers between the two LDS phases
constexpr int kSortEPT = 4;
constexpr int kSortChunk = kSortThreads * kSortEPT;  // events per workgroup in S1 / S3
constexpr int kSortLdsTiles = 4096;  // tile histogram + bases held in LDS (2 x 16 KB)
constexpr int kTileSortThreads = 512;

// t_min / t_max of the batch as order-preserving 64-bit keys, both reduced with atomicMax: [0] holds ~key(t_min),
// [1] key(t_max); all-zero = empty.  k_tile_sort turns them into the two doubles every other kernel reads.
__device__ __forceinline__ unsigned long long sort_f64_key(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double sort_f64_unkey(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

__device__ __forceinline__ int sort_voxel_bin(double tau, int T) {
    // reference edges for direction "first": e_k = k/T * (dtmax - dtmin) + dtmin with dt in [0,1]
    // (src/warp.py:342-345); the event belongs to the last k with e_k <= dt.
    int k = (int)(tau * (double)T);
    if (k > T - 1) k = T - 1;
    if (k < 0) k = 0;
    while (k > 0 && ((double)k / (double)T) > tau) --k;
    while (k + 1 < T && ((double)(k + 1) / (double)T) <= tau) ++k;
    return k;
}

// The 8 bits above the source pixel of an UN-BINNED handle's packed word hold the residual of the normalised time that fp32
// cannot: tau = (double)tau32 + q * ulp(tau32) / 256, q in [-128, 127] -- 32 significant bits, enough to decide on which
// side of a cell border an event lies exactly like the reference's fp64 arithmetic does (see warp_one).  Binned handles keep
// the voxel time bin there.
__device__ __forceinline__ uint32_t tau_residual8(double tn) {
    const float hi = (float)tn;
    const uint32_t eb = __float_as_uint(hi) & 0x7F800000u;
    if (eb < (32u << 23)) return 0u;  // |tau| < 2^-95: nothing to refine
    const double unit = (double)__uint_as_float(eb - (31u << 23));  // ulp(hi) / 256
    int q = (int)rint((tn - (double)hi) / unit);
    q = q < -128 ? -128 : (q > 127 ? 127 : q);
    return (uint32_t)q & 0xFFu;
}
__device__ __forceinline__ double tau_refined(uint2 e) {
    const uint32_t eb = e.y & 0x7F800000u;
    const int q = (int)e.x >> 24;  // sign-extended
    const double hi = (double)__uint_as_float(e.y);
    return eb < (32u << 23) ? hi : hi + (double)q * (double)__uint_as_float(eb - (31u << 23));
}

// One event as the sort sees it.
struct SortItem {
    int ix, iy;     // source pixel; ix < 0: not on the sensor (dropped)
    float rx, ry;   // fractional parts of the source coordinates
    float rxl, ryl; // ... and what fp32 could not hold of them: residual = (double)rx + (double)rxl to 2^-48 (round 5: the cells of events
                    // on a cell border are decided from the source coordinate the reference's fp64 arithmetic sees, warp_exact)
    double tn;      // time normalised to the batch
    bool frac;
    bool outside;   // kept although off the sensor (RawSource::keep_outside): (ix, iy) is the NEAREST sensor pixel, (rx, ry) the rest
};

// raw [n,4] = (x row, y column, t, p) in T (fp32 / fp64)
template <typename T>
struct RawSource {
    const T *ev;
    const double *tmm;  // (t_min, t_max) of the batch, on the device
    int H, W;
    int keyed;          // tmm still holds the keys of the reduction fused into S1 (see sort_f64_key)
    int keep_outside;   // cmax_set_keep_outside: finite events off the sensor are packed at the nearest sensor pixel + a residual
    __device__ __forceinline__ bool reduces_time() const { return keyed != 0; }
    __device__ __forceinline__ double time(int64_t i) const { return (double)ev[4 * i + 2]; }
    __device__ __forceinline__ SortItem pixel(int64_t i) const {  // S1: the pixel only
        return classify(ev[4 * i + 0], ev[4 * i + 1]);
    }
    __device__ __forceinline__ SortItem classify(T x, T y) const {
        SortItem it;
        const T fx = floor_t<T>(x), fy = floor_t<T>(y);
        it.ix = -1;
        it.iy = 0;
        it.frac = false;
        it.outside = false;
        if (fx >= (T)0 && fx < (T)H && fy >= (T)0 && fy < (T)W) {  // NaN fails every comparison -> dropped
            it.ix = (int)fx;
            it.iy = (int)fy;
            it.frac = x != fx || y != fy;
        } else if (keep_outside && fx > (T)-1048576 && fx < (T)1048576 && fy > (T)-1048576 && fy < (T)1048576) {  // (finite, sane)
            it.ix = fx < (T)0 ? 0 : (fx >= (T)H ? H - 1 : (int)fx);
            it.iy = fy < (T)0 ? 0 : (fy >= (T)W ? W - 1 : (int)fy);
            it.frac = true;
            it.outside = true;
        }
        return it;
    }
    // the loads of full() on their own, and the arithmetic behind them: the radix sort issues the loads of its NEXT step before it
    // works on the current one (cmax_radix_sort.h)
    struct Fetched {
        T x, y, t;
    };
    __device__ __forceinline__ Fetched fetch(int64_t i) const { return Fetched{ev[4 * i + 0], ev[4 * i + 1], ev[4 * i + 2]}; }
    __device__ __forceinline__ SortItem resolve(const Fetched &f, int64_t) const {
        SortItem it = classify(f.x, f.y);
        if (it.ix >= 0) {
            const double rxd = (double)f.x - (double)it.ix, ryd = (double)f.y - (double)it.iy;
            it.rx = (float)rxd;
            it.ry = (float)ryd;
            it.rxl = (float)(rxd - (double)it.rx);
            it.ryl = (float)(ryd - (double)it.ry);
            double tmin = tmm[0], tmax = tmm[1];
            if (keyed) {
                const unsigned long long *k = reinterpret_cast<const unsigned long long *>(tmm);
                tmin = sort_f64_unkey(~k[0]);
                tmax = sort_f64_unkey(k[1]);
            }
            const double per = tmax - tmin;
            it.tn = per > 0 ? ((double)f.t - tmin) / per : 0.0;
        }
        return it;
    }
    __device__ __forceinline__ SortItem full(int64_t i) const {
        const T x = ev[4 * i + 0], y = ev[4 * i + 1];
        SortItem it = classify(x, y);
        if (it.ix >= 0) {
            const double rxd = (double)x - (double)it.ix, ryd = (double)y - (double)it.iy;  // exact (on the sensor: ix = floor(x))
            it.rx = (float)rxd;
            it.ry = (float)ryd;
            it.rxl = (float)(rxd - (double)it.rx);
            it.ryl = (float)(ryd - (double)it.ry);
            double tmin = tmm[0], tmax = tmm[1];
            if (keyed) {
                const unsigned long long *k = reinterpret_cast<const unsigned long long *>(tmm);
                tmin = sort_f64_unkey(~k[0]);
                tmax = sort_f64_unkey(k[1]);
            }
            const double per = tmax - tmin;
            it.tn = per > 0 ? ((double)ev[4 * i + 2] - tmin) / per : 0.0;
        }
        return it;
    }
};

// the handle's own packed SoA (re-binning)
struct PackedSource {
    const uint2 *evp;
    const float *rx, *ry;  // written only for batches with fractional source coordinates (has_frac)
    const float2 *rl;      // (rxl, ryl), likewise
    const double *tau64;
    int has_frac;
    __device__ __forceinline__ bool reduces_time() const { return false; }
    __device__ __forceinline__ double time(int64_t) const { return 0.0; }
    __device__ __forceinline__ SortItem pixel(int64_t i) const {
        SortItem it;
        const uint32_t pk = evp[i].x;
        it.ix = (int)(pk & 0xFFFu);
        it.iy = (int)((pk >> 12) & 0xFFFu);
        it.frac = false;
        it.outside = false;
        return it;
    }
    struct Fetched {
        uint32_t pk;
        float rx, ry;
        float2 lo;
        double tn;
    };
    __device__ __forceinline__ Fetched fetch(int64_t i) const {
        Fetched f;
        f.pk = evp[i].x;
        f.rx = has_frac ? rx[i] : 0.f;
        f.ry = has_frac ? ry[i] : 0.f;
        f.lo = has_frac ? rl[i] : make_float2(0.f, 0.f);
        f.tn = tau64[i];
        return f;
    }
    __device__ __forceinline__ SortItem resolve(const Fetched &f, int64_t) const {
        SortItem it;
        it.ix = (int)(f.pk & 0xFFFu);
        it.iy = (int)((f.pk >> 12) & 0xFFFu);
        it.frac = false;
        it.outside = false;
        it.rx = f.rx;
        it.ry = f.ry;
        it.rxl = f.lo.x;
        it.ryl = f.lo.y;
        it.tn = f.tn;
        return it;
    }
    __device__ __forceinline__ SortItem full(int64_t i) const {
        SortItem it = pixel(i);
        it.rx = has_frac ? rx[i] : 0.f;
        it.ry = has_frac ? ry[i] : 0.f;
        const float2 lo = has_frac ? rl[i] : make_float2(0.f, 0.f);
        it.rxl = lo.x;
        it.ryl = lo.y;
        it.tn = tau64[i];
        return it;
    }
};

struct SortOut {
    uint2 *evp;
    float *rx, *ry;
    float2 *rl;  // (rxl, ryl): low parts of the fractional residuals (SortItem)
    double *tau64;
};

// S0.  Everything the sort accumulates into, cleared by one launch (four hipMemsetAsync nodes cost ~3 us each):
// tile counts [ntiles + 1], tile cursors [ntiles], flags [first_flag, 4), the keyed time extremes (optional).
__global__ void __launch_bounds__(256) k_sort_clear(int *__restrict__ tile_count, int *__restrict__ tile_cursor, int ntiles, int *__restrict__ flags,
                                                    int first_flag, unsigned long long *__restrict__ tmm_keys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= ntiles) tile_count[i] = 0;
    if (i < ntiles) tile_cursor[i] = 0;
    if (i >= first_flag && i < 4) flags[i] = 0;
    if (tmm_keys && i < 2) tmm_keys[i] = 0ull;  // "empty" for both atomicMax reductions
}

// S1.  tile_count[ntiles] += events per tile; flags[0] = any fractional source coordinate, flags[1] += dropped events,
// flags[2] += events kept off the sensor;
// tmm_keys (RawSource with keyed extremes): batch time extremes, two atomics per workgroup.
template <typename SRC>
__global__ void __launch_bounds__(kSortThreads)
k_bucket_hist(SRC src, int64_t n, int ntc, int ntiles, int *__restrict__ tile_count, int *__restrict__ flags,
              unsigned long long *__restrict__ tmm_keys) {
    __shared__ int s_hist[kSortLdsTiles];
    __shared__ double s_lo[kSortThreads / kWave], s_hi[kSortThreads / kWave];
    const bool lds = ntiles <= kSortLdsTiles;
    if (lds) {
        for (int t = threadIdx.x; t < ntiles; t += kSortThreads) s_hist[t] = 0;
        __syncthreads();
    }
    const int64_t base = (int64_t)blockIdx.x * kSortChunk;
    int dropped = 0, outside = 0;
    bool frac = false;
    double lo = INFINITY, hi = -INFINITY;
#pragma unroll
    for (int u = 0; u < kSortEPT; ++u) {
        const int64_t i = base + u * kSortThreads + threadIdx.x;
        if (i >= n) continue;
        const SortItem it = src.pixel(i);
        if (src.reduces_time()) {
            const double t = src.time(i);
            lo = fmin(lo, t);
            hi = fmax(hi, t);
        }
        if (it.ix < 0) {
            ++dropped;
            continue;
        }
        frac = frac || it.frac;
        outside += it.outside ? 1 : 0;
        const int tile = (it.ix >> 4) * ntc + (it.iy >> 4);
        if (lds) atomicAdd(&s_hist[tile], 1);
        else atomicAdd(&tile_count[tile], 1);
    }
    if (frac) flags[0] = 1;
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) dropped += __shfl_xor(dropped, o, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0 && dropped) atomicAdd(&flags[1], dropped);
    if (__any(outside != 0)) {  // (rare: batches from a sensor have none)
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) outside += __shfl_xor(outside, o, kWave);
        if ((threadIdx.x & (kWave - 1)) == 0) atomicAdd(&flags[2], outside);
    }
    if (src.reduces_time()) {
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
            lo = fmin(lo, __shfl_xor(lo, o, kWave));
            hi = fmax(hi, __shfl_xor(hi, o, kWave));
        }
        if ((threadIdx.x & (kWave - 1)) == 0) {
            s_lo[threadIdx.x / kWave] = lo;
            s_hi[threadIdx.x / kWave] = hi;
        }
    }
    __syncthreads();
    if (src.reduces_time() && threadIdx.x == 0) {
        for (int w = 1; w < kSortThreads / kWave; ++w) {
            lo = fmin(lo, s_lo[w]);
            hi = fmax(hi, s_hi[w]);
        }
        if (lo <= hi) {
            atomicMax(&tmm_keys[0], ~sort_f64_key(lo));
            atomicMax(&tmm_keys[1], sort_f64_key(hi));
        }
    }
    if (lds) {
        for (int t = threadIdx.x; t < ntiles; t += kSortThreads) {
            const int c = s_hist[t];
            if (c) atomicAdd(&tile_count[t], c);
        }
    }
}

// S3.  tile_off: exclusive scan of the tile counts; tile_cursor: zero on entry.
template <typename SRC>
__global__ void __launch_bounds__(kSortThreads)
k_bucket_scatter(SRC src, int64_t n, int ntc, int ntiles, int T, const int *__restrict__ tile_off, int *__restrict__ tile_cursor,
                 const int *__restrict__ flags, SortOut out) {
    __shared__ int s_hist[kSortLdsTiles];  // count, then running slot
    __shared__ int s_base[kSortLdsTiles];
    const bool lds = ntiles <= kSortLdsTiles;
    // flags[0] (complete: S1 wrote it): any fractional source coordinate.  Sensor batches have none; the rx / ry arrays
    // are then never read by anyone, and two of the four scattered stores per event are saved here and in S4.
    const bool frac = flags[0] != 0;
    const int64_t base = (int64_t)blockIdx.x * kSortChunk;
    if (lds) {
        for (int t = threadIdx.x; t < ntiles; t += kSortThreads) s_hist[t] = 0;
        __syncthreads();
    }
    SortItem item[kSortEPT];
    int tile[kSortEPT];
#pragma unroll
    for (int u = 0; u < kSortEPT; ++u) {
        const int64_t i = base + u * kSortThreads + threadIdx.x;
        tile[u] = -1;
        if (i < n) {
            item[u] = src.full(i);
            if (item[u].ix >= 0) tile[u] = (item[u].ix >> 4) * ntc + (item[u].iy >> 4);
        }
        if (lds && tile[u] >= 0) atomicAdd(&s_hist[tile[u]], 1);
    }
    if (lds) {
        __syncthreads();
        for (int t = threadIdx.x; t < ntiles; t += kSortThreads) {
            const int c = s_hist[t];
            if (c) s_base[t] = tile_off[t] + atomicAdd(&tile_cursor[t], c);
            s_hist[t] = 0;
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < kSortEPT; ++u) {
        if (tile[u] < 0) continue;
        const SortItem &it = item[u];
        const int pos = lds ? s_base[tile[u]] + atomicAdd(&s_hist[tile[u]], 1) : tile_off[tile[u]] + atomicAdd(&tile_cursor[tile[u]], 1);
        // top byte: the voxel time bin (binned handle) or the time residual beyond fp32 (un-binned handle, tau_residual8)
        const uint32_t bin = T > 0 ? (uint32_t)sort_voxel_bin(it.tn, T) : tau_residual8(it.tn);
        out.evp[pos] = make_uint2((uint32_t)it.ix | ((uint32_t)it.iy << 12) | (bin << 24), __float_as_uint((float)it.tn));
        if (frac) {
            out.rx[pos] = it.rx;
            out.ry[pos] = it.ry;
            out.rl[pos] = make_float2(it.rxl, it.ryl);
        }
        out.tau64[pos] = it.tn;
    }
}

// S4.  One workgroup per tile: LDS counting sort of the tile's bucket into the final SoA.  Key inside the tile:
//   un-binned handle   pixel-in-tile (256 keys)
//   binned handle      (time bin, pixel-in-tile) while T * 256 counters fit kTileKeysMax, else the time bin alone.
// With the pixel in the key the events of a (tile, bin) group come out pixel by pixel: neighbouring lanes of the event
// kernels then gather neighbouring voxel entries, and consecutive events of a thread mostly share (pixel, bin) -- they warp to
// the same cell (their times differ by less than a bin), so K1 sums their votes and the voxel K3 their gradient terms in
// registers before the LDS atomics.
// group_start: [ntiles + 1] (T == 0) or [ntiles * T + 1]; active[tile] (T == 0): source pixels of this tile that hold
// events; tmm_keys (optional): the keyed batch extremes S1 reduced, converted to the two doubles by the first workgroup
// (nothing in this launch reads them).
constexpr int kTileKeysMax = 8192;  // 32 KB of counters: (bin, pixel) keys up to T = 32
__global__ void __launch_bounds__(kTileSortThreads)
k_tile_sort(int ntiles, int T, const int *__restrict__ tile_off, SortOut in, SortOut out, int *__restrict__ group_start, int *__restrict__ active,
            const int *__restrict__ flags, unsigned long long *__restrict__ tmm_keys) {
    const bool frac = flags[0] != 0;
    __shared__ int s_cnt[kTileKeysMax];  // counts, then running cursors
    __shared__ int s_wave[kTileSortThreads / kWave];
    __shared__ int s_nz;
    const int tile = blockIdx.x;
    const int b = tile_off[tile], e = tile_off[tile + 1];
    const int t = threadIdx.x;
    if (tmm_keys && tile == 0 && t == 0) {
        const unsigned long long k0 = tmm_keys[0], k1 = tmm_keys[1];
        double *d = reinterpret_cast<double *>(tmm_keys);
        d[0] = (k0 | k1) ? sort_f64_unkey(~k0) : (double)INFINITY;
        d[1] = (k0 | k1) ? sort_f64_unkey(k1) : -(double)INFINITY;
    }
    const bool fine = T > 0 && T * 256 <= kTileKeysMax;  // (bin, pixel) key
    const int nkey = T > 0 ? (fine ? T * 256 : T) : 256;
    for (int k = t; k < nkey; k += kTileSortThreads) s_cnt[k] = 0;
    if (t == 0) s_nz = 0;
    __syncthreads();
    auto sub_key = [&](uint32_t pk) -> int {
        const int pix = (int)((((pk & 0xFFFu) & 15u) << 4) | (((pk >> 12) & 0xFFFu) & 15u));
        return T > 0 ? (fine ? (int)(pk >> 24) * 256 + pix : (int)(pk >> 24)) : pix;
    };
    for (int i = b + t; i < e; i += kTileSortThreads) atomicAdd(&s_cnt[sub_key(in.evp[i].x)], 1);
    __syncthreads();
    // exclusive scan of the nkey counters in place: thread t owns `per` consecutive counters
    const int per = (nkey + kTileSortThreads - 1) / kTileSortThreads;
    const int k0 = t * per, k1 = min(k0 + per, nkey);
    int sum = 0, nz = 0;
    for (int k = k0; k < k1; ++k) {
        sum += s_cnt[k];
        nz += s_cnt[k] != 0;
    }
    const int lane = t & (kWave - 1), wave = t / kWave;
    int incl = sum;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const int v = __shfl_up(incl, o, kWave);
        if (lane >= o) incl += v;
    }
    if (lane == kWave - 1) s_wave[wave] = incl;
    if (T == 0 && active) {
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) nz += __shfl_xor(nz, o, kWave);
        if (lane == 0 && nz) atomicAdd(&s_nz, nz);
    }
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < wave; ++w) run += s_wave[w];
    for (int k = k0; k < k1; ++k) {
        const int c = s_cnt[k];
        s_cnt[k] = run;  // first position of key k, relative to the tile
        if (T > 0 && (fine ? (k & 255) == 0 : true)) group_start[tile * T + (fine ? k >> 8 : k)] = b + run;
        run += c;
    }
    if (t == 0) {
        if (T == 0) group_start[tile] = b;
        if (tile == ntiles - 1) group_start[T > 0 ? ntiles * T : ntiles] = e;
    }
    __syncthreads();
    // one plain store per tile: thousands of atomics on one counter serialise (12 ns each: 170 us for 3600 tiles)
    if (T == 0 && active && t == 0) active[tile] = s_nz;
    // two events per thread and round: the loads of a round are independent of its LDS atomics
    for (int i0 = b + 2 * t; i0 < e; i0 += 2 * kTileSortThreads) {
        const bool two = i0 + 1 < e;
        const uint2 ea = in.evp[i0], eb = two ? in.evp[i0 + 1] : make_uint2(0u, 0u);
        float rxa = 0.f, rya = 0.f, rxb = 0.f, ryb = 0.f;
        float2 rla = make_float2(0.f, 0.f), rlb = make_float2(0.f, 0.f);
        if (frac) {
            rxa = in.rx[i0];
            rya = in.ry[i0];
            rla = in.rl[i0];
            if (two) {
                rxb = in.rx[i0 + 1];
                ryb = in.ry[i0 + 1];
                rlb = in.rl[i0 + 1];
            }
        }
        const double ta = in.tau64[i0], tb = two ? in.tau64[i0 + 1] : 0.0;
        const int pa = b + atomicAdd(&s_cnt[sub_key(ea.x)], 1);
        out.evp[pa] = ea;
        if (frac) {
            out.rx[pa] = rxa;
            out.ry[pa] = rya;
            out.rl[pa] = rla;
        }
        out.tau64[pa] = ta;
        if (two) {
            const int pb = b + atomicAdd(&s_cnt[sub_key(eb.x)], 1);
            out.evp[pb] = eb;
            if (frac) {
                out.rx[pb] = rxb;
                out.ry[pb] = ryb;
                out.rl[pb] = rlb;
            }
            out.tau64[pb] = tb;
        }
    }
}

// S4b (time SLABS, round 4: cmax_set_time_slabs).  k_tile_sort leaves the (tile, bin) groups tile-major -- a tile's bins one after the
// other: what the voxel gradient's accumulators want.  For LARGE MOTIONS of a 2-DoF or dense objective the groups are re-ordered
// slab-major inside every tile row: (tile row, time slab, tile column).  Consecutive groups are then neighbouring tiles of ONE slab, a
// segment of three of them spans 1/S of the batch's duration, and its LDS window is the tiles' extent plus 1/S of the displacement
// range instead of all of it (150 px over the batch: the window no longer overflows; 80 px: a quarter of the cells to clear and flush).
//   k_slab_offsets  one workgroup: new start of every group in slab-major order (exclusive scan of the group sizes in that order)
//   k_slab_regroup  one workgroup per group: copies the group to its new place
__device__ __forceinline__ int slab_old_group(int gnew, int ntc, int T) {  // (row, slab, col) -> (row, col, slab)
    const int col = gnew % ntc, rs = gnew / ntc, slab = rs % T, row = rs / T;
    return (row * ntc + col) * T + slab;
}
__global__ void __launch_bounds__(1024) k_slab_offsets(const int *__restrict__ old_start, int ngroups, int ntc, int T, int *__restrict__ new_start) {
    __shared__ int part[1024];
    const int t = threadIdx.x, per = (ngroups + 1023) / 1024, b = t * per, e = min(b + per, ngroups);
    int sum = 0;
    for (int g = b; g < e; ++g) {
        const int go = slab_old_group(g, ntc, T);
        sum += old_start[go + 1] - old_start[go];
    }
    part[t] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan of the per-thread totals
        const int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = old_start[0] + part[t] - sum;
    for (int g = b; g < e; ++g) {
        const int go = slab_old_group(g, ntc, T);
        new_start[g] = run;
        run += old_start[go + 1] - old_start[go];
    }
    if (t == 1023) new_start[ngroups] = old_start[0] + part[1023];
}
__global__ void __launch_bounds__(256) k_slab_regroup(const int *__restrict__ old_start, const int *__restrict__ new_start, int ntc, int T, SortOut in, SortOut out,
                                                      const int *__restrict__ flags) {
    const bool frac = flags[0] != 0;
    const int gnew = blockIdx.x, go = slab_old_group(gnew, ntc, T);
    const int b = old_start[go], n = old_start[go + 1] - b, d = new_start[gnew];
    for (int i = threadIdx.x; i < n; i += 256) {
        out.evp[d + i] = in.evp[b + i];
        out.tau64[d + i] = in.tau64[b + i];
        if (frac) {
            out.rx[d + i] = in.rx[b + i];
            out.ry[d + i] = in.ry[b + i];
            out.rl[d + i] = in.rl[b + i];
        }
    }
}

// S5 (un-binned handles).  Inside a run of equal source pixel the tile sort leaves the events in the order its LDS atomics
// happened to produce.  Ordering every run BY TIME puts events that warp to neighbouring places next to each other: K1 sums the
// votes of consecutive events of a thread that fall into the same cell in registers before its LDS atomics, and with
// time-ordered runs half of the neighbouring pairs share their cell on a dense batch (cfg3: 52 % instead of 13 %; cfg2 28 %
// instead of 6 %) -- a third fewer LDS atomics, the resource K1 is short of.  One thread per event: rank inside its run by
// (time, index), runs longer than kRunSortMax are copied as they are (the rank costs O(run) loads per event).
constexpr int kRunSortMax = 192;
constexpr int kRunSortChunk = 1024;            // events per workgroup (256 threads x 4)
constexpr int kRunSortHalo = kRunSortMax + 8;  // neighbours staged on either side: a run is followed at most kRunSortMax + 1 events far
constexpr int kRunSortSpan = kRunSortChunk + 2 * kRunSortHalo;  // staged positions
constexpr int kRunSortPer = (kRunSortSpan + 255) / 256;          // ... per thread (consecutive) in the two boundary scans
// The (pixel key, time bits) of a chunk and its halos are staged in LDS once (round 3: with global loads the kernel cost O(run
// length) memory round trips per event).  Round 5: the RUN of an event -- first and one-past-last staged position of its pixel --
// comes from two scans over the staged keys (the last run head at or before a position, the first one behind it) instead of each
// thread walking outwards from its event with dependent LDS reads (~140 serial reads per event at 69 events per pixel: 64M events
// 4.45 ms, a quarter of a TB/s), and the ranking loop reads four neighbours per step, all independent: the kernel is then bound by
// LDS throughput and the 32 B per event it moves (profiles/r05_set_events.txt).
__global__ void __launch_bounds__(256) k_run_time_sort(SortOut in, SortOut out, const int *__restrict__ total, const int *__restrict__ flags) {
    __shared__ uint32_t s_key[kRunSortSpan + 1];   // [q + 1]: s_key[0] is the position in front of the staged range
    __shared__ uint32_t s_tau[kRunSortSpan];
    __shared__ short s_rs[kRunSortSpan], s_re[kRunSortSpan];  // run start / one past the run's end (staged positions); -1 / span + 1: out of sight
    __shared__ int s_wave[2][256 / kWave];
    const int64_t n = *total;  // events that survived the packing (the grid covers the batch as it came in)
    const int64_t base = (int64_t)blockIdx.x * kRunSortChunk;
    if (base >= n) return;
    const bool frac = flags[0] != 0;
    const int64_t lo = base - kRunSortHalo;  // global index of staged position 0
    const int t = threadIdx.x, lane = t & (kWave - 1), wave = t / kWave;
    // keys outside the batch: two different values no event has (the top byte is masked off real keys), so that the batch's first and
    // last run end where the batch ends; the position in front of the staged range: unknown -> a third value, handled as "out of sight"
    for (int q = t; q < kRunSortSpan + 1; q += 256) {
        const int64_t g = lo + q - 1;
        uint32_t key = g < 0 ? 0xFFFFFFFEu : 0xFFFFFFFFu;
        if (g >= 0 && g < n) {
            const uint2 e = in.evp[g];
            key = e.x & 0x00FFFFFFu;
            if (q >= 1) s_tau[q - 1] = e.y;
        } else if (q >= 1) {
            s_tau[q - 1] = 0u;
        }
        s_key[q] = key;
    }
    __syncthreads();
    // head[q]: position q starts a run (its key differs from the one in front of it).  Position 0's predecessor is staged too
    // (s_key[0]) unless it lies in front of the batch.
    auto head = [&](int q) { return s_key[q + 1] != s_key[q]; };
    const int q0 = t * kRunSortPer, q1 = min(q0 + kRunSortPer, kRunSortSpan);
    // forward: last head at or before q (-1: none staged)
    int last = -1;
    for (int q = q0; q < q1; ++q) last = head(q) ? q : last;
    int incl = last;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const int v = __shfl_up(incl, o, kWave);
        if (lane >= o) incl = max(incl, v);
    }
    if (lane == kWave - 1) s_wave[0][wave] = incl;
    // backward: first head behind q (span + 1: none staged)
    int first = kRunSortSpan + 1;
    for (int q = q1 - 1; q >= q0; --q) first = head(q) ? q : first;
    int incb = first;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const int v = __shfl_down(incb, o, kWave);
        if (lane + o < kWave) incb = min(incb, v);
    }
    if (lane == 0) s_wave[1][wave] = incb;
    __syncthreads();
    {
        int carry = __shfl_up(incl, 1, kWave);  // exclusive over the lanes in front of this one
        if (lane == 0) carry = -1;
        for (int w = 0; w < wave; ++w) carry = max(carry, s_wave[0][w]);
        int run = carry;
        for (int q = q0; q < q1; ++q) {
            run = head(q) ? q : run;
            s_rs[q] = (short)run;
        }
        int carryb = __shfl_down(incb, 1, kWave);  // exclusive over the lanes behind this one
        if (lane == kWave - 1) carryb = kRunSortSpan + 1;
        for (int w = wave + 1; w < 256 / kWave; ++w) carryb = min(carryb, s_wave[1][w]);
        int nxt = carryb;
        for (int q = q1 - 1; q >= q0; --q) {
            s_re[q] = (short)nxt;  // the first head BEHIND q
            nxt = head(q) ? q : nxt;
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kRunSortChunk / 256; ++u) {
        const int li = u * 256 + t;  // index inside the chunk
        const int64_t i = base + li;
        if (i >= n) continue;
        const int qi = li + kRunSortHalo;
        const int b = s_rs[qi], e = s_re[qi];
        int64_t pos = i;
        // the whole run in sight (its head is staged -- a head at staged position 0 counts only where the position in front of it is
        // known: s_key[0] is a real key or the front of the batch -- and so is the head of the next run) and short enough: rank by (time, index)
        if (b >= 0 && e <= kRunSortSpan && e - b <= kRunSortMax && e - b > 1) {
            const uint32_t ti = s_tau[qi];  // tau >= 0: the fp32 bit patterns order like the values
            int rank = 0;
            int j = b;
            for (; j + 4 <= e; j += 4) {
                const uint32_t t0 = s_tau[j], t1 = s_tau[j + 1], t2 = s_tau[j + 2], t3 = s_tau[j + 3];
                rank += (t0 < ti || (t0 == ti && j < qi)) ? 1 : 0;
                rank += (t1 < ti || (t1 == ti && j + 1 < qi)) ? 1 : 0;
                rank += (t2 < ti || (t2 == ti && j + 2 < qi)) ? 1 : 0;
                rank += (t3 < ti || (t3 == ti && j + 3 < qi)) ? 1 : 0;
            }
            for (; j < e; ++j) {
                const uint32_t tj = s_tau[j];
                rank += (tj < ti || (tj == ti && j < qi)) ? 1 : 0;
            }
            pos = lo + b + rank;
        }
        const uint2 ev = in.evp[i];
        out.evp[pos] = ev;
        if (frac) {
            out.rx[pos] = in.rx[i];
            out.ry[pos] = in.ry[i];
            out.rl[pos] = in.rl[i];
        }
        out.tau64[pos] = in.tau64[i];
    }
}

}  // namespace cmax
