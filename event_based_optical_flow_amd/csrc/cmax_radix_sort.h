// Per-batch preparation for LARGE batches: a STABLE least-significant-digit radix sort of the packed events (round 5).
//
// Same result as the two-level counting sort of cmax_sort_kernels.h -- source tile (16 x 16) major, inside a tile by pixel (un-binned
// handles) or by (time bin, pixel) / time bin, group starts per tile or (tile, bin) -- with one difference that makes its last stage
// unnecessary: every pass keeps the input order of equal keys.  Sensor batches arrive time-ordered (src/utils/event_utils.py:18-47,
// src/data_loader/mvsec.py:178-207), so the events of a pixel come out BY TIME without k_run_time_sort (and without its 192-event
// limit), and the packed order no longer depends on the arrival order of LDS / global atomics: it is a function of the batch alone.
//
//   key = group << 8 | pixel-in-tile   (group = tile, or tile * T + time bin)      [binned handles with T > 32: key = group]
//   digits of <= 6 bits, lowest first; pass 0 reads the RAW events (and packs while scattering): its digit -- pixel bits, and on un-binned
//   handles the lowest tile bits -- follows from the source pixel alone, the time bin (which needs the batch's extremes) sits above it
//   per pass:  R1 k_rs_hist*    <= 512 workgroups, each owning a contiguous range of the input: digit histogram in LDS -> hist[digit][wg]
//              R2 exclusive scan of hist in (digit, wg) order (the caller's scan kernels)
//              R3 k_rs_scatter* the same workgroups walk their range IN ORDER, 512 events at a time: the rank of an event among the
//                               events of its digit = running offset of (wg, digit) + events of that digit in the waves in front of its
//                               own (one byte per (digit, wave) in LDS) + lanes of its own wave in front of it (a match-any built from
//                               one ballot per digit bit)
//   then       R4 k_rs_meta     one pass over the sorted keys: group starts (empty groups included), source pixels that hold events,
//                               the batch's time extremes as doubles
// What a pass costs depends on HOW MANY OUTPUT STREAMS a workgroup writes (measured, 64M events at 720p, 2 GB moved per pass:
// 16 streams 0.47 ms, 32: 0.53, 64: 0.66, 128: 0.82, 1024: 1.41 -- scattered 8-byte stores over hundreds of pages), so the key (20 bits
// at 720p) is sorted in four 5-bit passes rather than two 10-bit ones: 3.5 ms against 3.9, and against 4.75 for the counting sort, whose
// scatter writes 3600 streams (profiles/r05_set_events.txt).  Below ~8M events the two pipelines tie and the counting sort (fewer
// launches) is kept.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cmax_common.h"
#include "cmax_sort_kernels.h"

namespace cmax {

constexpr int kRsThreads = 512;          // R1 / R3
constexpr int kRsWaves = kRsThreads / kWave;
constexpr int kRsSlots = 4;              // events per thread and iteration of R3 (their loads are in flight together)
constexpr int kRsChunk = kRsThreads * kRsSlots;
constexpr int kRsMaxGroups = 1024;       // workgroups per pass (four per CU: all resident)
constexpr int kRsMaxDigitBits = 6;   // (see sort_events: a pass is as fast as its output streams are few)
static_assert(kRsWaves == 8, "one byte per wave in a 64-bit LDS word");

struct RsKey {
    int ntc, T, fine;  // tile columns; time bins (0: un-binned); fine: the key carries the pixel-in-tile below the group
};
__device__ __forceinline__ uint32_t rs_key(uint32_t x, const RsKey &k) {
    const uint32_t ix = x & 0xFFFu, iy = (x >> 12) & 0xFFFu, bin = x >> 24;
    const uint32_t tile = (ix >> 4) * (uint32_t)k.ntc + (iy >> 4);
    const uint32_t group = k.T > 0 ? tile * (uint32_t)k.T + bin : tile;
    return k.fine ? (group << 8) | ((ix & 15u) << 4) | (iy & 15u) : group;
}
__device__ __forceinline__ uint32_t rs_group(uint32_t x, const RsKey &k) {
    const uint32_t ix = x & 0xFFFu, iy = (x >> 12) & 0xFFFu, bin = x >> 24;
    const uint32_t tile = (ix >> 4) * (uint32_t)k.ntc + (iy >> 4);
    return k.T > 0 ? tile * (uint32_t)k.T + bin : tile;
}

// one packed event with everything the sort carries along
struct RsItem {
    uint2 e;
    double tau;
    float rx, ry;
    float2 rl;
};
__device__ __forceinline__ RsItem rs_pack(const SortItem &it, int T) {
    RsItem r;
    // top byte: the voxel time bin (binned handle) or the time residual beyond fp32 (un-binned handle, tau_residual8)
    const uint32_t bin = T > 0 ? (uint32_t)sort_voxel_bin(it.tn, T) : tau_residual8(it.tn);
    r.e = make_uint2((uint32_t)it.ix | ((uint32_t)it.iy << 12) | (bin << 24), __float_as_uint((float)it.tn));
    r.tau = it.tn;
    r.rx = it.rx;
    r.ry = it.ry;
    r.rl = make_float2(it.rxl, it.ryl);
    return r;
}

// R0: everything the sort accumulates into
__global__ void __launch_bounds__(256) k_rs_clear(int *__restrict__ active, int nactive, int *__restrict__ flags, int first_flag,
                                                  unsigned long long *__restrict__ tmm_keys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nactive) active[i] = 0;
    if (i >= first_flag && i < 4) flags[i] = 0;
    if (tmm_keys && i < 2) tmm_keys[i] = 0ull;
}

// the batch's time extremes alone (binned handles whose FIRST digit already needs the time bin: T > 32)
template <typename SRC>
__global__ void __launch_bounds__(kRsThreads) k_rs_time_extremes(SRC src, int64_t n, unsigned long long *__restrict__ tmm_keys) {
    __shared__ double s_lo[kRsWaves], s_hi[kRsWaves];
    double lo = INFINITY, hi = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * kRsThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kRsThreads) {
        const double t = src.time(i);
        lo = fmin(lo, t);
        hi = fmax(hi, t);
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        lo = fmin(lo, __shfl_xor(lo, o, kWave));
        hi = fmax(hi, __shfl_xor(hi, o, kWave));
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        s_lo[threadIdx.x / kWave] = lo;
        s_hi[threadIdx.x / kWave] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kRsWaves; ++w) {
            lo = fmin(lo, s_lo[w]);
            hi = fmax(hi, s_hi[w]);
        }
        if (lo <= hi) {
            atomicMax(&tmm_keys[0], ~sort_f64_key(lo));
            atomicMax(&tmm_keys[1], sort_f64_key(hi));
        }
    }
}

// R1, pass 0 (from the source).  hist [D][nwg]; flags[0] = any fractional source coordinate, flags[1] += dropped events, flags[2] +=
// events kept from off the sensor; tmm_keys (keyed extremes): batch time extremes, two atomics per workgroup.  Dynamic LDS: D ints.
template <typename SRC>
__global__ void __launch_bounds__(kRsThreads)
k_rs_hist_src(SRC src, int64_t n, int64_t range, RsKey key, int bits, int *__restrict__ hist, int *__restrict__ flags,
              unsigned long long *__restrict__ tmm_keys) {
    extern __shared__ int s_rshist[];
    __shared__ double s_lo[kRsWaves], s_hi[kRsWaves];
    const int D = 1 << bits, nwg = gridDim.x;
    for (int d = threadIdx.x; d < D; d += kRsThreads) s_rshist[d] = 0;
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * range, r1 = min(n, r0 + range);
    int dropped = 0, outside = 0;
    bool frac = false;
    double lo = INFINITY, hi = -INFINITY;
    for (int64_t i = r0 + threadIdx.x; i < r1; i += kRsThreads) {
        // (the pixel alone while the first digit is the pixel-in-tile: the time bin -- which needs the extremes reduced HERE -- sits above it)
        const SortItem it = key.fine ? src.pixel(i) : src.full(i);
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
        uint32_t x = (uint32_t)it.ix | ((uint32_t)it.iy << 12);
        if (!key.fine && key.T > 0) x |= (uint32_t)sort_voxel_bin(it.tn, key.T) << 24;
        atomicAdd(&s_rshist[rs_key(x, key) & (uint32_t)(D - 1)], 1);
    }
    if (frac) flags[0] = 1;
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) dropped += __shfl_xor(dropped, o, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0 && dropped) atomicAdd(&flags[1], dropped);
    if (__any(outside != 0)) {
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
        for (int w = 1; w < kRsWaves; ++w) {
            lo = fmin(lo, s_lo[w]);
            hi = fmax(hi, s_hi[w]);
        }
        if (lo <= hi) {
            atomicMax(&tmm_keys[0], ~sort_f64_key(lo));
            atomicMax(&tmm_keys[1], sort_f64_key(hi));
        }
    }
    for (int d = threadIdx.x; d < D; d += kRsThreads) hist[(int64_t)d * nwg + blockIdx.x] = s_rshist[d];
}

// R1, later passes (from the packed words)
__global__ void __launch_bounds__(kRsThreads)
k_rs_hist(const uint2 *__restrict__ evp, const int *__restrict__ total, int64_t range, RsKey key, int shift, int bits, int *__restrict__ hist) {
    extern __shared__ int s_rshist[];
    const int64_t n = *total;  // events that survived the packing (the previous pass's scan left their number behind its offsets)
    const int D = 1 << bits, nwg = gridDim.x;
    for (int d = threadIdx.x; d < D; d += kRsThreads) s_rshist[d] = 0;
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * range, r1 = min(n, r0 + range);
    for (int64_t i = r0 + threadIdx.x; i < r1; i += kRsThreads) atomicAdd(&s_rshist[(rs_key(evp[i].x, key) >> shift) & (uint32_t)(D - 1)], 1);
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += kRsThreads) hist[(int64_t)d * nwg + blockIdx.x] = s_rshist[d];
}

// lanes of this wave whose digit equals mine (valid lanes only): one ballot per digit bit
__device__ __forceinline__ unsigned long long rs_match(uint32_t digit, bool valid, int bits) {
    unsigned long long m = __ballot(valid);
    for (int b = 0; b < bits; ++b) {
        const bool bit = ((digit >> b) & 1u) != 0u;
        const unsigned long long bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}
__device__ __forceinline__ int rs_byte_sum(unsigned long long v) {
    return (int)(__builtin_amdgcn_sad_u8((unsigned)v, 0u, 0u) + __builtin_amdgcn_sad_u8((unsigned)(v >> 32), 0u, 0u));
}
// Position of one event per thread among the events of its digit (all threads call; two barriers).  s_base[D]: running offset of
// (this workgroup, digit); s_cnt[D]: one byte per wave = events of the digit that the wave holds in THIS step (zero between steps).
__device__ __forceinline__ int64_t rs_position(uint32_t digit, bool valid, int bits, int *s_base, unsigned long long *s_cnt) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const unsigned long long m = rs_match(digit, valid, bits);
    const unsigned long long below = m & ((1ull << lane) - 1ull);
    const int r = __popcll(below);
    const bool leader = valid && below == 0ull;
    unsigned char *bytes = reinterpret_cast<unsigned char *>(s_cnt);
    if (leader) bytes[digit * 8u + (unsigned)wave] = (unsigned char)__popcll(m);
    __syncthreads();
    int pre = 0, tot = 0;
    int64_t pos = -1;
    if (valid) {
        const unsigned long long c = s_cnt[digit];
        pre = rs_byte_sum(c & ((1ull << (8 * wave)) - 1ull));  // waves in front of mine
        tot = rs_byte_sum(c);
        pos = (int64_t)s_base[digit] + pre + r;
    }
    __syncthreads();
    if (leader) {
        bytes[digit * 8u + (unsigned)wave] = 0;
        if (pre == 0) s_base[digit] += tot;  // the first wave that holds the digit (no events of it in front: pre == 0) moves the offset on
    }
    return pos;
}

__device__ __forceinline__ void rs_store(const SortOut &out, int64_t pos, const RsItem &it, bool frac) {
    out.evp[pos] = it.e;
    if (frac) {
        out.rx[pos] = it.rx;
        out.ry[pos] = it.ry;
        out.rl[pos] = it.rl;
    }
    out.tau64[pos] = it.tau;
}

// R3, pass 0: pack + scatter by the first digit.  base [D][nwg] = exclusive scan of R1's histogram.  Dynamic LDS: D ints + D 64-bit words.
template <typename SRC>
__global__ void __launch_bounds__(kRsThreads)
k_rs_scatter_src(SRC src, int64_t n, int64_t range, RsKey key, int bits, const int *__restrict__ base, const int *__restrict__ flags, SortOut out) {
    extern __shared__ int s_rsdyn[];
    const int D = 1 << bits, nwg = gridDim.x;
    int *s_base = s_rsdyn;
    unsigned long long *s_cnt = reinterpret_cast<unsigned long long *>(s_rsdyn + D);
    for (int d = threadIdx.x; d < D; d += kRsThreads) {
        s_base[d] = base[(int64_t)d * nwg + blockIdx.x];
        s_cnt[d] = 0ull;
    }
    const bool frac = flags[0] != 0;  // (complete: R1 wrote it)
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * range, r1 = min(n, r0 + range);
    // The loads of iteration k + 1 are issued BEFORE iteration k is ranked and stored: memory returns in order, so a wait for loads that
    // were issued ahead of a batch of stores does not wait for those stores (with load - rank - store per iteration every iteration paid
    // a load latency plus a store drain: 64M events 1.5 ms for this pass instead of see profiles/r05_set_events.txt).
    typename SRC::Fetched nxt[kRsSlots];
    auto fetch = [&](int64_t b0) {
#pragma unroll
        for (int u = 0; u < kRsSlots; ++u) {  // (step u holds events b0 + 512 u .. + 511: the steps are walked in order)
            const int64_t i = b0 + (int64_t)u * kRsThreads + threadIdx.x;
            nxt[u] = src.fetch(i < r1 ? i : r1 - 1);  // (unconditional at a clamped index: all in flight at once)
        }
    };
    if (r0 < r1) fetch(r0);
    for (int64_t b0 = r0; b0 < r1; b0 += kRsChunk) {
        RsItem item[kRsSlots];
        bool valid[kRsSlots];
#pragma unroll
        for (int u = 0; u < kRsSlots; ++u) {
            const int64_t i = b0 + (int64_t)u * kRsThreads + threadIdx.x;
            const SortItem it = src.resolve(nxt[u], i);
            valid[u] = i < r1 && it.ix >= 0;
            item[u] = rs_pack(it, key.T);
        }
        if (b0 + kRsChunk < r1) fetch(b0 + kRsChunk);  // workgroup-uniform
#pragma unroll
        for (int u = 0; u < kRsSlots; ++u) {
            if (b0 + (int64_t)u * kRsThreads >= r1) break;  // workgroup-uniform
            const uint32_t digit = valid[u] ? rs_key(item[u].e.x, key) & (uint32_t)(D - 1) : 0u;
            const int64_t pos = rs_position(digit, valid[u], bits, s_base, s_cnt);
            if (valid[u]) rs_store(out, pos, item[u], frac);
        }
    }
}

// R3, later passes
__global__ void __launch_bounds__(kRsThreads)
k_rs_scatter(SortOut in, const int *__restrict__ total, int64_t range, RsKey key, int shift, int bits, const int *__restrict__ base,
             const int *__restrict__ flags, SortOut out) {
    extern __shared__ int s_rsdyn[];
    const int64_t n = *total;
    const int D = 1 << bits, nwg = gridDim.x;
    int *s_base = s_rsdyn;
    unsigned long long *s_cnt = reinterpret_cast<unsigned long long *>(s_rsdyn + D);
    for (int d = threadIdx.x; d < D; d += kRsThreads) {
        s_base[d] = base[(int64_t)d * nwg + blockIdx.x];
        s_cnt[d] = 0ull;
    }
    const bool frac = flags[0] != 0;
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * range, r1 = min(n, r0 + range);
    RsItem nxt[kRsSlots];  // (the next iteration's loads ahead of this iteration's stores: see k_rs_scatter_src)
    auto fetch = [&](int64_t b0) {
#pragma unroll
        for (int u = 0; u < kRsSlots; ++u) {
            const int64_t i = b0 + (int64_t)u * kRsThreads + threadIdx.x;
            const int64_t j = i < r1 ? i : r1 - 1;  // (unconditional loads at a clamped index: all in flight at once)
            nxt[u].e = in.evp[j];
            nxt[u].tau = in.tau64[j];
            if (frac) {
                nxt[u].rx = in.rx[j];
                nxt[u].ry = in.ry[j];
                nxt[u].rl = in.rl[j];
            }
        }
    };
    if (r0 < r1) fetch(r0);
    for (int64_t b0 = r0; b0 < r1; b0 += kRsChunk) {
        RsItem item[kRsSlots];
        bool valid[kRsSlots];
#pragma unroll
        for (int u = 0; u < kRsSlots; ++u) {
            item[u] = nxt[u];
            valid[u] = b0 + (int64_t)u * kRsThreads + threadIdx.x < r1;
        }
        if (b0 + kRsChunk < r1) fetch(b0 + kRsChunk);  // workgroup-uniform
#pragma unroll
        for (int u = 0; u < kRsSlots; ++u) {
            if (b0 + (int64_t)u * kRsThreads >= r1) break;  // workgroup-uniform
            const uint32_t digit = (rs_key(item[u].e.x, key) >> shift) & (uint32_t)(D - 1);
            const int64_t pos = rs_position(digit, valid[u], bits, s_base, s_cnt);
            if (valid[u]) rs_store(out, pos, item[u], frac);
        }
    }
}

// R4: group starts from the sorted keys (group_start[g] = first event of group g; empty groups take the next group's start;
// group_start[ngroups] = n), source pixels that hold events (un-binned handles: += into active[workgroup % nactive], zero on entry),
// and the keyed extremes as the two doubles every other kernel reads.
__global__ void __launch_bounds__(256)
k_rs_meta(const uint2 *__restrict__ evp, const int *__restrict__ total, RsKey key, int ngroups, int *__restrict__ group_start,
          int *__restrict__ active, int nactive, unsigned long long *__restrict__ tmm_keys) {
    __shared__ int s_n[256 / kWave];
    const int64_t n = *total;
    if (tmm_keys && blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long k0 = tmm_keys[0], k1 = tmm_keys[1];
        double *d = reinterpret_cast<double *>(tmm_keys);
        d[0] = (k0 | k1) ? sort_f64_unkey(~k0) : (double)INFINITY;
        d[1] = (k0 | k1) ? sort_f64_unkey(k1) : -(double)INFINITY;
    }
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int heads = 0;
    if (n == 0 && i == 0)  // every event was dropped: all groups empty
        for (int q = 0; q <= ngroups; ++q) group_start[q] = 0;
    if (i < n) {
        const uint32_t x = evp[i].x, xp = i > 0 ? evp[i - 1].x : 0u;
        const int g = (int)rs_group(x, key), gp = i > 0 ? (int)rs_group(xp, key) : -1;
        for (int q = gp + 1; q <= g; ++q) group_start[q] = (int)i;
        if (i == n - 1)
            for (int q = g + 1; q <= ngroups; ++q) group_start[q] = (int)n;
        heads = (i == 0 || ((x ^ xp) & 0x00FFFFFFu) != 0u) ? 1 : 0;  // a new source pixel
    }
    if (active) {
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) heads += __shfl_xor(heads, o, kWave);
        if ((threadIdx.x & (kWave - 1)) == 0) s_n[threadIdx.x / kWave] = heads;
        __syncthreads();
        if (threadIdx.x == 0) {
            int a = 0;
            for (int w = 0; w < 256 / kWave; ++w) a += s_n[w];
            if (a) atomicAdd(&active[blockIdx.x % nactive], a);
        }
    }
}

}  // namespace cmax
