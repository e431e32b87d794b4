// One launch instead of three: the workgroups of cfg2's K1, K3 and finish as ROLES of one grid (2 N + 1 workgroups of 512
// threads), the dependencies carried by device-scope counters instead of kernel boundaries.  The dispatcher hands out
// workgroups in the order of their linear id, so every vote workgroup holds its CU slot before the first gradient
// workgroup gets one: a gradient workgroup that waits can only wait for workgroups that are already running.
//   role A (b < N)        two dependent loads, then one device-scope atomic per thread into the image; arrive(A)
//   role B (N <= b < 2N)  two dependent loads (the events), WAIT(A), read the image, write a partial sum; arrive(B)
//   role C (b == 2N)      WAIT(B), sum the partials, write the result, reset the counters
// Timed against the same bodies as three dependent launches.  Checks the result (a stale read of the image shows).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_chain.hip -o tools/microbench_chain
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

constexpr int kLine = 32;  // ints per 128-byte line
constexpr int kMaxSub = 32;
// sync block layout (lines): [0..32) sub-counters A, [32..64) sub-counters B, 64 master A, 65 master B, 66 flag A, 67 flag B
constexpr int kSyncLines = 2 * kMaxSub + 4;
struct Sync {
    int *base;
    __device__ int *subA(int i) const { return base + i * kLine; }
    __device__ int *subB(int i) const { return base + (kMaxSub + i) * kLine; }
    __device__ int *masterA() const { return base + (2 * kMaxSub) * kLine; }
    __device__ int *masterB() const { return base + (2 * kMaxSub + 1) * kLine; }
    __device__ int *flagA() const { return base + (2 * kMaxSub + 2) * kLine; }
    __device__ int *flagB() const { return base + (2 * kMaxSub + 3) * kLine; }
};

__device__ __forceinline__ int ld_agent(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// all memory operations of this workgroup have been performed at device scope, then one arrival
__device__ __forceinline__ void arrive(int *sub, int per_sub, int *master, int nsub, int *flag, int epoch) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int a = __hip_atomic_fetch_add(sub, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a == per_sub - 1) {
            const int m = __hip_atomic_fetch_add(master, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (m == nsub - 1) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// single level: one no-return atomic per workgroup on one of NSUB lines; the waiting side sums the lines (epoch-free:
// the counters are reset by role C, the target is the number of workgroups)
__device__ __forceinline__ void arrive1(int *sub) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) (void)__hip_atomic_fetch_add(sub, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int NSUB, int SLEEP>
__device__ __forceinline__ bool wait_count(const int *sub0, int target) {
    __shared__ int s_ok;
    if (threadIdx.x < 64) {
        int ok = 0;
        for (int spin = 0; spin < (1 << 22); ++spin) {
            int v = (int)threadIdx.x < NSUB ? ld_agent(sub0 + threadIdx.x * kLine) : 0;
#pragma unroll
            for (int o = 1; o < NSUB; o <<= 1) v += __shfl_xor(v, o);
            if (__builtin_amdgcn_readfirstlane(v) >= target) {
                ok = 1;
                break;
            }
            if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
        }
        if (threadIdx.x == 0) s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0;
}

template <bool INV>
__device__ __forceinline__ bool wait_flag(const int *flag, int epoch) {
    __shared__ int s_ok;
    if (threadIdx.x == 0) {
        int ok = 0;
        for (int spin = 0; spin < (1 << 22); ++spin) {
            if (ld_agent(flag) == epoch) {
                ok = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        s_ok = ok;
    }
    __syncthreads();
    if (INV) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return s_ok != 0;
}

__device__ __forceinline__ void body_A(int b, const int *idx, const float *src, float *img, int npix, int n) {
    const int i = (b * blockDim.x + threadIdx.x) % n;
    const float v = src[idx[i]] + 1.f;
    atomicAdd(&img[(b * 512 + threadIdx.x) % npix], v);  // device scope
}

template <bool INV, int NSUB = 0, int SLEEP = 2, bool NOWAIT = false>
__device__ __forceinline__ void body_B(int b, const int *idx, const float *src, const float *img, int npix, int n, double *part,
                                       const int *flagA, int epoch, bool chained) {
    const int i = (b * blockDim.x + threadIdx.x) % n;
    const float e = src[idx[i]];  // the events: independent of the image
    if (chained && !NOWAIT) {
        if (NSUB == 0) {
            if (!wait_flag<INV>(flagA, epoch)) return;
        } else {
            if (!wait_count<NSUB, SLEEP>(flagA, epoch)) return;  // flagA = first sub-counter, epoch = target
        }
    }
    float acc = e;
    // this workgroup's slice of the image
    const int chunk = (npix + gridDim.x / 2 - 1) / (gridDim.x / 2);
    for (int p = b * chunk + threadIdx.x; p < min(npix, (b + 1) * chunk); p += blockDim.x) acc += (chained && !INV) ? ld_agent(img + p) : img[p];
    __shared__ float s[512];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int k = 256; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) s[threadIdx.x] += s[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (chained) __hip_atomic_store(&part[b], (double)s[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else part[b] = (double)s[0];
    }
}

__device__ __forceinline__ void body_C(int N, const double *part, double *result, bool chained) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < N; i += blockDim.x) acc += chained ? ld_agent(part + i) : part[i];
    __shared__ double s[512];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int k = blockDim.x / 2; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) s[threadIdx.x] += s[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) result[0] = s[0];
}

__global__ void k_A(const int *idx, const float *src, float *img, int npix, int n) { body_A(blockIdx.x, idx, src, img, npix, n); }
__global__ void k_B(const int *idx, const float *src, const float *img, int npix, int n, double *part) {
    body_B<false>(blockIdx.x, idx, src, img, npix, n, part, nullptr, 0, false);
}
__global__ void k_C(int N, const double *part, double *result) { body_C(N, part, result, false); }

template <bool INV>
__global__ void __launch_bounds__(512) k_chain(int N, const int *idx, const float *src, float *img, int npix, int n, double *part, double *result,
                                               Sync sy, int epoch) {
    const int b = blockIdx.x;
    if (b < N) {
        body_A(b, idx, src, img, npix, n);
        arrive(sy.subA(b & 7), (N - (b & 7) + 7) / 8, sy.masterA(), 8, sy.flagA(), epoch);
    } else if (b < 2 * N) {
        const int bb = b - N;
        body_B<INV>(bb, idx, src, img, npix, n, part, sy.flagA(), epoch, true);
        arrive(sy.subB(bb & 7), (N - (bb & 7) + 7) / 8, sy.masterB(), 8, sy.flagB(), epoch);
    } else {
        if (!wait_flag<false>(sy.flagB(), epoch)) {
            if (threadIdx.x == 0) result[1] = -1.0;  // watchdog
            return;
        }
        body_C(N, part, result, true);
        if (threadIdx.x < 2 * kMaxSub + 2) __hip_atomic_store(sy.base + threadIdx.x * kLine, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// single-level variant; NOWAIT: the waits removed (wrong results: the lower bound of the overlapped execution)
template <int NSUB, int SLEEP, bool NOWAIT>
__global__ void __launch_bounds__(512) k_chain1(int N, const int *idx, const float *src, float *img, int npix, int n, double *part, double *result,
                                                Sync sy) {
    const int b = blockIdx.x;
    if (b < N) {
        body_A(b, idx, src, img, npix, n);
        arrive1(sy.subA(b % NSUB));
    } else if (b < 2 * N) {
        const int bb = b - N;
        body_B<false, NSUB, SLEEP, NOWAIT>(bb, idx, src, img, npix, n, part, sy.subA(0), N, true);
        arrive1(sy.subB(bb % NSUB));
    } else {
        if (!NOWAIT && !wait_count<NSUB, SLEEP>(sy.subB(0), N)) {
            if (threadIdx.x == 0) result[1] = -1.0;  // watchdog
            return;
        }
        body_C(N, part, result, true);
        if (threadIdx.x < 2 * kMaxSub) __hip_atomic_store(sy.base + threadIdx.x * kLine, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <typename F>
static double time_us(F f, int iters) {
    for (int i = 0; i < 50; ++i) f();
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; ++i) f();
    hipDeviceSynchronize();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 704;
    const int n = 1 << 20, npix = 260 * 346;
    int *idx, *sync;
    float *src, *img;
    double *part, *result;
    hipMalloc(&idx, n * sizeof(int));
    hipMalloc(&src, n * sizeof(float));
    hipMalloc(&img, npix * sizeof(float));
    hipMalloc(&part, N * sizeof(double));
    hipMalloc(&result, 2 * sizeof(double));
    hipMalloc(&sync, kSyncLines * kLine * sizeof(int));
    hipMemset(idx, 0, n * sizeof(int));
    hipMemset(src, 0, n * sizeof(float));
    hipMemset(sync, 0, kSyncLines * kLine * sizeof(int));
    hipMemset(result, 0, 2 * sizeof(double));
    hipStream_t s;
    hipStreamCreate(&s);
    const Sync sy{sync};
    const int iters = 2000;
    int epoch = 0;
    const double expect_per_eval = (double)N * 512;  // every A thread adds 1
    auto check = [&](const char *what, int evals) {
        double r[2];
        hipMemcpy(r, result, sizeof(r), hipMemcpyDeviceToHost);
        printf("   %-34s result %.1f, expected %.1f%s%s\n", what, r[0], expect_per_eval * evals, r[0] == expect_per_eval * evals ? "  OK" : "  MISMATCH",
               r[1] < 0 ? "  WATCHDOG" : "");
    };
    int evals = 0;
    double t3 = time_us([&] {
        hipLaunchKernelGGL(k_A, dim3(N), dim3(512), 0, s, idx, src, img, npix, n);
        hipLaunchKernelGGL(k_B, dim3(N), dim3(512), 0, s, idx, src, img, npix, n, part);
        hipLaunchKernelGGL(k_C, dim3(1), dim3(512), 0, s, N, part, result);
        ++evals;
    }, iters);
    printf("three dependent launches (A, B, C), %d workgroups each           %6.2f us per evaluation\n", N, t3);
    check("three launches", evals);
    for (int reps : {1, 2, 8, 32}) {  // the three launches replayed from a hipGraph holding `reps` evaluations
        hipGraph_t g;
        hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int r = 0; r < reps; ++r) {
            hipLaunchKernelGGL(k_A, dim3(N), dim3(512), 0, s, idx, src, img, npix, n);
            hipLaunchKernelGGL(k_B, dim3(N), dim3(512), 0, s, idx, src, img, npix, n, part);
            hipLaunchKernelGGL(k_C, dim3(1), dim3(512), 0, s, N, part, result);
        }
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        const double t = time_us([&] {
            hipGraphLaunch(ge, s);
            evals += reps;
        }, iters / reps) / reps;
        printf("three launches replayed from a hipGraph, %2d evaluations per graph      %6.2f us per evaluation\n", reps, t);
        check("graph replay", evals);
        hipGraphExecDestroy(ge);
        hipGraphDestroy(g);
    }
    double tc = time_us([&] {
        ++epoch;
        hipLaunchKernelGGL(k_chain<false>, dim3(2 * N + 1), dim3(512), 0, s, N, idx, src, img, npix, n, part, result, sy, epoch);
        ++evals;
    }, iters);
    printf("ONE launch, roles chained by counters, image read with sc1 loads   %6.2f us per evaluation\n", tc);
    check("chained (sc1 loads)", evals);
#define RUN1(NSUB, SLEEP, NOWAIT, LABEL)                                                                                                  \
    {                                                                                                                                     \
        double t = time_us([&] {                                                                                                          \
            hipLaunchKernelGGL((k_chain1<NSUB, SLEEP, NOWAIT>), dim3(2 * N + 1), dim3(512), 0, s, N, idx, src, img, npix, n, part, result, sy); \
            ++evals;                                                                                                                      \
        }, iters);                                                                                                                        \
        printf("ONE launch, single-level counters, %-36s %6.2f us per evaluation\n", LABEL, t);                                         \
        if (!NOWAIT) check(LABEL, evals);                                                                                                 \
    }
    RUN1(8, 2, false, "8 lines, s_sleep 2")
    RUN1(8, 0, false, "8 lines, no sleep")
    RUN1(32, 2, false, "32 lines, s_sleep 2")
    RUN1(32, 0, false, "32 lines, no sleep")
    RUN1(32, 8, false, "32 lines, s_sleep 8")
    RUN1(8, 2, true, "NO WAITS (lower bound, wrong result)")
    hipMemset(img, 0, npix * sizeof(float));  // (the unsynchronised run leaves the running sum undefined)
    hipMemset(sync, 0, kSyncLines * kLine * sizeof(int));
    hipDeviceSynchronize();
    evals = 0;
    double ti = time_us([&] {
        ++epoch;
        hipLaunchKernelGGL(k_chain<true>, dim3(2 * N + 1), dim3(512), 0, s, N, idx, src, img, npix, n, part, result, sy, epoch);
        ++evals;
    }, iters);
    printf("ONE launch, roles chained, acquire fence (buffer_inv sc1) + plain loads %6.2f us per evaluation\n", ti);
    return 0;
}
