// What a grid-wide wait costs against the kernel boundary it would replace, with the grids of cfg2's evaluation (VERDICT r4 #5a:
// "same workgroup votes, hands over, gathers" in one launch).  Every phase is the shortest memory chain a real kernel has (index load
// -> data load -> store: two dependent round trips); the barrier is the XCD-hierarchical one of the guide's price list (per-XCD
// arrival counter, the XCD's last arriver goes on to a top counter and releases its XCD through a generation word; one polling
// lane per workgroup, every spin bounded).
//   A  two launches (G x 512, G x 512)                 B  one launch: phase, grid barrier, phase
//   (B without any fence = the floor of the wait itself; it publishes nothing)
//   C  three launches (G x 512, G x 512, 1 x 256)      D  one launch: phase, barrier, phase, fan-in to workgroup 0, phase by workgroup 0
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_gridbarrier.hip -o tools/microbench_gridbarrier
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

struct Barrier {           // one 128-byte line per word that is polled or added to
    unsigned xcd[8][32];
    unsigned gen[8][32];
    unsigned top[32];
    unsigned err[32];
};

__device__ __forceinline__ unsigned poll(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// all threads of all workgroups call it; epoch counts the barriers since the counters were cleared (monotonic counters: no reset race)
// MODE 0: arrive + poll only (publishes nothing: the floor of the wait).  MODE 1: every workgroup release-fences before it arrives and
// acquire-fences after.  MODE 2: the price list's recipe -- every thread drains its own stores, only the XCD's last arriver
// release-fences (its write-back covers the L2 all workgroups of the XCD stored into), everybody acquire-fences.  MODE 3: as 2 without the
// acquire of the non-leaders (what a consumer that reads the published words with sc1 loads would pay).
template <int MODE>
__device__ void grid_barrier(Barrier *b, unsigned epoch, unsigned nwg) {
    if (MODE >= 2) __builtin_amdgcn_s_waitcnt(0);   // vmcnt/lgkmcnt; stores of gfx9 count in vmcnt
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned x = blockIdx.x & 7u, mine = (nwg - x + 7u) / 8u;
        if (MODE == 1) __atomic_thread_fence(__ATOMIC_RELEASE);
        const unsigned old = __hip_atomic_fetch_add(&b->xcd[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        if (old + 1u == mine * epoch) {
            if (MODE >= 2) __atomic_thread_fence(__ATOMIC_RELEASE);
            __hip_atomic_fetch_add(&b->top[0], 1u, MODE == 1 ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (poll(&b->top[0]) < 8u * epoch && ++spins < (1 << 16)) __builtin_amdgcn_s_sleep(1);
            if (MODE >= 2) __atomic_thread_fence(__ATOMIC_ACQUIRE);
            __hip_atomic_store(&b->gen[x][0], epoch, MODE == 1 ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (poll(&b->gen[x][0]) < epoch && ++spins < (1 << 16)) __builtin_amdgcn_s_sleep(1);
            if (MODE == 2) __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        if (spins >= (1 << 16)) b->err[0] = 1u;
        if (MODE == 1) __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
}

__device__ __forceinline__ void touch(const int *__restrict__ idx, const float *src, float *dst, int n, int salt) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x + salt) % n;
    const float v = __builtin_nontemporal_load(src + idx[i]);   // not a stale line of an earlier phase
    dst[i] = v + 1.f;
}

__global__ void k_phase(const int *__restrict__ idx, const float *src, float *dst, int n, int salt) { touch(idx, src, dst, n, salt); }

extern __shared__ float s_pad[];
template <int MODE>
__global__ __launch_bounds__(512) void k_two_phases(const int *__restrict__ idx, float *a, float *bb, int n, Barrier *bar, unsigned epoch) {
    if (threadIdx.x == 9999) s_pad[0] = 0.f;
    touch(idx, a, bb, n, 0);
    grid_barrier<MODE>(bar, epoch, gridDim.x);
    touch(idx, bb, a, n, 64);
}
__global__ __launch_bounds__(512) void k_three_phases(const int *__restrict__ idx, float *a, float *bb, int n, Barrier *bar, unsigned epoch, unsigned *fan) {
    if (threadIdx.x == 9999) s_pad[0] = 0.f;
    touch(idx, a, bb, n, 0);
    grid_barrier<2>(bar, epoch, gridDim.x);
    touch(idx, bb, a, n, 64);
    // fan-in: everybody arrives, only workgroup 0 waits
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&fan[(blockIdx.x & 7u) * 32u], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (blockIdx.x == 0) {
            int spins = 0;
            for (unsigned x = 0; x < 8u; ++x) {
                const unsigned mine = (gridDim.x - x + 7u) / 8u;
                while (poll(&fan[x * 32u]) < mine * epoch && ++spins < (1 << 16)) __builtin_amdgcn_s_sleep(1);
            }
            if (spins >= (1 << 16)) bar->err[0] = 2u;
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
    }
    if (blockIdx.x != 0) return;
    __syncthreads();
    if (threadIdx.x < 256) touch(idx, a, bb, n, 128);
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

int main() {
    const int n = 1 << 20;
    int *idx;
    float *a, *b;
    Barrier *bar;
    unsigned *fan;
    hipMalloc(&idx, n * sizeof(int));
    hipMalloc(&a, n * sizeof(float));
    hipMalloc(&b, n * sizeof(float));
    hipMalloc(&bar, sizeof(Barrier));
    hipMalloc(&fan, 8 * 32 * sizeof(unsigned));
    hipMemset(idx, 0, n * sizeof(int));
    hipMemset(a, 0, n * sizeof(float));
    hipMemset(b, 0, n * sizeof(float));
    hipMemset(bar, 0, sizeof(Barrier));
    hipStream_t s;
    hipStreamCreate(&s);
    const int iters = 2000;
    const size_t lds = 24 * 1024;  // cfg2's windows: what decides how many workgroups a CU holds
    int per_cu = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_three_phases, 512, lds);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("%s: %d CUs, %d workgroups of 512 threads + 24 KB LDS per CU resident\n", prop.name, prop.multiProcessorCount, per_cu);
    for (int G : {256, 512, 704, 1000}) {
        if (G > per_cu * prop.multiProcessorCount) {
            printf("G = %d: not resident, skipped\n", G);
            continue;
        }
        // eager launches (the host may bound these) and the same sequences replayed from a hipGraph, 8 evaluations per graph
        // (every graph starts by clearing the barrier words, so that the epochs 1..8 baked into it are right on every replay)
        double t[2][8];
        unsigned errs = 0;
        for (int graph = 0; graph < 2; ++graph) {
            for (int v = 0; v < 8; ++v) {
                unsigned epoch = 0;
                auto body = [&] {
                    ++epoch;
                    if (v == 0 || v == 2) {
                        hipLaunchKernelGGL(k_phase, dim3(G), dim3(512), lds, s, idx, a, b, n, 0);
                        hipLaunchKernelGGL(k_phase, dim3(G), dim3(512), lds, s, idx, b, a, n, 64);
                        if (v == 2) hipLaunchKernelGGL(k_phase, dim3(1), dim3(256), 0, s, idx, a, b, n, 128);
                    } else if (v == 1) {
                        hipLaunchKernelGGL(k_two_phases<1>, dim3(G), dim3(512), lds, s, idx, a, b, n, bar, epoch);
                    } else if (v == 3) {
                        hipLaunchKernelGGL(k_three_phases, dim3(G), dim3(512), lds, s, idx, a, b, n, bar, epoch, fan);
                    } else if (v == 5) {
                        hipLaunchKernelGGL(k_two_phases<0>, dim3(G), dim3(512), lds, s, idx, a, b, n, bar, epoch);
                    } else if (v == 6) {
                        hipLaunchKernelGGL(k_two_phases<2>, dim3(G), dim3(512), lds, s, idx, a, b, n, bar, epoch);
                    } else if (v == 7) {
                        hipLaunchKernelGGL(k_two_phases<3>, dim3(G), dim3(512), lds, s, idx, a, b, n, bar, epoch);
                    } else {
                        hipLaunchKernelGGL(k_phase, dim3(G), dim3(512), lds, s, idx, a, b, n, 0);
                    }
                };
                auto clear = [&] {
                    hipMemsetAsync(bar, 0, sizeof(Barrier) - sizeof(bar->err), s);
                    hipMemsetAsync(fan, 0, 8 * 32 * sizeof(unsigned), s);
                };
                if (!graph) {
                    clear();
                    t[0][v] = time_us(body, iters);
                } else {
                    hipGraph_t g;
                    hipGraphExec_t ge;
                    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
                    clear();
                    for (int r = 0; r < 8; ++r) body();
                    hipStreamEndCapture(s, &g);
                    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
                    t[1][v] = time_us([&] { hipGraphLaunch(ge, s); }, iters / 8) / 8;
                    hipGraphExecDestroy(ge);
                    hipGraphDestroy(g);
                }
            }
        }
        Barrier hb;
        hipMemcpy(&hb, bar, sizeof(Barrier), hipMemcpyDeviceToHost);
        errs = hb.err[0];
        for (int graph = 0; graph < 2; ++graph)
            printf("G = %4d workgroups, %s: one phase alone %5.2f us | two launches %5.2f | one launch + grid barrier: arrive + poll only %5.2f, "
                   "leader releases / all acquire %5.2f, leader releases / leader acquires %5.2f, every workgroup both fences %5.2f | three "
                   "launches %5.2f, one launch + barrier + fan-in %5.2f   (spin limit hit: %u)\n", G, graph ? "graph x8" : "eager   ",
                   t[graph][4], t[graph][0], t[graph][5], t[graph][6], t[graph][7], t[graph][1], t[graph][2], t[graph][3], errs);
    }
    return 0;
}
