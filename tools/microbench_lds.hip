// LDS accumulate micro-benchmark (design input for K1): rate of ds_add_f32 / ds_add_u32
// with random, conflict-free and same-address patterns, 256-thread workgroups, 2048 workgroups.
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/microbench_lds.hip -o tools/microbench_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ unsigned hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

// MODE 0: random f32, 1: conflict-free f32 (lane -> lane + 64*k), 2: random u32, 3: conflict-free u32,
// 4: same-address f32 (all lanes one address), 5: random f32 but 2x2 quad per "event" (p, p+1, p+W, p+W+1)
// 6: 4-lane duplicates (lanes 4k..4k+3 share an address)   7: plain non-atomic ds_write random (lower bound)
// 8: random u64 (8-byte aligned)   9: u32 2x2 quads (the K1 vote pattern)   10: u64 pairs = 2x2 quad as two 8-byte adds
template <int MODE>
__global__ void __launch_bounds__(256) k_lds(float *out, int per_thread) {
    __shared__ float w[4096 + 128];
    unsigned *wu = reinterpret_cast<unsigned *>(w);
    for (int i = threadIdx.x; i < 4096 + 128; i += blockDim.x) w[i] = 0.f;
    __syncthreads();
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned lane = threadIdx.x & 63u;
    unsigned h = hash(t);
    float acc = 0.f;
    for (int i = 0; i < per_thread; ++i) {
        h = h * 1664525u + 1013904223u;
        unsigned p = (h >> 8) & 4095u;
        if (MODE == 0) unsafeAtomicAdd(&w[p], 1.0f);
        if (MODE == 1) unsafeAtomicAdd(&w[(lane + 64u * (unsigned)i) & 4095u], 1.0f);
        if (MODE == 2) atomicAdd(&wu[p], 3u);
        if (MODE == 3) atomicAdd(&wu[(lane + 64u * (unsigned)i) & 4095u], 3u);
        if (MODE == 4) unsafeAtomicAdd(&w[(unsigned)i & 4095u], 1.0f);
        if (MODE == 5) {
            unsafeAtomicAdd(&w[p], 1.0f);
            unsafeAtomicAdd(&w[p + 1], 1.0f);
            unsafeAtomicAdd(&w[p + 64], 1.0f);
            unsafeAtomicAdd(&w[p + 65], 1.0f);
        }
        if (MODE == 6) unsafeAtomicAdd(&w[hash((t >> 2) * 977u + i) & 4095u], 1.0f);
        if (MODE == 7) w[p] = (float)i;
        if (MODE == 8) atomicAdd(reinterpret_cast<unsigned long long *>(w) + (p >> 1), 0x0000000300000005ull);
        if (MODE == 9) {
            atomicAdd(&wu[p], 3u);
            atomicAdd(&wu[p + 1], 3u);
            atomicAdd(&wu[p + 64], 3u);
            atomicAdd(&wu[p + 65], 3u);
        }
        if (MODE == 11) acc += w[p] + w[p + 1] + w[p + 64] + w[p + 65];  // the K3 gather pattern (plain reads)
        if (MODE == 12) { atomicAdd(&wu[p], 3u); atomicAdd(&wu[p + 1], 3u); atomicAdd(&wu[p + 65], 3u); atomicAdd(&wu[p + 66], 3u); }  // odd row stride
        if (MODE == 10) {
            atomicAdd(reinterpret_cast<unsigned long long *>(w) + (p >> 1), 0x0000000300000005ull);
            atomicAdd(reinterpret_cast<unsigned long long *>(w) + (p >> 1) + 32, 0x0000000300000005ull);
        }
    }
    __syncthreads();
    float s = 0;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) s += w[i];
    if (s + acc == -1.f) out[0] = s;
}

int main(int argc, char **argv) {
    float *out;
    if (hipMalloc(&out, 1024) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    // argv[1]: operations per thread (default 32 = the round-1 figures, launch-dominated: 5-7 us per launch; 2048 gives the sustained rate)
    const int blocks = 2048, per = argc > 1 ? atoi(argv[1]) : 32;
    auto run = [&](const char *name, auto kern, double mult) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, per);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        const int reps = 20;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, per);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        double n = (double)blocks * 256 * per * mult;
        printf("%-40s %8.2f us  %8.1f G ops/s  %6.2f ops/clk/CU\n", name, ms / reps * 1e3, n / (ms / reps * 1e-3) / 1e9,
               n / (ms / reps * 1e-3) / 256 / 2.4e9);
    };
    run("f32 random (4096 window)", k_lds<0>, 1);
    run("f32 conflict-free", k_lds<1>, 1);
    run("u32 random", k_lds<2>, 1);
    run("u32 conflict-free", k_lds<3>, 1);
    run("f32 same address (broadcast)", k_lds<4>, 1);
    run("f32 random 2x2 quads", k_lds<5>, 4);
    run("f32 4-lane duplicates", k_lds<6>, 1);
    run("plain ds_write random", k_lds<7>, 1);
    run("u64 random", k_lds<8>, 1);
    run("u32 random 2x2 quads (per quad)", k_lds<9>, 1);
    run("u64 pairs = 2x2 quad (per quad)", k_lds<10>, 1);
    run("plain reads 2x2 quads (per quad)", k_lds<11>, 1);
    run("u32 2x2 quads, odd row stride (per quad)", k_lds<12>, 1);
    return 0;
}
