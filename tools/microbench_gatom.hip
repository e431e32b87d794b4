// Global-memory atomic micro-benchmark: fp32 vs u32 vs u64 atomics, scattered over a 260x346 image.
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/microbench_gatom.hip -o tools/microbench_gatom
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ unsigned hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

template <int MODE>
__global__ void __launch_bounds__(256) k_scatter(void *img, int npix, int per_thread) {
    unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned h = hash(t);
    for (int i = 0; i < per_thread; ++i) {
        h = h * 1664525u + 1013904223u;
        unsigned p = (h >> 4) % (unsigned)npix;
        if (MODE == 0) unsafeAtomicAdd(&((float *)img)[p], 1.0f);
        if (MODE == 1) atomicAdd(&((unsigned *)img)[p], 3u);
        if (MODE == 2) atomicAdd(&((unsigned long long *)img)[p], 3ull);
        if (MODE == 3) unsafeAtomicAdd(&((double *)img)[p], 1.0);
        if (MODE == 4) {  // 2x2 quad u32
            unsigned *q = (unsigned *)img;
            atomicAdd(&q[p], 3u); atomicAdd(&q[p + 1], 3u); atomicAdd(&q[p + 346], 3u); atomicAdd(&q[p + 347], 3u);
        }
        if (MODE == 5) {  // 2x2 quad u64
            unsigned long long *q = (unsigned long long *)img;
            atomicAdd(&q[p], 3ull); atomicAdd(&q[p + 1], 3ull); atomicAdd(&q[p + 346], 3ull); atomicAdd(&q[p + 347], 3ull);
        }
    }
}

int main() {
    const int npix = 260 * 346;
    void *img;
    if (hipMalloc(&img, (npix + 1024) * 8) != hipSuccess) return 1;
    (void)hipMemset(img, 0, (npix + 1024) * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int blocks = 2048, per = 8;
    auto run = [&](const char *name, auto kern, double mult) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, img, npix, per);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        const int reps = 20;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, img, npix, per);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        double n = (double)blocks * 256 * per * mult;
        printf("%-36s %8.2f us  %8.1f G atomics/s\n", name, ms / reps * 1e3, n / (ms / reps * 1e-3) / 1e9);
    };
    run("global f32 scattered", k_scatter<0>, 1);
    run("global u32 scattered", k_scatter<1>, 1);
    run("global u64 scattered", k_scatter<2>, 1);
    run("global f64 scattered", k_scatter<3>, 1);
    run("global u32 2x2 quads", k_scatter<4>, 4);
    run("global u64 2x2 quads", k_scatter<5>, 4);
    return 0;
}
