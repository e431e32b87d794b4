// Micro-benchmark that sized the IWE accumulation design (DESIGN.md "K1"): throughput of
//   (a) scattered device-scope fp32 atomics into a 260x346 image (what a naive vote does)
//   (b) the same with workgroup-scope atomics (resolved in the XCD's L2)  -- NOT a correct IWE, rate only
//   (c) coalesced device-scope atomics (what an LDS-tile flush does)
//   (d) LDS fp32 atomics with random addresses in a 64x64 window
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/microbench_atomics.hip -o /tmp/mb
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

template <int SCOPE>
__global__ void __launch_bounds__(256) k_scatter(float *img, int npix, int per_thread) {
    unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < per_thread; ++i) {
        unsigned p = hash(t * 131u + i) % (unsigned)npix;
        if (SCOPE == 0) unsafeAtomicAdd(&img[p], 1.0f);
        else __hip_atomic_fetch_add(&img[p], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// neighbouring lanes hit neighbouring pixels (4 votes of one event style: same 2x2 quad locality)
__global__ void __launch_bounds__(256) k_scatter_local(float *img, int W, int H, int per_thread) {
    unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    // each block owns a 16x16 tile; lanes scatter within tile + 16 px halo
    int ntc = W / 16, tile = blockIdx.x % (ntc * (H / 16));
    int r0 = (tile / ntc) * 16, c0 = (tile % ntc) * 16;
    for (int i = 0; i < per_thread; ++i) {
        unsigned h = hash(t * 131u + i);
        int r = r0 + (int)(h % 30u), c = c0 + (int)((h >> 8) % 30u);
        if (r < H && c < W) unsafeAtomicAdd(&img[r * W + c], 1.0f);
    }
}

__global__ void __launch_bounds__(256) k_coalesced(float *img, int npix, int per_thread) {
    unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned n = gridDim.x * blockDim.x;
    for (int i = 0; i < per_thread; ++i) {
        unsigned p = (t + (unsigned)i * n) % (unsigned)npix;
        unsafeAtomicAdd(&img[p], 1.0f);
    }
}

__global__ void __launch_bounds__(256) k_lds(float *out, int per_thread, int win) {
    extern __shared__ float w[];
    for (int i = threadIdx.x; i < win; i += blockDim.x) w[i] = 0.f;
    __syncthreads();
    unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < per_thread; ++i) {
        unsigned p = hash(t * 131u + i) % (unsigned)win;
        unsafeAtomicAdd(&w[p], 1.0f);
    }
    __syncthreads();
    float s = 0;
    for (int i = threadIdx.x; i < win; i += blockDim.x) s += w[i];
    if (s == -1.f) out[0] = s;
}

int main() {
    const int H = 260, W = 346, npix = H * W;
    float *img, *out;
    CK(hipMalloc(&img, npix * sizeof(float) * 8));
    CK(hipMalloc(&out, 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch, double atomics) {
        for (int i = 0; i < 3; ++i) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        const int reps = 20;
        for (int i = 0; i < reps; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.2f us/launch  %8.2f G atomics/s\n", name, ms / reps * 1e3, atomics / (ms / reps * 1e-3) / 1e9);
        return 0;
    };
    const int blocks = 2048, per = 8;  // 2048*256*8 = 4.19 M atomics
    const double na = (double)blocks * 256 * per;
    run("scatter device-scope (4.2M, 260x346)", [&] { hipLaunchKernelGGL(k_scatter<0>, dim3(blocks), dim3(256), 0, 0, img, npix, per); }, na);
    run("scatter workgroup-scope (L2) (4.2M)", [&] { hipLaunchKernelGGL(k_scatter<1>, dim3(blocks), dim3(256), 0, 0, img, npix, per); }, na);
    run("scatter device-scope tile-local 30x30", [&] { hipLaunchKernelGGL(k_scatter_local, dim3(blocks), dim3(256), 0, 0, img, W, H, per); }, na);
    run("coalesced device-scope (4.2M)", [&] { hipLaunchKernelGGL(k_coalesced, dim3(blocks), dim3(256), 0, 0, img, npix, per); }, na);
    run("coalesced device-scope (0.26M, flush-like)", [&] { hipLaunchKernelGGL(k_coalesced, dim3(512), dim3(256), 0, 0, img, npix, 2); }, 512.0 * 256 * 2);
    for (int win : {1024, 4096, 16384}) {
        char nm[96];
        snprintf(nm, sizeof nm, "LDS atomics random in %d-float window (16.8M)", win);
        run(nm, [&] { hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), win * sizeof(float), 0, out, 32, win); }, (double)blocks * 256 * 32);
    }
    run("empty-ish launch (1 block)", [&] { hipLaunchKernelGGL(k_coalesced, dim3(1), dim3(64), 0, 0, img, npix, 1); }, 64);
    return 0;
}
