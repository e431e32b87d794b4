// Launch floor of cfg2's evaluation structure on MI355X: three DEPENDENT launches on one stream with the grids of
// K1 (704 x 256), K3 (704 x 512) and k_finish_deferred (1 x 256), the kernels doing (a) nothing, (b) one dependent
// global load + one store per thread (the shortest memory chain a real kernel has), timed in steady state.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_launch.hip -o tools/microbench_launch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void k_empty(int *p) {
    if (p == nullptr && threadIdx.x == 12345) p[0] = 0;
}
__global__ void k_touch(const int *__restrict__ idx, const float *__restrict__ src, float *__restrict__ dst, int n) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) % n;
    dst[i] = src[idx[i]] + 1.f;  // segment-load -> event-load -> store: two dependent memory round trips
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
    float *src, *dst;
    hipMalloc(&idx, n * sizeof(int));
    hipMalloc(&src, n * sizeof(float));
    hipMalloc(&dst, n * sizeof(float));
    hipMemset(idx, 0, n * sizeof(int));
    hipMemset(src, 0, n * sizeof(float));
    hipStream_t s;
    hipStreamCreate(&s);
    const int iters = 2000;
    double e1 = time_us([&] { hipLaunchKernelGGL(k_empty, dim3(704), dim3(256), 0, s, (int *)dst); }, iters);
    double e3 = time_us([&] {
        hipLaunchKernelGGL(k_empty, dim3(704), dim3(256), 0, s, (int *)dst);
        hipLaunchKernelGGL(k_empty, dim3(704), dim3(512), 0, s, (int *)dst);
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(256), 0, s, (int *)dst);
    }, iters);
    double t3 = time_us([&] {
        hipLaunchKernelGGL(k_touch, dim3(704), dim3(256), 0, s, idx, src, dst, n);
        hipLaunchKernelGGL(k_touch, dim3(704), dim3(512), 0, s, idx, src, dst, n);
        hipLaunchKernelGGL(k_touch, dim3(1), dim3(256), 0, s, idx, src, dst, n);
    }, iters);
    double t2 = time_us([&] {
        hipLaunchKernelGGL(k_touch, dim3(704), dim3(256), 0, s, idx, src, dst, n);
        hipLaunchKernelGGL(k_touch, dim3(704), dim3(512), 0, s, idx, src, dst, n);
    }, iters);
    // the same triple replayed from a hipGraph (one graph launch per evaluation; 1 and 8 evaluations per graph)
    double g1 = 0, g8 = 0;
    for (int reps : {1, 8}) {
        hipGraph_t g;
        hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int r = 0; r < reps; ++r) {
            hipLaunchKernelGGL(k_touch, dim3(704), dim3(256), 0, s, idx, src, dst, n);
            hipLaunchKernelGGL(k_touch, dim3(704), dim3(512), 0, s, idx, src, dst, n);
            hipLaunchKernelGGL(k_touch, dim3(1), dim3(256), 0, s, idx, src, dst, n);
        }
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        const double t = time_us([&] { hipGraphLaunch(ge, s); }, iters / reps) / reps;
        (reps == 1 ? g1 : g8) = t;
        hipGraphExecDestroy(ge);
        hipGraphDestroy(g);
    }
    printf("one empty launch 704x256, back to back                         %6.2f us per launch\n", e1);
    printf("three dependent EMPTY launches (704x256, 704x512, 1x256)       %6.2f us per triple\n", e3);
    printf("three dependent launches, two dependent memory trips each      %6.2f us per triple\n", t3);
    printf("two dependent launches (704x256, 704x512), same kernels        %6.2f us per pair\n", t2);
    printf("the triple with memory trips replayed from a hipGraph (1 per graph)  %6.2f us per triple\n", g1);
    printf("the triple with memory trips replayed from a hipGraph (8 per graph)  %6.2f us per triple\n", g8);
    return 0;
}
