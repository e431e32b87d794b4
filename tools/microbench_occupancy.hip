// Residency census (round 6): how many 512-thread workgroups does a CU of gfx950 hold at once as a function of their LDS allocation, VGPRs and
// SGPRs?  Every workgroup stamps the wall clock at entry and spins ~30 us; the workgroups that start within 2 us of the first are the resident
// set.  Result (profiles/r06_microbench_occupancy.txt): 4 per CU up to 40 KB of LDS with <= 64 VGPRs and <= 80 SGPRs; THREE with 72 VGPRs or
// ~90 SGPRs, whatever the LDS.  (Why it was written: tools/timeline.py showed K3 at three per CU -- its -DCMAX_TIMELINE build takes 83 SGPRs /
// 66 VGPRs in K3; the production kernels stay at <= 80 / <= 64, tools/resources.sh.)
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench_occupancy.hip -o tools/microbench_occupancy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
extern __shared__ int s_dyn[];
template <int V, int S>
__global__ void __launch_bounds__(512) k_census(unsigned long long *start, int spin_ticks) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) start[blockIdx.x] = t0;
    s_dyn[threadIdx.x] = (int)t0;
    if (V == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    if (V == 72) asm volatile("v_mov_b32 v71, 0" ::: "v71");
    if (V == 56) asm volatile("v_mov_b32 v55, 0" ::: "v55");
    if (S == 70) asm volatile("s_mov_b32 s68, 0" ::: "s68");
    if (S == 90) asm volatile("s_mov_b32 s88, 0" ::: "s88");
    __syncthreads();
    while (wall_clock64() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
    if (s_dyn[(threadIdx.x + 1) & 511] == 0x7fffffff) start[0] = 0;
}
template <int V, int S>
void run(const char *name, unsigned long long *d, int nwg) {
    std::vector<unsigned long long> h(nwg);
    (void)hipFuncSetAttribute((const void *)k_census<V, S>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int sizes[] = {25504, 32768, 33696, 33704, 33736, 33952, 34816, 36864, 40960};
    for (int lds : sizes) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL((k_census<V, S>), dim3(nwg), dim3(512), lds, 0, d, 3000);
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(h.data(), d, nwg * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        const unsigned long long first = *std::min_element(h.begin(), h.end());
        int resident = 0;
        for (auto t : h) resident += (t - first) <= 200 ? 1 : 0;
        printf("%s LDS %6d: %4d resident = %.2f per CU\n", name, lds, resident, resident / 256.0);
    }
}
int main() {
    const int nwg = 2048;
    unsigned long long *d;
    if (hipMalloc(&d, nwg * sizeof(unsigned long long)) != hipSuccess) return 1;
    run<0, 0>("vgpr small, sgpr small", d, nwg);
    run<56, 70>("vgpr 56, sgpr ~70   ", d, nwg);
    run<64, 70>("vgpr 64, sgpr ~70   ", d, nwg);
    run<64, 90>("vgpr 64, sgpr ~90   ", d, nwg);
    run<72, 70>("vgpr 72, sgpr ~70   ", d, nwg);
    return 0;
}
