// VALU issue-rate micro-benchmark (round 6, design input for the "VALU diet" of the big-segment K1 / K3): how many wave
// instructions per clock and SIMD do the instruction forms the event kernels are made of sustain at 8 waves per SIMD --
// v_fma_f32 against v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (two fp32 lanes per issue), and the conversion / integer forms
// (v_cvt_rpi_i32_f32, v_floor_f32, v_med3_f32, v_mad_u32_u24, v_bfe_u32, v_cvt_f32_u32, v_pk_max_u16, DPP moves).
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench_valu.hip -o tools/microbench_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float float2_v __attribute__((ext_vector_type(2)));

// 16 independent chains per thread, ITER iterations of 16 instructions of form MODE
template <int MODE>
__global__ void __launch_bounds__(512, 8) k_valu(float *out, int iters, float seed) {
    float x[16];
    float2_v p[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = seed + (float)(threadIdx.x + i);
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = float2_v{x[2 * i], x[2 * i + 1]};
    const float c0 = seed * 0.5f, c1 = seed * 0.25f;
    const float2_v pc0 = {c0, c1}, pc1 = {c1, c0};
    const unsigned long long mask = 0x5555555555555555ull ^ (unsigned long long)iters;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c0), "v"(c1));
            if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i & 7]) : "v"(pc0), "v"(pc1));
            if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(pc0));
            if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(pc0));
            if (MODE == 4) asm volatile("v_cvt_rpi_i32_f32 %0, %0" : "+v"(x[i]));
            if (MODE == 5) asm volatile("v_floor_f32 %0, %0" : "+v"(x[i]));
            if (MODE == 6) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c0), "v"(c1));
            if (MODE == 7) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c0), "v"(c1));
            if (MODE == 8) asm volatile("v_bfe_u32 %0, %0, 4, 4" : "+v"(x[i]));
            if (MODE == 9) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x[i]));
            if (MODE == 10) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(x[i]) : "v"(c0));
            if (MODE == 11) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
            if (MODE == 12) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(c0));
            if (MODE == 13) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x[i]));
            if (MODE == 14) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c0), "v"(c1));
            if (MODE == 15) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0] op_sel_hi:[0,1,1]" : "+v"(p[i & 7]) : "v"(pc0), "v"(pc1));
            if (MODE == 16) asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(x[i]) : "v"(c0), "v"(c1));
            if (MODE == 17) asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(x[i]) : "v"(c0));
            if (MODE == 18) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(c0));
            if (MODE == 19) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i & 7]) : "s"(pc0), "v"(pc1));
            if (MODE == 20) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c0));
            if (MODE == 21) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c0));
            if (MODE == 22) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(x[i]) : "v"(c0));
            if (MODE == 23) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(c0), "v"(c1));
            if (MODE == 24) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c0));
            if (MODE == 25) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i]) : "v"(c0));
            if (MODE == 26) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(c0));
            if (MODE == 27) asm volatile("v_lshlrev_b32 %0, 4, %0" : "+v"(x[i]));
            if (MODE == 28) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c0), "s"(mask));
            if (MODE == 29) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x[i]), "v"(c0) : "vcc");
            if (MODE == 30) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[i]) : "v"(c0));
            if (MODE == 31) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "s"(c0), "v"(c1));
            if (MODE == 32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "s"(c0));
            if (MODE == 33) { if (i & 1) asm volatile("v_bfe_u32 %0, %0, 4, 4" : "+v"(x[i])); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c0), "v"(c1)); }
            if (MODE == 34) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i & 3]) : "v"(c0), "v"(c1));  // 4 chains only: dependent every 4th
            if (MODE == 35) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c0), "v"(c1));
            if (MODE == 36) asm volatile("v_fract_f32 %0, %0" : "+v"(x[i]));
            if (MODE == 37) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(c0));
            if (MODE == 38) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(c0));
            if (MODE == 39) asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(x[i]) : "v"(c0));
            if (MODE == 40) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(x[i]) : "s"(c0));
            if (MODE == 41) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(x[i]));
            if (MODE == 42) asm volatile("v_rndne_f32 %0, %0" : "+v"(x[i]));
            if (MODE == 43) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[i]) : "s"(c0), "v"(c1));
            if (MODE == 44) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x[i]) : "v"(c0), "v"(c1));
            if (MODE == 45) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c0), "v"(c1));
            if (MODE == 46) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(x[i]));
            if (MODE == 47) asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(x[i]));
            if (MODE == 48) asm volatile("v_ldexp_f32 %0, %0, 3" : "+v"(x[i]));
            if (MODE == 49) asm volatile("v_mad_u32_u16 %0, %0, %1, %2 op_sel:[1,0,0,0]" : "+v"(x[i]) : "s"(c0), "v"(c1));
            if (MODE == 50) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "+v"(x[i]) : "v"(c1));
            if (MODE == 51) asm volatile("v_cvt_flr_i32_f32 %0, %0" : "+v"(x[i]));
            // floor by a round-down add of 1.5 * 2^23 between two mode switches, two adds per switch pair (the warp's x and y)
            if (MODE == 52 && (i & 1) == 0) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2\n\tv_add_f32 %0, 0x4b400000, %0\n\tv_add_f32 %1, 0x4b400000, %1\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0" : "+v"(x[i]), "+v"(x[i + 1]));
            if (MODE == 53) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(x[i]) : "v"(c0));
            if (MODE == 54) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(x[i]) : "s"(c0), "v"(c1));
            if (MODE == 55) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x[i]) : "s"(c0));
            if (MODE == 56) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c0), "v"(c1));
            if (MODE == 57) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(x[i]));
            if (MODE == 58) asm volatile("v_lshrrev_b32 %0, 8, %0" : "+v"(x[i]));
            if (MODE == 59) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c0), "v"(c1));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    if (s == -1.2345f) out[0] = s;
}

int main(int argc, char **argv) {
    float *out;
    if (hipMalloc(&out, 1024) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int blocks = 1024 * 4, iters = argc > 1 ? atoi(argv[1]) : 512;  // 1024 workgroups of 8 waves = 8 waves on each of the 1024 SIMDs; 4 rounds
    auto run = [&](const char *name, auto kern) {
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, out, iters, 1.5f);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        const int reps = 5;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, out, iters, 1.5f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double wave_insts = (double)blocks * 8 * iters * 16;
        const double t = ms / reps * 1e-3;
        // cycles per wave instruction and SIMD at 2.4 GHz, 1024 SIMDs
        printf("%-44s %9.1f us  %7.3f cycles / wave instruction / SIMD (2.4 GHz)\n", name, t * 1e6, t * 2.4e9 * 1024 / wave_insts);
    };
    run("v_fma_f32", k_valu<0>);
    run("v_pk_fma_f32", k_valu<1>);
    run("v_pk_fma_f32 op_sel broadcast", k_valu<15>);
    run("v_pk_fma_f32 one SGPR-pair source", k_valu<19>);
    run("v_pk_mul_f32", k_valu<2>);
    run("v_pk_add_f32", k_valu<3>);
    run("v_cvt_rpi_i32_f32", k_valu<4>);
    run("v_cvt_i32_f32", k_valu<13>);
    run("v_floor_f32", k_valu<5>);
    run("v_med3_f32", k_valu<6>);
    run("v_max3_f32 |abs| modifiers", k_valu<16>);
    run("v_mad_u32_u24", k_valu<7>);
    run("v_mul_lo_u32", k_valu<12>);
    run("v_bfe_u32", k_valu<8>);
    run("v_cvt_f32_u32", k_valu<9>);
    run("v_pk_max_u16", k_valu<10>);
    run("v_mov_b32 dpp row_shr:1", k_valu<11>);
    run("v_add3_u32", k_valu<14>);
    run("v_lshl_or_b32", k_valu<17>);
    run("v_cndmask_b32", k_valu<18>);
    run("v_cndmask_b32_e64 (SGPR-pair mask)", k_valu<28>);
    run("v_cmp_lt_f32 vcc", k_valu<29>);
    run("v_mul_f32", k_valu<20>);
    run("v_mul_f32 SGPR operand", k_valu<32>);
    run("v_mul_f32 inline 0.5", k_valu<47>);
    run("v_add_f32", k_valu<21>);
    run("v_add_f32 inline 1.0", k_valu<46>);
    run("v_sub_f32", k_valu<22>);
    run("v_sub_f32 SGPR operand", k_valu<40>);
    run("v_fmac_f32", k_valu<23>);
    run("v_fma_f32 SGPR operand", k_valu<31>);
    run("v_fma_f32 4 chains (dependent every 4th)", k_valu<34>);
    run("v_max_f32", k_valu<24>);
    run("v_min3_f32", k_valu<35>);
    run("v_fract_f32", k_valu<36>);
    run("v_rndne_f32", k_valu<42>);
    run("v_ldexp_f32", k_valu<48>);
    run("v_cvt_f32_i32", k_valu<41>);
    run("v_mov_b32", k_valu<37>);
    run("v_and_b32", k_valu<25>);
    run("v_xor_b32", k_valu<38>);
    run("v_add_u32", k_valu<26>);
    run("v_lshlrev_b32", k_valu<27>);
    run("v_lshl_add_u32", k_valu<39>);
    run("v_mul_u32_u24", k_valu<30>);
    run("v_mad_u32_u24 SGPR operand", k_valu<43>);
    run("v_bfi_b32", k_valu<44>);
    run("v_perm_b32", k_valu<45>);
    run("alternating v_fma_f32 / v_bfe_u32", k_valu<33>);
    run("v_mad_u32_u16 op_sel hi, SGPR factor", k_valu<49>);
    run("v_add_u32_sdwa src1 WORD_0", k_valu<50>);
    run("v_cvt_flr_i32_f32", k_valu<51>);
    run("round-down v_add_f32 x2 between 2 s_setreg (per add)", k_valu<52>);
    run("v_pk_min_u16", k_valu<53>);
    run("v_mad_i32_i24 SGPR factor", k_valu<54>);
    run("v_sub_u32 SGPR", k_valu<55>);
    run("v_and_or_b32", k_valu<56>);
    run("v_cvt_f32_ubyte0", k_valu<57>);
    run("v_lshrrev_b32 8", k_valu<58>);
    run("v_med3_i32", k_valu<59>);
    return 0;
}
