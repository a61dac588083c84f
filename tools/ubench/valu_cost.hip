// Run ON THE GPU BOX (built in the build container: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_cost tools/ubench/valu_cost.hip).
// Issue cost per SIMD of the vector instructions the cone trace's loop is made of: eight waves per SIMD, four independent chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>

#define KERNEL(NAME, I0, I1, I2, I3)                                                                          \
__global__ __launch_bounds__(256) void NAME(float* out, int iters, float seed) {                             \
    float a = seed + threadIdx.x, b = a + 1.0f, c = a + 2.0f, d = a + 3.0f;                                   \
    const float m = 1.0000001f, k = 0.5f;                                                                     \
    unsigned ua = threadIdx.x, ub = ua + 1, uc = ua + 2, ud = ua + 3;                                         \
    for (int i = 0; i < iters; i++) {                                                                         \
        _Pragma("unroll") for (int u = 0; u < 16; u++) {                                                       \
            asm volatile(I0 : "+v"(a), "+v"(ua) : "v"(m), "v"(k) : "vcc", "s20", "s21");                      \
            asm volatile(I1 : "+v"(b), "+v"(ub) : "v"(m), "v"(k) : "vcc", "s20", "s21");                      \
            asm volatile(I2 : "+v"(c), "+v"(uc) : "v"(m), "v"(k) : "vcc", "s20", "s21");                      \
            asm volatile(I3 : "+v"(d), "+v"(ud) : "v"(m), "v"(k) : "vcc", "s20", "s21");                      \
        }                                                                                                     \
    }                                                                                                         \
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + (float)(ua + ub + uc + ud);                         \
}
#define SAME(NAME, I) KERNEL(NAME, I, I, I, I)

SAME(k_fma, "v_fma_f32 %0, %0, %2, %3")
SAME(k_fmac, "v_fmac_f32 %0, %0, %2")
SAME(k_mul, "v_mul_f32 %0, %0, %2")
SAME(k_sub, "v_sub_f32 %0, %0, %3")
SAME(k_mix, "v_fma_mix_f32 %0, %0, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,0,1]")
SAME(k_fract, "v_fract_f32 %0, %0")
SAME(k_cvt_i, "v_cvt_i32_f32 %1, %0")
SAME(k_cvt_f16, "v_cvt_f32_f16 %0, %1")
SAME(k_mad24, "v_mad_u32_u24 %1, %1, %1, %1")
SAME(k_lshl_add, "v_lshl_add_u32 %1, %1, 4, %1")
SAME(k_min3, "v_minimum3_f32 %0, %0, %2, %3")
SAME(k_minf, "v_min_f32 %0, %0, %2")
SAME(k_cmp, "v_cmp_nge_f32 vcc, %0, %2")
SAME(k_rcp, "v_rcp_f32 %0, %0")
SAME(k_sqrt, "v_sqrt_f32 %0, %0")
SAME(k_fixup, "v_div_fixup_f32 %0, %0, %2, %3")
SAME(k_fmaak, "v_fmaak_f32 %0, %0, %2, 0x3ea8f5c3")
SAME(k_mul_lo, "v_mul_lo_u32 %1, %1, %1")
SAME(k_mul_i24, "v_mul_i32_i24 %1, %1, %1")
SAME(k_readlane, "v_readlane_b32 s20, %0, 3")
SAME(k_cndmask, "v_cndmask_b32 %0, %0, %2, vcc")
SAME(k_salu, "s_add_u32 s20, s20, 1")
SAME(k_nop, "s_nop 0")
KERNEL(k_fma_fract, "v_fma_f32 %0, %0, %2, %3", "v_fract_f32 %0, %0", "v_fma_f32 %0, %0, %2, %3", "v_fract_f32 %0, %0")
KERNEL(k_fma_mix, "v_fma_f32 %0, %0, %2, %3", "v_fma_mix_f32 %0, %0, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,0,1]", "v_fma_f32 %0, %0, %2, %3", "v_fma_mix_f32 %0, %0, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,0,1]")
KERNEL(k_fma_cvt, "v_fmac_f32 %0, %0, %2", "v_cvt_i32_f32 %1, %0", "v_fmac_f32 %0, %0, %2", "v_cvt_i32_f32 %1, %0")
KERNEL(k_fract_mad, "v_fract_f32 %0, %0", "v_mad_u32_u24 %1, %1, %1, %1", "v_fract_f32 %0, %0", "v_mad_u32_u24 %1, %1, %1, %1")
KERNEL(k_sub_mul, "v_sub_f32 %0, %0, %3", "v_mul_f32 %0, %0, %2", "v_sub_f32 %0, %0, %3", "v_mul_f32 %0, %0, %2")
KERNEL(k_3fma_1fract, "v_fmac_f32 %0, %0, %2", "v_fmac_f32 %0, %0, %2", "v_fmac_f32 %0, %0, %2", "v_fract_f32 %0, %0")
KERNEL(k_fma_salu, "v_fma_f32 %0, %0, %2, %3", "s_add_u32 s20, s20, 1", "v_fma_f32 %0, %0, %2, %3", "s_add_u32 s21, s21, 1")
SAME(k_add_u32, "v_add_u32 %1, %1, %1")
SAME(k_lshl, "v_lshlrev_b32 %1, 3, %1")
SAME(k_and, "v_and_b32 %1, 0xffff, %1")
SAME(k_floor, "v_floor_f32 %0, %0")
SAME(k_max, "v_max_f32 %0, %0, %2")
SAME(k_add, "v_add_f32 %0, %0, %2")
SAME(k_mov, "v_mov_b32 %0, %2")

typedef void (*kern_t)(float*, int, float);
static void run(const char* name, kern_t k) {
    static float* out = nullptr;
    if (!out) (void)hipMalloc(&out, 2048 * 256 * sizeof(float));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 5000;
    hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, out, 50, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, out, iters, 1.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = 2048.0 * 4 * (double)iters * 64;
    printf("%-44s %8.3f ms  %7.1f G wave-instructions/s  %5.2f cycles per instruction and SIMD at 2.4 GHz\n", name, ms, wave_instr / (ms * 1e-3) / 1e9,
           (ms * 1e-3) * 2.4e9 / (wave_instr / 1024.0));
}

int main() {
    run("v_fma_f32", k_fma); run("v_fmac_f32", k_fmac); run("v_mul_f32", k_mul); run("v_sub_f32", k_sub);
    run("v_fma_mix_f32", k_mix); run("v_fract_f32", k_fract); run("v_cvt_i32_f32", k_cvt_i); run("v_cvt_f32_f16", k_cvt_f16);
    run("v_mad_u32_u24", k_mad24); run("v_lshl_add_u32", k_lshl_add); run("v_minimum3_f32", k_min3); run("v_min_f32", k_minf);
    run("v_cmp_nge_f32", k_cmp); run("v_rcp_f32", k_rcp); run("v_sqrt_f32", k_sqrt); run("v_div_fixup_f32", k_fixup); run("v_fmaak_f32", k_fmaak);
    run("v_mul_lo_u32", k_mul_lo); run("v_mul_i32_i24", k_mul_i24); run("v_readlane_b32", k_readlane);
    run("v_cndmask_b32", k_cndmask); run("s_add_u32", k_salu); run("s_nop 0", k_nop);
    run("v_add_u32", k_add_u32); run("v_lshlrev_b32", k_lshl); run("v_and_b32", k_and); run("v_floor_f32", k_floor); run("v_max_f32", k_max);
    run("v_add_f32", k_add); run("v_mov_b32", k_mov);
    run("v_fma_f32 + v_fract_f32 (per instr)", k_fma_fract); run("v_fma_f32 + v_fma_mix_f32 (per instr)", k_fma_mix);
    run("v_fmac_f32 + v_cvt_i32_f32 (per instr)", k_fma_cvt); run("v_fract_f32 + v_mad_u32_u24 (per instr)", k_fract_mad);
    run("v_sub_f32 + v_mul_f32 (per instr)", k_sub_mul); run("3 v_fmac_f32 + 1 v_fract_f32 (per instr)", k_3fma_1fract);
    run("v_fma_f32 + s_add_u32 (per instr)", k_fma_salu);
    return 0;
}
