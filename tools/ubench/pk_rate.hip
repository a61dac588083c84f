// Run ON THE GPU BOX (built in the build container: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/pk_rate tools/ubench/pk_rate.hip).
// Issue rate of v_fma_f32 against v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 with eight waves per SIMD: does a packed f32 instruction
// cost one issue slot (two results per slot) or two?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void spin(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, a4 = a0 + 4.0f, a5 = a0 + 5.0f, a6 = a0 + 6.0f, a7 = a0 + 7.0f;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float m = 1.0000001f, c = 0.5f;
    const f32x2 m2 = {m, m}, c2 = {c, c};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (MODE == 0) {           // 8 scalar fma = 8 results
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(m), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(m), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(m), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a4) : "v"(m), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a5) : "v"(m), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a6) : "v"(m), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a7) : "v"(m), "v"(c));
            } else if (MODE == 1) {    // 4 packed fma = 8 results
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(m2), "v"(c2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(m2), "v"(c2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(m2), "v"(c2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(m2), "v"(c2));
            } else if (MODE == 2) {    // 4 packed add
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p0) : "v"(c2));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p1) : "v"(c2));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p2) : "v"(c2));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p3) : "v"(c2));
            } else if (MODE == 3) {    // 4 packed mul
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p0) : "v"(m2));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p1) : "v"(m2));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p2) : "v"(m2));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p3) : "v"(m2));
            } else {                   // 4 scalar fma (the same instruction count as the packed modes)
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(m), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(m), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(m), "v"(c));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
static void run(const char* name, int instr_per_iter) {
    float* out; hipMalloc(&out, 2048 * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    spin<MODE><<<2048, 256>>>(out, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    spin<MODE><<<2048, 256>>>(out, iters, 1.0f);      // 2048 workgroups of 4 waves = 8 waves per SIMD on 256 CUs
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = 2048.0 * 4 * (double)iters * 8 * instr_per_iter;
    printf("%-28s %8.3f ms  %7.1f G wave-instructions/s  (%.2f cycles per instruction and SIMD at 2.4 GHz)\n", name, ms, wave_instr / (ms * 1e-3) / 1e9,
           (ms * 1e-3) * 2.4e9 / (wave_instr / 1024.0));
    hipFree(out);
}

int main() {
    run<0>("8 x v_fma_f32", 8);
    run<4>("4 x v_fma_f32", 4);
    run<1>("4 x v_pk_fma_f32", 4);
    run<2>("4 x v_pk_add_f32", 4);
    run<3>("4 x v_pk_mul_f32", 4);
    return 0;
}
