// VALU issue-rate calibration for gfx950: how many wave64 fp32 VALU instructions per second does the chip retire when every SIMD is
// fed by W resident waves of independent v_fma_f32 chains?  (The "bound: valu" rooflines in bench.py price against this.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu tools/ubench/valu.hip && tools/ubench/valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int CHAINS>
__global__ __launch_bounds__(256) void fma_kernel(float* out, int iters, float a, float b) {
    float v[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) v[c] = (float)threadIdx.x + (float)c;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(a), "v"(b));
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += v[c];
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;     // never true; keeps the chains alive
}

// the same chains as v_pk_fma_f32 (two fp32 FMAs per lane and instruction): does a packed instruction cost one issue slot or two?
template <int CHAINS>
__global__ __launch_bounds__(256) void pk_fma_kernel(float* out, int iters, float a, float b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 v[CHAINS];
    const f2 a2 = {a, a}, b2 = {b, b};
#pragma unroll
    for (int c = 0; c < CHAINS; c++) v[c] = f2{(float)threadIdx.x + (float)c, (float)c};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(a2), "v"(b2));
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += v[c].x + v[c].y;
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CHAINS>
static void run_pk(int waves_per_simd, int iters) {
    float* out;
    (void)hipMalloc(&out, 1 << 20);
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(pk_fma_kernel<CHAINS>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(pk_fma_kernel<CHAINS>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * 4.0 * (double)iters * 8.0 * CHAINS;
    printf("v_pk_fma_f32: chains %d  waves/SIMD %d  %.3f ms  %.1f G wave-instr/s  (%.1f G fp32 FMA lanes x 64 /s)\n", CHAINS, waves_per_simd, ms,
           wave_instr / (ms * 1e-3) / 1e9, 2.0 * wave_instr / (ms * 1e-3) / 1e9);
    (void)hipFree(out);
}

template <int CHAINS>
static void run(int waves_per_simd, int iters) {
    float* out;
    (void)hipMalloc(&out, 1 << 20);
    const int cus = 256;
    const int blocks = cus * waves_per_simd;           // 256 threads = 4 waves = one wave per SIMD of a CU
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(fma_kernel<CHAINS>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(fma_kernel<CHAINS>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * 4.0 * (double)iters * 8.0 * CHAINS;
    printf("chains %d  waves/SIMD %d  %.3f ms  %.1f G wave-instr/s  (%.2f per SIMD-cycle at 2.4 GHz)\n", CHAINS, waves_per_simd, ms,
           wave_instr / (ms * 1e-3) / 1e9, wave_instr / (ms * 1e-3) / (1024.0 * 2.4e9));
    (void)hipFree(out);
}

int main() {
    for (int w : {1, 2, 4, 8}) run<8>(w, 20000);
    for (int w : {1, 2, 4, 8}) run<1>(w, 20000);
    for (int w : {2, 4, 8}) run_pk<8>(w, 20000);
    return 0;
}
