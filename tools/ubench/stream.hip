// Micro-benchmark: what does the memory path allow for the particle step's access pattern?
//   mode 0: 12 dword loads + 16 dword stores per lane (one slot per lane, SoA planes)       [the step kernel's pattern]
//   mode 1: same bytes with 16-byte accesses (4 slots per lane)
//   mode 2: AoS float4: 3 x 16 B loads + 4 x 16 B stores per slot (one slot per lane)
//   mode 3: mode 0 with non-temporal stores      mode 4: mode 0 with non-temporal loads and stores
//   mode 5: two units per wave, both units' loads issued before any arithmetic (half the waves, one generation at N = 1 M)
//   mode 6: persistent waves (8 per SIMD), each walking units with the next unit's loads in flight during the arithmetic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float __attribute__((address_space(1))) gfloat;
typedef float __attribute__((ext_vector_type(4))) vf4;
typedef vf4 __attribute__((address_space(1))) gvf4;

__global__ __launch_bounds__(256) void k_dword(float* b, long S, int alu) {
    gfloat* base = (gfloat*)b;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    float v[12];
#pragma unroll
    for (int c = 0; c < 12; c++) v[c] = base[c * S + i];
    float acc = 0;
#pragma unroll
    for (int c = 0; c < 12; c++) acc += v[c];
    for (int k = 0; k < alu; k++) acc = acc * 1.0001f + 0.5f;
#pragma unroll
    for (int c = 0; c < 8; c++) base[c * S + i] = v[c] + acc;
#pragma unroll
    for (int c = 12; c < 20; c++) base[c * S + i] = acc;
}
template <bool NT_LOAD>
__global__ __launch_bounds__(256) void k_dword_nt(float* b, long S, int alu) {
    gfloat* base = (gfloat*)b;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    float v[12];
#pragma unroll
    for (int c = 0; c < 12; c++) v[c] = NT_LOAD ? __builtin_nontemporal_load(base + c * S + i) : base[c * S + i];
    float acc = 0;
#pragma unroll
    for (int c = 0; c < 12; c++) acc += v[c];
    for (int k = 0; k < alu; k++) acc = acc * 1.0001f + 0.5f;
#pragma unroll
    for (int c = 0; c < 8; c++) __builtin_nontemporal_store(v[c] + acc, base + c * S + i);
#pragma unroll
    for (int c = 12; c < 20; c++) __builtin_nontemporal_store(acc, base + c * S + i);
}
__global__ __launch_bounds__(256) void k_two(float* b, long S, int alu) {
    gfloat* base = (gfloat*)b;
    const long i0 = ((long)blockIdx.x * 256 + (threadIdx.x & ~63u)) * 2 + (threadIdx.x & 63u);
    float v[2][12];
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int c = 0; c < 12; c++) v[u][c] = base[c * S + i0 + u * 64];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        float acc = 0;
#pragma unroll
        for (int c = 0; c < 12; c++) acc += v[u][c];
        for (int k = 0; k < alu; k++) acc = acc * 1.0001f + 0.5f;
        const long i = i0 + u * 64;
#pragma unroll
        for (int c = 0; c < 8; c++) base[c * S + i] = v[u][c] + acc;
#pragma unroll
        for (int c = 12; c < 20; c++) base[c * S + i] = acc;
    }
}
__global__ __launch_bounds__(256) void k_persistent(float* b, long S, int alu, long units) {
    gfloat* base = (gfloat*)b;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), waves = (long)gridDim.x * 4;
    const unsigned lane = threadIdx.x & 63u;
    float nxt[12];
    long u = wave;
    if (u < units) {
#pragma unroll
        for (int c = 0; c < 12; c++) nxt[c] = base[c * S + u * 64 + lane];
    }
    while (u < units) {
        float v[12];
#pragma unroll
        for (int c = 0; c < 12; c++) v[c] = nxt[c];
        const long un = u + waves;
        if (un < units) {
#pragma unroll
            for (int c = 0; c < 12; c++) nxt[c] = base[c * S + un * 64 + lane];
        }
        float acc = 0;
#pragma unroll
        for (int c = 0; c < 12; c++) acc += v[c];
        for (int k = 0; k < alu; k++) acc = acc * 1.0001f + 0.5f;
        const long i = u * 64 + lane;
#pragma unroll
        for (int c = 0; c < 8; c++) base[c * S + i] = v[c] + acc;
#pragma unroll
        for (int c = 12; c < 20; c++) base[c * S + i] = acc;
        u = un;
    }
}
__global__ __launch_bounds__(256) void k_x4(float* b, long S, int alu) {
    gfloat* base = (gfloat*)b;
    long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    vf4 v[12];
#pragma unroll
    for (int c = 0; c < 12; c++) v[c] = *(const gvf4*)(base + c * S + i);
    float acc = 0;
#pragma unroll
    for (int c = 0; c < 12; c++) acc += v[c].x + v[c].y + v[c].z + v[c].w;
    for (int k = 0; k < alu; k++) acc = acc * 1.0001f + 0.5f;
#pragma unroll
    for (int c = 0; c < 8; c++) { vf4 o = v[c]; o.x += acc; *(gvf4*)(base + c * S + i) = o; }
#pragma unroll
    for (int c = 12; c < 20; c++) { vf4 o = {acc, acc, acc, acc}; *(gvf4*)(base + c * S + i) = o; }
}
__global__ __launch_bounds__(256) void k_aos(float4* b, long N, int alu) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    float4 p = b[i], v = b[N + i], a = b[2 * N + i];
    float acc = p.x + v.y + a.z;
    for (int k = 0; k < alu; k++) acc = acc * 1.0001f + 0.5f;
    b[i] = make_float4(p.x + acc, p.y, p.z, p.w); b[N + i] = make_float4(v.x + acc, v.y, v.z, v.w);
    b[3 * N + i] = make_float4(acc, acc, acc, acc); b[4 * N + i] = make_float4(acc, acc, acc, acc);
}
int main(int argc, char** argv) {
    long N = argc > 1 ? atol(argv[1]) : (1 << 20);
    int alu = argc > 2 ? atoi(argv[2]) : 0;
    const long pad = argc > 3 ? atol(argv[3]) : 0;        // floats added to the plane stride (planes a power of two apart share cache sets / channels)
    const long S = N + pad;
    float* d; CK(hipMalloc(&d, sizeof(float) * 20 * S)); CK(hipMemset(d, 0, sizeof(float) * 20 * S));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 7; mode++) {
        const int reps = 50;
        for (int r = 0; r < reps + 5; r++) {
            if (r == 5) CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_dword, dim3(N / 256), dim3(256), 0, 0, d, S, alu);
            else if (mode == 1) hipLaunchKernelGGL(k_x4, dim3(N / 1024), dim3(256), 0, 0, d, S, alu);
            else if (mode == 3) hipLaunchKernelGGL(k_dword_nt<false>, dim3(N / 256), dim3(256), 0, 0, d, N, alu);
            else if (mode == 4) hipLaunchKernelGGL(k_dword_nt<true>, dim3(N / 256), dim3(256), 0, 0, d, N, alu);
            else if (mode == 5) hipLaunchKernelGGL(k_two, dim3(N / 512), dim3(256), 0, 0, d, N, alu);
            else if (mode == 6) hipLaunchKernelGGL(k_persistent, dim3(256 * 8), dim3(256), 0, 0, d, N, alu, N / 64);
            else hipLaunchKernelGGL(k_aos, dim3(N / 256), dim3(256), 0, 0, (float4*)d, N, alu);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("N=%ld pad=%ld alu=%d mode=%d: %.2f us  %.2f TB/s (112 B/slot)\n", N, pad, alu, mode, ms * 1e3, N * 112.0 / ms / 1e9);
    }
    return 0;
}
