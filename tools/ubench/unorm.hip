// Does the texture path's own UNORM16 -> f32 conversion (typed buffer load, DATA_FORMAT 16_16 / 16_16_16_16, NUM_FORMAT unorm) return
// the correctly rounded x / 65535 for every code, at element-aligned and at 2-byte-aligned lane offsets?  And what does a typed gather cost
// next to the untyped one?  (The cone trace's unorm16 sampler decodes eight channels per sample with 4 VALU instructions each.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/unorm tools/ubench/unorm.hip && tools/ubench/unorm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// word 3 of a gfx9 buffer resource: DST_SEL_X/Y/Z/W [11:0] (4 = R, 5 = G, 6 = B, 7 = A, 0 = zero), NUM_FORMAT [14:12] (0 = unorm, 7 = float),
// DATA_FORMAT [18:15] (5 = 16_16, 12 = 16_16_16_16)
constexpr int kWord3_1616_unorm = 4 | (5 << 3) | (0 << 12) | (5 << 15);
constexpr int kWord3_16161616_unorm = 4 | (5 << 3) | (6 << 6) | (7 << 9) | (0 << 12) | (12 << 15);
constexpr int kWord3_1616_float = 4 | (5 << 3) | (7 << 12) | (5 << 15);

__device__ __forceinline__ f32x2 load_xy(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    f32x2 v;
    asm volatile("buffer_load_format_xy %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(off), "s"(r) : "memory");
    return v;
}
__device__ __forceinline__ f32x4 load_xyzw(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    f32x4 v;
    asm volatile("buffer_load_format_xyzw %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(off), "s"(r) : "memory");
    return v;
}

// codes[i] = i (uint16), 65536 + 8 entries.  out[0..65535] = .x of the 16_16 load at byte offset 2 i (every second one is 2-byte aligned
// only), out2 = .y of the same load (code i + 1), out4 = the four channels of the 16_16_16_16 load at 8-byte aligned offsets
__global__ void convert_kernel(const uint16_t* codes, float* out_x, float* out_y, float* out4, float* out_half) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)codes, 0, (65536 + 8) * 2, kWord3_1616_unorm);
    __amdgpu_buffer_rsrc_t r4 = __builtin_amdgcn_make_buffer_rsrc((void*)codes, 0, (65536 + 8) * 2, kWord3_16161616_unorm);
    __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc((void*)codes, 0, (65536 + 8) * 2, kWord3_1616_float);
    const f32x2 v = load_xy(r2, i * 2u);
    out_x[i] = v.x; out_y[i] = v.y;
    const f32x2 h = load_xy(rh, i * 2u);
    out_half[i] = h.x;
    if ((i & 3u) == 0u) {
        const f32x4 q = load_xyzw(r4, i * 2u);
        out4[i] = q.x; out4[i + 1] = q.y; out4[i + 2] = q.z; out4[i + 3] = q.w;
    }
}

// throughput: gather of 4-byte elements at random 2-byte-aligned offsets inside an L1-resident window, typed (format_xy) vs untyped dword
template <int TYPED>
__global__ __launch_bounds__(256) void gather_kernel(const char* base, float* out, int iters, uint32_t window) {
    __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)window + 64, TYPED ? kWord3_1616_unorm : 0x00020000);
    uint32_t state = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    float acc = 0.0f;
    for (int i = 0; i < iters; i++) {
        f32x2 v[4];
        uint32_t w[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            state = state * 1664525u + 1013904223u;
            const uint32_t off = ((state >> 8) % window) & ~1u;
            if (TYPED) asm volatile("buffer_load_format_xy %0, %1, %2, 0 offen" : "=v"(v[u]) : "v"(off), "s"(r2) : "memory");
            else asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(w[u]) : "v"(off), "s"(r2) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 4; u++) acc += TYPED ? (v[u].x + v[u].y) : __uint_as_float(w[u]);
    }
    if (acc == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    const int n = 65536;
    uint16_t* h = (uint16_t*)malloc((n + 8) * 2);
    for (int i = 0; i < n + 8; i++) h[i] = (uint16_t)i;
    uint16_t* d; float *dx, *dy, *d4, *dh;
    CK(hipMalloc(&d, (n + 8) * 2)); CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&dy, n * 4)); CK(hipMalloc(&d4, (n + 4) * 4)); CK(hipMalloc(&dh, n * 4));
    CK(hipMemcpy(d, h, (n + 8) * 2, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(convert_kernel, dim3(n / 256), dim3(256), 0, 0, d, dx, dy, d4, dh);
    CK(hipDeviceSynchronize());
    float* x = (float*)malloc(n * 4); float* y = (float*)malloc(n * 4); float* q = (float*)malloc(n * 4); float* hh = (float*)malloc(n * 4);
    CK(hipMemcpy(x, dx, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y, dy, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(q, d4, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hh, dh, n * 4, hipMemcpyDeviceToHost));
    int bad_x_even = 0, bad_x_odd = 0, bad_y = 0, bad_q = 0, shown = 0;
    for (int i = 0; i < n; i++) {
        const float want = (float)i / 65535.0f;
        uint32_t a, b; memcpy(&a, &x[i], 4); memcpy(&b, &want, 4);
        if (a != b) { ((i & 1) ? bad_x_odd : bad_x_even)++; if (shown++ < 8) printf("  code %d at offset %d: typed load %.9g (%08x), x/65535 %.9g (%08x)\n", i, 2 * i, x[i], a, want, b); }
        const float want_y = (float)((i + 1) & 0xFFFF) / 65535.0f;
        if (i + 1 < n && y[i] != want_y) bad_y++;
        if (q[i] != want) bad_q++;
    }
    printf("16_16 unorm .x vs RN(x/65535): %d mismatches at 4-byte aligned offsets, %d at 2-byte aligned offsets; .y: %d; 16_16_16_16 (8-byte aligned): %d\n",
           bad_x_even, bad_x_odd, bad_y, bad_q);
    // f16 format: exact conversion expected; report how many differ from the software conversion
    int bad_h = 0;
    for (int i = 0; i < n; i++) {
        _Float16 f; uint16_t c = (uint16_t)i; memcpy(&f, &c, 2);
        const float want = (float)f;
        if (memcmp(&want, &hh[i], 4) != 0 && !(want != want && hh[i] != hh[i])) { if (bad_h < 4) printf("  half code %04x: typed %.9g  cvt %.9g\n", i, hh[i], want); bad_h++; }
    }
    printf("16_16 float .x vs (float)half: %d mismatches (NaN payloads aside)\n", bad_h);

    // throughput
    const uint32_t window = 8192;
    char* dbase; CK(hipMalloc(&dbase, window + 4096)); CK(hipMemset(dbase, 1, window + 4096));
    float* dout; CK(hipMalloc(&dout, 1 << 22));
    for (int typed = 0; typed < 2; typed++) {
        const int blocks = 256 * 8, iters = 2000;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            if (typed) hipLaunchKernelGGL(gather_kernel<1>, dim3(blocks), dim3(256), 0, 0, dbase, dout, iters, window);
            else hipLaunchKernelGGL(gather_kernel<0>, dim3(blocks), dim3(256), 0, 0, dbase, dout, iters, window);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double loads = (double)blocks * 4 * iters * 4;
        printf("%s gather, 2-byte aligned 4-byte elements, 8 KB window: %.3f ms, %.1f G wave-loads/s, %.1f cycles per wave-load per CU at 2.4 GHz\n",
               typed ? "typed 16_16 unorm (format_xy)" : "untyped dword", ms, loads / (ms * 1e-3) / 1e9, 256.0 * 2.4e9 * ms * 1e-3 / loads);
    }
    return 0;
}
