// Vector-memory gather micro-benchmark for gfx950: what does ONE wave64 global load instruction cost in the texture path (address
// unit TA, L1 = TCP, data return TD) when every lane has its own address inside an L1-resident window?  The cone trace issues four
// such loads per SDF sample (hlsl_math.hpp, sample_distance_field); profiles/r02_summary.md reads TA 79 % / TD 92 % busy on cfg5.
//   forms:  dword | dwordx2 (8-byte aligned) | dwordx3 at a 2-byte aligned address | dwordx4 (8- / 16-byte aligned)
//   masks:  all 64 lanes | 16 lanes (one per quad) | 16 lanes (four whole quads) | 1 lane -- does the cost follow the active QUADS?
//   spread: each lane its own texel | the whole wave on two texels (cfg5's coarse field)
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/gather tools/ubench/gather.hip && tools/ubench/gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef const char __attribute__((address_space(1))) gbyte;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int FORM> struct Load;
template <> struct Load<0> { static __device__ __forceinline__ uint32_t at(gbyte* p) { typedef const uint32_t __attribute__((address_space(1), aligned(2))) T; return *(T*)p; } };
template <> struct Load<1> { static __device__ __forceinline__ uint32_t at(gbyte* p) { typedef const u32x2 __attribute__((address_space(1), aligned(8))) T; const u32x2 v = *(T*)p; return v.x ^ v.y; } };
template <> struct Load<2> { static __device__ __forceinline__ uint32_t at(gbyte* p) { typedef const u32x3 __attribute__((address_space(1), aligned(2))) T; const u32x3 v = *(T*)p; return v.x ^ v.z; } };
template <> struct Load<3> { static __device__ __forceinline__ uint32_t at(gbyte* p) { typedef const u32x4 __attribute__((address_space(1), aligned(8))) T; const u32x4 v = *(T*)p; return v.x ^ v.y ^ v.z ^ v.w; } };
template <> struct Load<4> { static __device__ __forceinline__ uint32_t at(gbyte* p) { typedef const uint32_t __attribute__((address_space(1), aligned(4))) T; return *(T*)p; } };

// window: bytes of the (per-CU shared) L1-resident window; offsets are generated per lane and iteration with an LCG, aligned to `align`
// and shifted by `skew` bytes (2 for the 2-byte aligned forms).  spread 0: per-lane addresses; 1: the wave sits on two adjacent texels.
template <int FORM>
__global__ __launch_bounds__(256) void gather_kernel(const char* base_, uint32_t* out, int iters, uint32_t window, uint32_t align, uint32_t skew,
                                                      unsigned long long mask, int spread) {
    gbyte* base = (gbyte*)base_;
    const unsigned lane = threadIdx.x & 63u;
    uint32_t state = (spread ? (threadIdx.x >> 6) : threadIdx.x) * 2654435761u + blockIdx.x * 40503u + 12345u;
    uint32_t acc[4] = { 0, 0, 0, 0 };
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                state = state * 1664525u + 1013904223u;
                uint32_t off = ((state >> 8) % window) & ~(align - 1u);
                if (spread) off += (lane & 1u) * 8u;
                acc[u] ^= Load<FORM>::at(base + off + skew);
            }
        }
    }
    const uint32_t s = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
    if (s == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int FORM>
static void run(const char* name, const char* d_base, uint32_t* d_out, uint32_t window, uint32_t align, uint32_t skew, unsigned long long mask, int spread,
                const char* what) {
    const int cus = 256, waves_per_simd = 8, iters = 2000;
    const int blocks = cus * waves_per_simd;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(gather_kernel<FORM>, dim3(blocks), dim3(256), 0, 0, d_base, d_out, iters, window, align, skew, mask, spread);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(gather_kernel<FORM>, dim3(blocks), dim3(256), 0, 0, d_base, d_out, iters, window, align, skew, mask, spread);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_cu = (double)waves_per_simd * 4.0 * iters * 4.0;
    printf("%-26s %-34s %8.3f ms  %7.2f ns per wave-load per CU  (%5.1f cycles at 2.35 GHz)\n", name, what, ms, ms * 1e6 / instr_per_cu,
           ms * 1e6 / instr_per_cu * 2.35);
}

int main() {
    char* d_base; uint32_t* d_out;
    CK(hipMalloc(&d_base, 64 << 20)); CK(hipMemset(d_base, 1, 64 << 20));
    CK(hipMalloc(&d_out, 64 << 20));
    const unsigned long long all = ~0ull, one_per_quad = 0x1111111111111111ull, four_quads = 0xFFFFull, one = 1ull, half_quads = 0x00000000FFFFFFFFull;
    const uint32_t W = 16384;   // L1-resident window
    run<4>("dword a4", d_base, d_out, W, 8, 0, all, 0, "64 lanes, own texel");
    run<0>("dword a2 (+2)", d_base, d_out, W, 8, 2, all, 0, "64 lanes, own texel");
    run<1>("dwordx2 a8", d_base, d_out, W, 8, 0, all, 0, "64 lanes, own texel");
    run<2>("dwordx3 a2 (+2)", d_base, d_out, W, 8, 2, all, 0, "64 lanes, own texel");
    run<3>("dwordx4 a8", d_base, d_out, W, 8, 0, all, 0, "64 lanes, own texel");
    run<3>("dwordx4 a16", d_base, d_out, W, 16, 0, all, 0, "64 lanes, own texel");
    run<4>("dword a4", d_base, d_out, W, 8, 0, all, 1, "64 lanes on two texels");
    run<0>("dword a2 (+2)", d_base, d_out, W, 8, 2, all, 1, "64 lanes on two texels");
    run<2>("dwordx3 a2 (+2)", d_base, d_out, W, 8, 2, all, 1, "64 lanes on two texels");
    run<3>("dwordx4 a8", d_base, d_out, W, 8, 0, all, 1, "64 lanes on two texels");
    run<4>("dword a4", d_base, d_out, W, 8, 0, half_quads, 0, "32 lanes = 8 whole quads");
    run<4>("dword a4", d_base, d_out, W, 8, 0, four_quads, 0, "16 lanes = 4 whole quads");
    run<4>("dword a4", d_base, d_out, W, 8, 0, one_per_quad, 0, "16 lanes, one per quad");
    run<4>("dword a4", d_base, d_out, W, 8, 0, one, 0, "1 lane");
    run<3>("dwordx4 a8", d_base, d_out, W, 8, 0, four_quads, 0, "16 lanes = 4 whole quads");
    run<3>("dwordx4 a8", d_base, d_out, W, 8, 0, one_per_quad, 0, "16 lanes, one per quad");
    // a window that misses L1 but sits in L2 (the atlas is 25 MB: L2 / Infinity-Cache resident)
    run<4>("dword a4", d_base, d_out, 32u << 20, 8, 0, all, 0, "64 lanes, 32 MB window");
    run<3>("dwordx4 a8", d_base, d_out, 32u << 20, 8, 0, all, 0, "64 lanes, 32 MB window");
    return 0;
}
