// FETCH_SIZE calibration for gfx950 (MI355X_MICROARCH.md, "HBM": the counter tallies 128-byte requests at 64 bytes, so a wide coalesced
// streaming read reports HALF its bytes -- "other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern").  The particle rasteriser's tile kernel gathers 64-byte sprite records by index (raster.hip: a.sprites[key & 0xFFFFFFFF]);
// r03-r05 doubled its FETCH_SIZE like a streaming read's and read 2.06 x the bytes of the records it touches.  This program reads a known
// byte count in three patterns over a 2 GiB array (past the 256 MiB Infinity Cache), one kernel each, for `rocprofv3 --pmc FETCH_SIZE`:
//   stream_kernel     every lane 16 B, consecutive lanes consecutive addresses: N x 16 B
//   records_kernel    every lane ONE 64-byte record (four 16-byte loads) at a shuffled index, each record once: N x 64 B
//   records8_kernel   every lane one 8-byte key, consecutive (the sorted keys of the rasteriser): N x 8 B
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/fetch_calib tools/ubench/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ src, uint32_t* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint4 v = src[i];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) out[0] = 1;
}
__global__ __launch_bounds__(256) void records_kernel(const uint4* __restrict__ src, uint32_t* out, size_t n_records, uint32_t mul, uint32_t mask) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_records) return;
    const size_t r = ((size_t)((uint32_t)i * mul) & mask);          // an odd multiplier modulo a power of two: a permutation of the records
    const uint4* p = src + 4 * r;
    const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    if ((a.x ^ b.y ^ c.z ^ d.w) == 0x12345678u) out[0] = 1;
}
__global__ __launch_bounds__(256) void records8_kernel(const uint2* __restrict__ src, uint32_t* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint2 v = src[i];
    if ((v.x ^ v.y) == 0x12345678u) out[0] = 1;
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    char* d = nullptr; uint32_t* out = nullptr;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(d, 1, bytes)); CK(hipMemset(out, 0, 64));
    const size_t n16 = bytes / 16, n64 = bytes / 64, n8 = bytes / 8;
    hipLaunchKernelGGL(stream_kernel, dim3((unsigned)(n16 / 256)), dim3(256), 0, 0, (const uint4*)d, out, n16);
    hipLaunchKernelGGL(records_kernel, dim3((unsigned)(n64 / 256)), dim3(256), 0, 0, (const uint4*)d, out, n64, 2654435761u, (uint32_t)(n64 - 1));
    hipLaunchKernelGGL(records8_kernel, dim3((unsigned)(n8 / 256)), dim3(256), 0, 0, (const uint2*)d, out, n8);
    CK(hipDeviceSynchronize());
    printf("bytes read by each kernel: %zu (2 GiB): stream_kernel 16 B per lane, records_kernel one shuffled 64-byte record per lane, records8_kernel 8 B per lane\n", bytes);
    CK(hipFree(d)); CK(hipFree(out));
    return 0;
}
