// Run ON THE GPU BOX (built in the build container: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/upload tools/ubench/upload.hip).
// A block in pinned host memory on its way to device memory in front of a kernel that needs it: hipMemcpyAsync against a copy KERNEL that
// reads the pinned block, by size -- the host's time per call and the device's time per (copy + dependent kernel) pair.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ void consumer(const unsigned* data, unsigned* out) { if (threadIdx.x == 0) out[0] = data[0] + 1u; }

int main() {
    const size_t sizes[] = { 256, 4096, 32768, 262144, 1048576, 4194304 };
    void* pinned; CK(hipHostMalloc(&pinned, 4194304, hipHostMallocDefault));
    void* dev; CK(hipMalloc(&dev, 4194304));
    unsigned* out; CK(hipMalloc(reinterpret_cast<void**>(&out), 64));
    void* pinned_dev; CK(hipHostGetDevicePointer(&pinned_dev, pinned, 0));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 200;
    for (size_t bytes : sizes) {
        for (int mode = 0; mode < 2; mode++) {
            const unsigned blocks = (unsigned)((bytes / 16 + 255) / 256 < 1024 ? (bytes / 16 + 255) / 256 : 1024);
            for (int warm = 0; warm < 2; warm++) {
                CK(hipStreamSynchronize(s));
                CK(hipEventRecord(e0, s));
                const auto t0 = std::chrono::steady_clock::now();
                for (int r = 0; r < reps; r++) {
                    if (mode == 0) CK(hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, s));
                    else hipLaunchKernelGGL(copy_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, static_cast<const uint4*>(pinned_dev), static_cast<uint4*>(dev), bytes / 16);
                    hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, s, static_cast<const unsigned*>(dev), out);
                }
                const auto t1 = std::chrono::steady_clock::now();
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (warm == 1)
                    printf("%8zu B  %-14s host %6.2f us per (copy + kernel) call pair, device %6.2f us per pair\n", bytes, mode == 0 ? "hipMemcpyAsync" : "copy kernel",
                           std::chrono::duration<double, std::micro>(t1 - t0).count() / reps, ms * 1e3 / reps);
            }
        }
    }
    return 0;
}
