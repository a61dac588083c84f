// Run ON THE GPU BOX, several processes at once: is a blocking hipMemcpy (pageable host memory -> device, null stream) ordered in front of a
// kernel launched right after it on a hipStreamNonBlocking stream?  (r06: fuzz sweeps with 24 processes on one GPU saw bursts of particle
// steps that read garbage from the engine's randomness table -- uploaded by hipMemcpy in ilm_engine_create, read by kernels on the
// context's non-blocking stream.)   hipcc --offload-arch=gfx950 -O2 tools/ubench/memcpy_order.hip -o tools/ubench/memcpy_order
//   tools/ubench/memcpy_order <iterations> <mode: 0 = hipMemcpy, 1 = hipMemcpyAsync on the kernel's stream + sync, 2 = hipMemcpy + hipDeviceSynchronize>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
__global__ void poison(uint32_t* p, size_t n, uint32_t v) { for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) p[i] = v; }
// wrong[0] = words that differ, wrong[1] = of them the poison value, wrong[2] = of them the PREVIOUS iteration's word, wrong[3] = lowest wrong index + 1
__global__ void count_wrong(const uint32_t* p, size_t n, uint32_t seed, uint32_t prev_seed, uint32_t poison_value, unsigned long long* wrong) {
    unsigned long long w = 0, wp = 0, wo = 0, first = ~0ull;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
        const uint32_t v = p[i];
        if (v != (uint32_t)(i * 2654435761u + seed)) { w++; wp += (v == poison_value); wo += (v == (uint32_t)(i * 2654435761u + prev_seed)); if (i < first) first = i; }
    }
    if (w) { atomicAdd(wrong, w); atomicAdd(wrong + 1, wp); atomicAdd(wrong + 2, wo); atomicMin(wrong + 3, first + 1); }
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 1000, mode = argc > 2 ? atoi(argv[2]) : 0;
    const size_t n = (size_t)807 * 653 * 4;                     // the randomness table: 807 x 653 float4 = 8.4 MB
    hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned long long* d_wrong; CHECK(hipMalloc(&d_wrong, 32));
    unsigned long long tp = 0, to = 0;
    std::vector<uint32_t> host(n);
    long bad_iters = 0; unsigned long long bad_words = 0;
    for (int it = 0; it < iters; it++) {
        uint32_t* d; CHECK(hipMalloc(&d, n * 4));
        hipLaunchKernelGGL(poison, dim3(1024), dim3(256), 0, s, d, n, 0xFFC00000u + (uint32_t)it);      // what a freed allocation may have left
        CHECK(hipStreamSynchronize(s));
        const uint32_t seed = (uint32_t)it * 977u + 13u;
        for (size_t i = 0; i < n; i++) host[i] = (uint32_t)(i * 2654435761u + seed);
        if (mode == 1) { CHECK(hipMemcpyAsync(d, host.data(), n * 4, hipMemcpyHostToDevice, s)); CHECK(hipStreamSynchronize(s)); }
        else { CHECK(hipMemcpy(d, host.data(), n * 4, hipMemcpyHostToDevice)); if (mode == 2) CHECK(hipDeviceSynchronize()); }
        const unsigned long long init[4] = { 0, 0, 0, ~0ull };
        CHECK(hipMemcpyAsync(d_wrong, init, 32, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(count_wrong, dim3(1024), dim3(256), 0, s, d, n, seed, (uint32_t)(it - 1) * 977u + 13u, 0xFFC00000u + (uint32_t)it, d_wrong);
        unsigned long long w4[4] = { 0, 0, 0, 0 }; CHECK(hipMemcpyAsync(w4, d_wrong, 32, hipMemcpyDeviceToHost, s)); CHECK(hipStreamSynchronize(s));
        if (w4[0]) { bad_iters++; bad_words += w4[0]; tp += w4[1]; to += w4[2]; if (bad_iters <= 2) printf("  iteration %d: %llu wrong words from index %llu: %llu poison, %llu the previous iteration's\n", it, w4[0], w4[3] - 1, w4[1], w4[2]); }
        CHECK(hipFree(d));
    }
    printf("mode %d: %ld of %d iterations read words the copy had not delivered (%llu words: %llu poison, %llu the previous iteration's)\n", mode, bad_iters, iters, bad_words, tp, to);
    return 0;
}
