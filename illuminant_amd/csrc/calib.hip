// calib.hip -- NOT part of the product library: the box's own HBM copy rate for bench.py's `roofline.calibrated_peak` (SURVEY 8d asks for
// a calibration run beside the spec figure).  Built into a library of its own (../lib/libilluminant_calib.so); nothing under
// illuminant_amd/ loads it.  A float4 copy kernel in three forms -- plain, non-temporal stores, non-temporal loads and stores -- one
// 16-byte element per lane (256 contiguous bytes... per four lanes: 1 KB per wave and instruction), timed with HIP events on a stream of
// its own; MI355X_MICROARCH.md quotes 6.29 TB/s for such a kernel against the 8 TB/s spec.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float vf4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void copy_kernel(const vf4* __restrict__ src, vf4* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const vf4 v = (MODE == 2) ? __builtin_nontemporal_load(src + i) : src[i];
    if (MODE >= 1) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
}

// bytes are copied `reps` times in each of the three forms; out_gb_per_s[m] = (bytes read + bytes written) / time of form m.  Returns a hipError_t.
extern "C" int ilm_calib_copy_rates(int device, size_t bytes, int reps, double out_gb_per_s[3]) {
#define CAL_TRY(x) do { const hipError_t e_ = (x); if (e_ != hipSuccess) { rc = (int)e_; goto done; } } while (0)
    int rc = 0;
    void *a = nullptr, *b = nullptr;
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const size_t n = bytes / sizeof(vf4);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    CAL_TRY(hipSetDevice(device));
    CAL_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CAL_TRY(hipEventCreate(&e0)); CAL_TRY(hipEventCreate(&e1));
    CAL_TRY(hipMalloc(&a, n * sizeof(vf4))); CAL_TRY(hipMalloc(&b, n * sizeof(vf4)));
    CAL_TRY(hipMemsetAsync(a, 1, n * sizeof(vf4), s)); CAL_TRY(hipMemsetAsync(b, 2, n * sizeof(vf4), s));
    for (int mode = 0; mode < 3; mode++) {
        for (int r = -2; r < reps; r++) {                    // two untimed copies first
            if (r == 0) CAL_TRY(hipEventRecord(e0, s));
            if (mode == 0) hipLaunchKernelGGL(copy_kernel<0>, dim3(blocks), dim3(256), 0, s, (const vf4*)a, (vf4*)b, n);
            else if (mode == 1) hipLaunchKernelGGL(copy_kernel<1>, dim3(blocks), dim3(256), 0, s, (const vf4*)a, (vf4*)b, n);
            else hipLaunchKernelGGL(copy_kernel<2>, dim3(blocks), dim3(256), 0, s, (const vf4*)a, (vf4*)b, n);
        }
        CAL_TRY(hipEventRecord(e1, s));
        CAL_TRY(hipEventSynchronize(e1));
        float ms = 0.0f;
        CAL_TRY(hipEventElapsedTime(&ms, e0, e1));
        out_gb_per_s[mode] = 2.0 * (double)(n * sizeof(vf4)) * (double)reps / ((double)ms * 1e-3) / 1e9;
    }
done:
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (s) (void)hipStreamDestroy(s);
    return rc;
#undef CAL_TRY
}
