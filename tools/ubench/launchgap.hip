// Micro-benchmark: time per back-to-back dependent launch on one stream as a function of the kernarg size (the particle step passes
// a ~3.9 KB launch descriptor by value) and of the grid (1 block vs 4352 blocks of 256 threads that do nothing).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int N> struct Arg { int v[N]; };
template <int N>
__global__ __launch_bounds__(256) void k(Arg<N> a, int* out) {
    if (a.v[N - 1] == 12345 && threadIdx.x == 0) out[blockIdx.x] = a.v[0];
}
template <int N>
void run(int* d, int blocks) {
    Arg<N> a; for (int i = 0; i < N; i++) a.v[i] = i;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int R = 2000;
    for (int r = 0; r < R + 20; r++) {
        if (r == 20) CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<N>, dim3(blocks), dim3(256), 0, 0, a, d);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("kernarg %4d B, %5d blocks: %.2f us per launch\n", (int)sizeof(a), blocks, ms / R * 1e3);
}
int main() {
    int* d; CK(hipMalloc(&d, 1 << 20));
    for (int blocks : {1, 4352}) { run<4>(d, blocks); run<64>(d, blocks); run<256>(d, blocks); run<1000>(d, blocks); }
    return 0;
}
