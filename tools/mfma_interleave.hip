// Diagnostic: a wave's own vector instructions between its v_mfma_f32_32x32x2_f32 (64 clocks each): how many
// independent FMAs fit under one MFMA without slowing the chain.  One wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int U>
__global__ void __launch_bounds__(256) inter(float* out, unsigned long long* ticks, int iters, float a0, float b0) {
    floatx16 acc;
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = a0 + i + threadIdx.x;
    float a = a0 + threadIdx.x, b = b0;
    const unsigned long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = __builtin_fmaf(x[u], b0, 1.0f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = clock64();
    float s = 0.f;
    for (int v = 0; v < 16; ++v) s += acc[v] + x[v];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
template <int U> void run(float* d, unsigned long long* t) {
    const int iters = 2000;
    inter<U><<<256, 256>>>(d, t, iters, 1.f, 0.5f);
    unsigned long long h = 0;
    hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("%2d FMAs after each MFMA: %.1f clocks per MFMA\n", U, (double)h / (iters * 16.0));
}
int main() {
    float* d; unsigned long long* t;
    if (hipMalloc(&d, 256 * 256 * 4) != hipSuccess || hipMalloc(&t, 64) != hipSuccess) return 1;
    run<0>(d, t); run<4>(d, t); run<8>(d, t); run<12>(d, t); run<14>(d, t); run<16>(d, t);
    return 0;
}
