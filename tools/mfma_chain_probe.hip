// Diagnostic: rate of v_mfma_f32_32x32x2_f32 against the number of independent accumulator chains per wave and the
// number of waves per SIMD, register operands only.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int NCH>
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters, float a0, float b0) {
    floatx16 acc[NCH];
    for (int t = 0; t < NCH; ++t) for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 64 / NCH; ++u)
#pragma unroll
            for (int t = 0; t < NCH; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < NCH; ++t) for (int v = 0; v < 16; ++v) s += acc[t][v];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NCH> void run(float* d, int wpb) {
    const int blocks = 256 * wpb, iters = 5000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<NCH><<<blocks, 256>>>(d, 100, 1.f, 2.f); hipDeviceSynchronize();
    hipEventRecord(e0); mfma_loop<NCH><<<blocks, 256>>>(d, iters, 1.f, 2.f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 64 * 4096.0;
    printf("%d chain(s), %d wave(s) per SIMD: %.1f ms, %.1f TFLOP/s\n", NCH, wpb, ms, flop / ms / 1e9);
}
int main() {
    float* d; if (hipMalloc(&d, 4096 * 256 * 4) != hipSuccess) return 1;
    for (int wpb : {1, 2}) { run<1>(d, wpb); run<2>(d, wpb); run<4>(d, wpb); }
    return 0;
}
