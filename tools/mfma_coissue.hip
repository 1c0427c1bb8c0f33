// Diagnostic: does ordinary vector work of a second wave on the same SIMD slow a wave's MFMA chain?
// One block of 8 waves per CU: waves 0-3 run v_mfma_f32_32x32x2_f32 chains, waves 4-7 (one per SIMD,
// beside an MFMA wave) run independent FMAs for at least as long.  Prints the MFMA waves' own rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(512) coissue(float* out, unsigned long long* ticks, int iters, float a0, float b0, int valu_iters) {
    const int wave = threadIdx.x >> 6;
    if (wave >= 4) {
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = a0 + i + threadIdx.x;
        for (int i = 0; i < valu_iters; ++i)
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = __builtin_fmaf(x[u], b0, 1.0f);
        float s = 0.f; for (int i = 0; i < 8; ++i) s += x[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
        return;
    }
    floatx16 acc[2];
    for (int t = 0; t < 2; ++t) for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < 2; ++t) for (int v = 0; v < 16; ++v) s += acc[t][v];
    const unsigned long long t1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + wave] = t1 - t0;
}
int main() {
    float* d; unsigned long long* t; hipMalloc(&d, 256 * 512 * 4); hipMalloc(&t, 1024 * 8);
    const int iters = 20000;
    for (int valu : {0, 1, 2, 4, 8}) {        // valu: vector instructions of the neighbour per MFMA of this wave (MFMA = 64 cycles, FMA = 4)
        const int valu_iters = (int)((long long)iters * 32 * valu / 8 * 3 / 2);   // 1.5x as long as the MFMA chain would need at that ratio
        coissue<<<256, 512>>>(d, t, 100, 1.f, 2.f, 100); hipDeviceSynchronize();
        coissue<<<256, 512>>>(d, t, iters, 1.f, 2.f, valu_iters); hipDeviceSynchronize();
        std::vector<unsigned long long> h(1024); hipMemcpy(h.data(), t, 1024 * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto x : h) avg += (double)x; avg /= 1024;       // 100 MHz ticks
        const double sec = avg * 1e-8, flop_per_wave = (double)iters * 32 * 4096.0;
        printf("neighbour issues ~%d vector instructions per MFMA: MFMA wave takes %.2f ms -> %.1f TFLOP/s if all 1024 SIMDs ran at this rate\n",
               valu, sec * 1e3, flop_per_wave * 1024 / sec / 1e12);
    }
    return 0;
}
