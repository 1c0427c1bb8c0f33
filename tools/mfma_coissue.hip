// Diagnostic: how a wave's ordinary vector work and a v_mfma_f32_32x32x2_f32 chain of another wave on the same SIMD
// slow each other.  One block of 8 waves per CU: waves 0-3 run an MFMA chain, waves 4-7 (one per SIMD, beside an MFMA
// wave) run a fixed number of dependent-free FMAs.  Prints both groups' own durations, each also measured alone.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int NCH>
__global__ void __launch_bounds__(512) coissue(float* out, unsigned long long* ticks, int mfma_iters, int valu_iters, float a0, float b0) {
    const int wave = threadIdx.x >> 6;
    const unsigned long long t0 = clock64();
    float s = 0.f;
    if (wave >= 4) {
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = a0 + i + threadIdx.x;
        for (int i = 0; i < valu_iters; ++i)
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = __builtin_fmaf(x[u], b0, 1.0f);
        for (int i = 0; i < 8; ++i) s += x[i];
    } else {
        floatx16 acc[NCH];
        for (int c = 0; c < NCH; ++c) for (int v = 0; v < 16; ++v) acc[c][v] = 0.f;
        float a = a0 + threadIdx.x, b = b0;
        for (int i = 0; i < mfma_iters; ++i)
#pragma unroll
            for (int u = 0; u < 16 / NCH; ++u)
#pragma unroll
                for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        for (int c = 0; c < NCH; ++c) for (int v = 0; v < 16; ++v) s += acc[c][v];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    const unsigned long long t1 = clock64();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) ticks[wave] = t1 - t0;
}
template <int NCH> int run(float* d, unsigned long long* t) {
    const int MI = 4000, VI = 4000;     // 64000 MFMAs (64 clocks each), 32000 FMAs
    printf("%d accumulator chain(s) in the MFMA wave\n", NCH);
    for (int mode = 0; mode < 3; ++mode) {
        const int mi = mode == 1 ? 0 : MI, vi = mode == 0 ? 0 : VI;
        coissue<NCH><<<256, 512>>>(d, t, mi, vi, 1.f, 0.5f);
        unsigned long long h[8];
        if (hipMemcpy(h, t, 64, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        printf("%s: MFMA wave %.2f clocks per MFMA, FMA wave %.2f clocks per FMA\n", mode == 0 ? "MFMA alone" : mode == 1 ? "FMA alone " : "both      ",
               mi ? (double)h[0] / (mi * 16.0) : 0.0, vi ? (double)h[4] / (vi * 8.0) : 0.0);
    }
    return 0;
}
int main() {
    float* d; unsigned long long* t;
    if (hipMalloc(&d, 256 * 512 * 4) != hipSuccess || hipMalloc(&t, 64) != hipSuccess) return 1;
    return run<1>(d, t) || run<2>(d, t) || run<4>(d, t);
}
