// Diagnostic: v_mfma_f32_32x32x16_bf16 on gfx950 — (1) which element of A, B and D each lane holds, checked against a
// host product; (2) its rate alone and with U independent FMAs of the same wave after each MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ __bf16 tobf(float x) { return (__bf16)x; }
// assumed layout: lane l holds A[i = l % 32][k = 8 * (l / 32) + t], B[k = 8 * (l / 32) + t][j = l % 32], t = 0..7;
// D[v] = C[i = (v / 4) * 8 + (l / 32) * 4 + v % 4][j = l % 32]
__global__ void layout(const float* A, const float* B, float* D) {       // A 32x16, B 16x32 row-major, D 32x32
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int t = 0; t < 8; ++t) { a[t] = tobf(A[(l % 32) * 16 + 8 * (l / 32) + t]); b[t] = tobf(B[(8 * (l / 32) + t) * 32 + l % 32]); }
    floatx16 acc;
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int v = 0; v < 16; ++v) D[((v / 4) * 8 + (l / 32) * 4 + v % 4) * 32 + l % 32] = acc[v];
}
template <int U>
__global__ void __launch_bounds__(256) rate(float* out, unsigned long long* ticks, int iters, float a0, float b0) {
    floatx16 acc;
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    bf16x8 a, b;
    for (int t = 0; t < 8; ++t) { a[t] = tobf(a0 + threadIdx.x + t); b[t] = tobf(b0 + t); }
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = a0 + i + threadIdx.x;
    const unsigned long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
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
// eight waves per block: waves 0-3 stream bf16 MFMAs, waves 4-7 (one beside each on its SIMD) run FMAs; both groups'
// own durations, together and alone
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
        floatx16 acc;
        for (int v = 0; v < 16; ++v) acc[v] = 0.f;
        bf16x8 a, b;
        for (int t = 0; t < 8; ++t) { a[t] = tobf(a0 + threadIdx.x + t); b[t] = tobf(b0 + t); }
        for (int i = 0; i < mfma_iters; ++i)
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        for (int v = 0; v < 16; ++v) s += acc[v];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    const unsigned long long t1 = clock64();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) ticks[wave] = t1 - t0;
}
template <int U> void run(float* d, unsigned long long* t) {
    const int iters = 2000;
    rate<U><<<256, 256>>>(d, t, iters, 1.f, 0.5f);
    unsigned long long h = 0;
    hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("%2d FMAs after each bf16 MFMA: %.1f clocks per MFMA\n", U, (double)h / (iters * 16.0));
}
int main() {
    float hA[32 * 16], hB[16 * 32], hD[32 * 32];
    for (int i = 0; i < 32 * 16; ++i) { hA[i] = (float)((i * 7) % 13 - 6); hB[i] = (float)((i * 5) % 11 - 5); }
    float *dA, *dB, *dD; unsigned long long* t;
    if (hipMalloc(&dA, sizeof hA) != hipSuccess) return 1;
    hipMalloc(&dB, sizeof hB); hipMalloc(&dD, 256 * 256 * 4); hipMalloc(&t, 64);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    layout<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        float s = 0; for (int k = 0; k < 16; ++k) s += hA[i * 16 + k] * hB[k * 32 + j];
        if (fabsf(s - hD[i * 32 + j]) > 1e-3f) ++bad;
    }
    printf("layout check: %d of 1024 elements differ\n", bad);
    for (int mode = 0; mode < 3; ++mode) {
        const int MI = 8000, VI = 8000;            // 128000 MFMAs x 32.8 clocks = 4.2 M clocks; 64000 FMAs
        const int mi = mode == 1 ? 0 : MI, vi = mode == 0 ? 0 : VI;
        coissue<<<256, 512>>>(dD, t, mi, vi, 1.f, 0.5f);
        unsigned long long h[8];
        hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
        printf("%s: MFMA wave %llu clocks, FMA wave %llu clocks\n", mode == 0 ? "bf16 MFMA alone" : mode == 1 ? "FMA alone      " : "both           ", mi ? h[0] : 0ull, vi ? h[4] : 0ull);
    }
    run<0>(dD, t); run<2>(dD, t); run<4>(dD, t); run<6>(dD, t); run<8>(dD, t); run<12>(dD, t);
    return 0;
}
