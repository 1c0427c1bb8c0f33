// Latency of dependent random row reads as a function of the footprint they are spread over: one wavefront follows a
// chain of pseudo-random rows (each read depends on the one before); the time per read shows where address translation
// stops being covered by the TLBs.  Diagnostic tool: hipcc --offload-arch=gfx950 -O3 tools/tlb_probe.hip -o tools/bin/tlb_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void chase(const float4* __restrict__ buf, unsigned long long rows, unsigned row_chunks, int iters, unsigned* out, unsigned long long seed) {
    const int lane = threadIdx.x & 63;
    unsigned long long x = seed + blockIdx.x * 0x9E3779B97F4A7C15ull + (threadIdx.x >> 6) * 0xD1B54A32D192ED03ull;
    unsigned acc = 0;
    for (int i = 0; i < iters; ++i) {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        const unsigned long long r = (x >> 20) % rows;
        const float4 v = buf[r * row_chunks + lane];
        const unsigned w = __float_as_uint(v.x);
        acc += w;
        x += __builtin_amdgcn_readfirstlane(w) & 1u;     // the next address depends on the data
    }
    if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = acc;
}

int main(int argc, char** argv) {
    const unsigned row_bytes = argc > 1 ? atoi(argv[1]) : 3840;
    const unsigned row_chunks = row_bytes / 16;
    unsigned* d_out; CK(hipMalloc(&d_out, 1 << 20));
    for (double gb : {0.25, 0.5, 1.0, 2.0, 4.0, 8.0, 16.0, 32.0, 64.0}) {
        const unsigned long long bytes = (unsigned long long)(gb * (1ull << 30));
        float4* buf; if (hipMalloc(&buf, bytes) != hipSuccess) { printf("%.2f GiB: allocation failed\n", gb); break; }
        CK(hipMemset(buf, 0, bytes));
        const unsigned long long rows = bytes / row_bytes;
        for (int blocks : {1, 256, 1024, 4096}) {
            const int iters = blocks == 1 ? 2000 : 400;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            hipLaunchKernelGGL(chase, dim3(blocks), dim3(64), 0, 0, buf, rows, row_chunks, 50, d_out, 1ull);
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(chase, dim3(blocks), dim3(64), 0, 0, buf, rows, row_chunks, iters, d_out, 7ull);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%6.2f GiB, %4d wavefronts: %7.0f ns per dependent 1-KiB read, %8.1f GB/s in 1-KiB reads\n", gb, blocks, ms * 1e6 / iters, (double)blocks * iters * 1024 / (ms * 1e-3) / 1e9);
        }
        CK(hipFree(buf));
    }
    return 0;
}
