// Diagnostic: which SIMD of its CU each wave of a 512-thread block runs on (HW_ID bits 5:4).  The two-group exact
// k-NN kernel expects waves w and w + 4 of a block to share a SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(512) probe(unsigned* out) {
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
}
int main() {
    unsigned* d; const int blocks = 6;
    if (hipMalloc(&d, blocks * 8 * sizeof(unsigned)) != hipSuccess) { printf("no device\n"); return 1; }
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, d);
    unsigned h[blocks * 8];
    if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
    for (int b = 0; b < blocks; ++b) {
        printf("block %d:", b);
        for (int w = 0; w < 8; ++w) printf("  w%d simd %u cu %u", w, (h[b * 8 + w] >> 4) & 3, (h[b * 8 + w] >> 8) & 15);
        printf("\n");
    }
    hipFree(d);
    return 0;
}
