// Diagnostic: what HW_REG_LDS_ALLOC reports for the two co-resident blocks of a CU (70 KB of dynamic LDS each).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256, 2) void k(unsigned *out) {
    extern __shared__ unsigned char smem[];
    smem[threadIdx.x] = 1;
    const unsigned v = __builtin_amdgcn_s_getreg(6 | (0 << 6) | (11 << 11));
    const unsigned all = __builtin_amdgcn_s_getreg(6 | (0 << 6) | (31 << 11));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = v; out[2 * blockIdx.x + 1] = all; }
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(100);   // keep the first round resident
}
int main() {
    unsigned *d, h[1024];
    hipMalloc(&d, sizeof h);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k, dim3(512), dim3(256), 70176, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int nz = 0;
    for (int i = 0; i < 512; ++i) nz += h[2 * i] != 0;
    printf("blocks with LDS base != 0: %d of 512; samples:", nz);
    for (int i = 0; i < 8; ++i) printf(" %u/0x%x", h[2 * i], h[2 * i + 1]);
    printf("\n");
    return 0;
}
