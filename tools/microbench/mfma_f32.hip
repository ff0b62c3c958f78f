// Microbenchmark: issue rate of the fp32 MFMA instructions the training kernels use, as cycles per instruction per
// wave, for 1 / 2 / 4 independent accumulator chains (one wave per SIMD: 256 threads per block, one block per CU).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_f32.hip -o tools/microbench/mfma_f32 && tools/microbench/mfma_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS, int SHAPE>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters) {
    f32x16 a32[4];
    f32x4 a16[4];
    for (int c = 0; c < 4; ++c) { for (int i = 0; i < 16; ++i) a32[c][i] = 0.f; for (int i = 0; i < 4; ++i) a16[c][i] = 0.f; }
    float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (SHAPE == 32) a32[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a32[c], 0, 0, 0);
                else a16[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a16[c], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < 4; ++c) { for (int i = 0; i < 16; ++i) s += a32[c][i]; for (int i = 0; i < 4; ++i) s += a16[c][i]; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int CHAINS, int SHAPE>
void run(const char *name, int blocks) {
    float *out;
    unsigned long long *cyc, h = 0;
    hipMalloc(&out, blocks * 256 * 4);
    hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipLaunchKernelGGL((k<CHAINS, SHAPE>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<CHAINS, SHAPE>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 8 * CHAINS;
    const double flop = (SHAPE == 32 ? 4096.0 : 2048.0) * n * 4 * blocks;
    printf("%-28s chains %d blocks %4d: %7.1f cycles / MFMA / wave, %7.1f TFLOP/s, %.3f ms\n", name, CHAINS, blocks, (double)h / n, flop / (ms * 1e-3) / 1e12, ms);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int blocks : {1, 256}) {
        run<1, 32>("v_mfma_f32_32x32x2_f32", blocks);
        run<2, 32>("v_mfma_f32_32x32x2_f32", blocks);
        run<4, 32>("v_mfma_f32_32x32x2_f32", blocks);
        run<1, 16>("v_mfma_f32_16x16x4_f32", blocks);
        run<2, 16>("v_mfma_f32_16x16x4_f32", blocks);
        run<4, 16>("v_mfma_f32_16x16x4_f32", blocks);
    }
    return 0;
}
