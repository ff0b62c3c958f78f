// Microbenchmark: sustained f16 MFMA rate and shader clock for the 32x32x16 and 16x16x32 shapes
// under the fused tower's conditions (two waves per SIMD, operands re-read from LDS every step,
// three products per accumulator, random data).  Build: hipcc --offload-arch=gfx950 -O3 mfma_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void k(const uint4 *src, float *out, int iters, unsigned long long *clk) {
    extern __shared__ uint4 lds[];   // 64 KB used; the launch may ask for more to force one block per CU
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 256) lds[i] = src[i];
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[4];
        for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
            f16x8 f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const uint4 q = lds[((it * 8 + j) * 64 + lane) & 4095]; f[j] = *reinterpret_cast<const f16x8 *>(&q); }
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[(a & 1) + 2 * (p == 1)], f[4 + (a >> 1) + 2 * (p == 2)], acc[a], 0, 0, 0);
        }
        for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) sum += acc[a][r];
    } else {
        f32x4 acc[16];
        for (int a = 0; a < 16; ++a) for (int r = 0; r < 4; ++r) acc[a][r] = 0.f;
        for (int it = 0; it < iters; ++it) {     // one iteration = 32 channels = 2 of the 32x32x16 iterations' k
            f16x8 f[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) { const uint4 q = lds[((it * 16 + j) * 64 + lane) & 4095]; f[j] = *reinterpret_cast<const f16x8 *>(&q); }
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int a = 0; a < 16; ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[(a & 3) + 4 * (p == 1)], f[8 + (a >> 2) + 4 * (p == 2)], acc[a], 0, 0, 0);
        }
        for (int a = 0; a < 16; ++a) for (int r = 0; r < 4; ++r) sum += acc[a][r];
    }
    out[blockIdx.x * 256 + tid] = sum;
    if (tid == 0 && blockIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

int main() {
    std::vector<_Float16> h(4096 * 8);
    srand(1);
    for (auto &x : h) x = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    uint4 *d; float *o; unsigned long long *c;
    hipMalloc(&d, 4096 * 16); hipMalloc(&o, 512 * 256 * 4); hipMalloc(&c, 16);
    hipMemcpy(d, h.data(), 4096 * 16, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void *)k<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int occ : {2, 1})
    for (int shape : {32, 16, 32, 16}) {
        const size_t ldsb = occ == 2 ? 65536 : 100 * 1024;
        const int iters = shape == 32 ? 40000 : 20000;      // same flops per wave
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(512 / (3 - occ)), dim3(256), ldsb, 0, d, o, iters, c);
        else hipLaunchKernelGGL(k<16>, dim3(512 / (3 - occ)), dim3(256), ldsb, 0, d, o, iters, c);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long hc[2]; hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
        const double flops = (512.0 / (3 - occ)) * 4 * (shape == 32 ? 40000.0 * 12 * 32768 : 20000.0 * 48 * 16384);
        printf("%d block(s)/CU, shape %dx%d: %.2f ms, %.1f TFLOP/s issued f16, shader clock %.0f MHz\n", occ, shape, shape, ms,
               flops / ms / 1e9, 100.0 * hc[0] / hc[1]);
    }
    return 0;
}
