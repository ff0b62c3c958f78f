// Microbenchmark: what a consumer kernel pays to read what the previous kernel wrote (the training step is a chain
// of such pairs): 256 blocks x 256 threads, every thread requests 16 x 16 bytes (62 KB per block, contiguous per block)
// and stamps the shader clock when they have all arrived.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/kernel_chain.hip -o tools/microbench/kernel_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k_write(float4 *dst, float v) {
    const size_t base = (size_t)blockIdx.x * 256 * 8;
    for (int k = 0; k < 8; ++k) dst[base + threadIdx.x + 256 * k] = make_float4(v, v + 1, v + 2, v + 3);
}

__global__ __launch_bounds__(256) void k_read(const float4 *a, const float4 *b, float *sink, unsigned long long *cyc, int nload) {
    const size_t base = (size_t)blockIdx.x * 256 * 8;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float4 va[8], vb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        va[k] = k < nload ? a[base + threadIdx.x + 256 * k] : make_float4(0, 0, 0, 0);
        vb[k] = k < nload ? b[base + threadIdx.x + 256 * k] : make_float4(0, 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += va[k].x + vb[k].y;
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) atomicAdd(cyc, t1 - t0);
}

int main() {
    const int blocks = 256;
    const size_t n4 = (size_t)blocks * 256 * 8;      // 8 MB per buffer of float4... 512 K float4 = 8 MB
    float4 *a, *b, *old_a, *old_b;
    float *sink;
    unsigned long long *cyc;
    hipMalloc(&a, n4 * 16); hipMalloc(&b, n4 * 16); hipMalloc(&old_a, n4 * 16); hipMalloc(&old_b, n4 * 16);
    hipMalloc(&sink, 4); hipMalloc(&cyc, 8);
    hipMemset(old_a, 0, n4 * 16); hipMemset(old_b, 0, n4 * 16);
    hipDeviceSynchronize();
    auto run = [&](const char *name, bool fresh, int nload) {
        hipMemset(cyc, 0, 8);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        const int reps = 200;
        for (int w = 0; w < 2; ++w) {
            if (w == 1) { hipDeviceSynchronize(); hipMemset(cyc, 0, 8); hipEventRecord(e0); }
            for (int i = 0; i < reps; ++i) {
                if (fresh) {
                    hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, a, (float)i);
                    hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, b, (float)i);
                }
                hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, fresh ? a : old_a, fresh ? b : old_b, sink, cyc, nload);
            }
        }
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-44s loads/thread %2d: %8.0f cycles until the data is there (mean over waves), %.1f us per launch group\n", name, 2 * nload,
               (double)h / (reps * blocks * 4.0), 1e3 * ms / reps);
    };
    run("read what the two previous kernels wrote", true, 8);
    run("read buffers nobody writes (same pattern)", false, 8);
    run("read what the two previous kernels wrote", true, 2);
    run("read buffers nobody writes (same pattern)", false, 2);
    run("read what the two previous kernels wrote", true, 1);
    return 0;
}
