// Microbenchmark: does the ORDER in which the split-f16 tower issues its 48 MFMAs per k-step (4 position tiles x 4
// channel tiles x {hi*hi, lo*hi, hi*lo}) change the power-limited rate?  Same operands, same products, same
// accumulators; data distributed like the tower's (ReLU activations: half zero; weights ~N(0, 0.05); lo = residue
// of the f16 rounding).  Two waves per SIMD, fragments re-read from LDS every step.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_order.hip -o mfma_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// slot j of a step: 0-3 wh[n], 4-7 wl[n], 8-11 xh[m], 12-15 xl[m]
template <int ORDER>
__global__ __launch_bounds__(256, 2) void k(const uint4 *src, float *out, int iters) {
    extern __shared__ uint4 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 256) lds[i] = src[i];
    __syncthreads();
    f32x4 acc[4][4];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        f16x8 f[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) { const uint4 q = lds[((it * 16 + j) * 64 + lane) & 4095]; f[j] = *reinterpret_cast<const f16x8 *>(&q); }
#define MF(m, n, p) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[(n) + 4 * ((p) == 1)], f[8 + (m) + 4 * ((p) == 2)], acc[m][n], 0, 0, 0); __builtin_amdgcn_sched_barrier(0)
        if (ORDER == 0) {          // the tower's: per (m, n) the three products back to back on one accumulator
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 2 * h; n < 2 * h + 2; ++n)
#pragma unroll
                        for (int p = 0; p < 3; ++p) { MF(m, n, p); }
        } else if (ORDER == 1) {   // product outermost
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) { MF(m, n, p); }
        } else if (ORDER == 2) {   // B (activation) operand held: xh[m] x (wh 0-3, wl 0-3), then xl[m] x wh 0-3
#pragma unroll
            for (int m = 0; m < 4; ++m) {
#pragma unroll
                for (int n = 0; n < 4; ++n) { MF(m, n, 0); }
#pragma unroll
                for (int n = 0; n < 4; ++n) { MF(m, n, 1); }
#pragma unroll
                for (int n = 0; n < 4; ++n) { MF(m, n, 2); }
            }
        } else if (ORDER == 3) {   // A (weight) operand held: wh[n] x (xh 0-3, xl 0-3), then wl[n] x xh 0-3
#pragma unroll
            for (int n = 0; n < 4; ++n) {
#pragma unroll
                for (int m = 0; m < 4; ++m) { MF(m, n, 0); }
#pragma unroll
                for (int m = 0; m < 4; ++m) { MF(m, n, 2); }
#pragma unroll
                for (int m = 0; m < 4; ++m) { MF(m, n, 1); }
            }
        } else if (ORDER == 4) {   // both change every MFMA (worst case for operand latches)
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int m = 0; m < 4; ++m) { MF(m, (m + d) & 3, p); }
        } else if (ORDER == 5) {   // the tower's order, the small (lo) products first
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 2 * h; n < 2 * h + 2; ++n) { MF(m, n, 1); MF(m, n, 2); MF(m, n, 0); }
        } else if (ORDER == 6) {   // same accumulator, ONE operand changes between consecutive products
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 2 * h; n < 2 * h + 2; ++n) { MF(m, n, 1); MF(m, n, 0); MF(m, n, 2); }
        } else if (ORDER == 7) {   // order 6 as a snake: the pair's second triple runs backwards (xl held across the seam)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    MF(m, 2 * h, 1); MF(m, 2 * h, 0); MF(m, 2 * h, 2);
                    MF(m, 2 * h + 1, 2); MF(m, 2 * h + 1, 0); MF(m, 2 * h + 1, 1);
                }
        } else if (ORDER == 8) {   // all four channel tiles of a position tile, same-accumulator triples (no halves)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) { MF(m, n, 0); MF(m, n, 1); MF(m, n, 2); }
        }
    }
    float sum = 0.f;
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 4; ++r) sum += acc[m][n][r];
    out[blockIdx.x * 256 + tid] = sum;
}

// RUN = 3 * NK products back to back on one accumulator: NK k-steps' fragments held at once (NK x 64 VGPRs)
template <int NK>
__global__ __launch_bounds__(256, 2) void k2(const uint4 *src, float *out, int iters) {
    extern __shared__ uint4 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 256) lds[i] = src[i];
    __syncthreads();
    f32x4 acc[4][4];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; it += NK) {
        f16x8 f[NK][16];
#pragma unroll
        for (int kk = 0; kk < NK; ++kk)
#pragma unroll
            for (int j = 0; j < 16; ++j) { const uint4 q = lds[(((it + kk) * 16 + j) * 64 + lane) & 4095]; f[kk][j] = *reinterpret_cast<const f16x8 *>(&q); }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int kk = 0; kk < NK; ++kk)
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[kk][n + 4 * (p == 1)], f[kk][8 + m + 4 * (p == 2)], acc[m][n], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
    }
    float sum = 0.f;
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) for (int r = 0; r < 4; ++r) sum += acc[m][n][r];
    out[blockIdx.x * 256 + tid] = sum;
}

template <int NK>
static void run2(const uint4 *d, float *o, int iters) {
    hipFuncSetAttribute((const void *)k2<NK>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, ms = 0.f, tot = 0.f;
    for (int rep = 0; rep < 42; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k2<NK>, dim3(512), dim3(256), 72 * 1024, 0, d, o, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) { tot += ms; if (ms < best) best = ms; }
    }
    const double fl = 512.0 * 4 * iters * 48 * 16384.0;
    printf("run of %d on one accumulator: mean %.3f ms  best %.3f ms  %.1f TFLOP/s issued (mean)\n", 3 * NK, tot / 40, best, fl / (tot / 40 * 1e-3) / 1e12);
}

static float gauss() { float u = (rand() + 1.0f) / (RAND_MAX + 2.0f), v = (rand() + 1.0f) / (RAND_MAX + 2.0f); return sqrtf(-2.f * logf(u)) * cosf(6.2831853f * v); }

template <int ORDER>
static void run(const uint4 *d, float *o, int iters) {
    hipFuncSetAttribute((const void *)k<ORDER>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, ms = 0.f, tot = 0.f;
    for (int rep = 0; rep < 42; ++rep) {     // 6 x ~8 ms back to back: the clock settles under the load
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<ORDER>, dim3(512), dim3(256), 72 * 1024, 0, d, o, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) { tot += ms; if (ms < best) best = ms; }
    }
    const double fl = 512.0 * 4 * iters * 48 * 16384.0;
    printf("order %d: mean %.3f ms  best %.3f ms  %.1f TFLOP/s issued (mean)\n", ORDER, tot / 40, best, fl / (tot / 40 * 1e-3) / 1e12);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 6000;
    std::vector<_Float16> h(4096 * 8);
    srand(1);
    for (int i = 0; i < 4096; ++i) {
        const int slot = (i / 64) % 16;
        for (int e = 0; e < 8; ++e) {
            const bool weight = slot < 8, lo = (slot & 4) != 0;
            float v = weight ? 0.05f * gauss() : fmaxf(0.f, gauss());
            const _Float16 hi = (_Float16)v;
            h[i * 8 + e] = lo ? (_Float16)(v - (float)hi) : hi;
        }
    }
    uint4 *d; float *o;
    hipMalloc(&d, 4096 * 16); hipMalloc(&o, 512 * 256 * 4);
    hipMemcpy(d, h.data(), 4096 * 16, hipMemcpyHostToDevice);
    for (int round = 0; round < 3; ++round) {
        run<0>(d, o, iters); run<1>(d, o, iters); run<2>(d, o, iters); run<3>(d, o, iters); run<4>(d, o, iters); run<5>(d, o, iters); run<6>(d, o, iters); run<7>(d, o, iters); run<8>(d, o, iters); run2<1>(d, o, iters); run2<2>(d, o, iters); run2<3>(d, o, iters);
    }
    return 0;
}
