// Microbenchmark: would Winograd F(2x2,3x3) pay for the 6x64 tower?  
//
// One 3x3 64->64 layer on an 11x11 board is, as F(2x2,3x3), 16 independent GEMMs [36 tiles x 64 cin] x [64 x 64
// cout] (one per position xi of the 4x4 transformed tile) between an input transform (B^T d B, fp32 VALU + a
// hi/lo f16 split of 16 values per 4 outputs) and an output transform (A^T m A).  With the f16x3 operand split
// that is 864 MFMAs (16x16x32) per board and layer against 1 728 for the direct convolution -- but each GEMM has
// K = 64 only, and the accumulators cost five times the registers (the four outputs of a tile are folded from
// the sixteen products: Y[4] + M per tile-channel pair instead of one accumulator), so the register tile
// shrinks from 4 x 4 to at most 8 tile pairs and the operand bytes per MFMA double.
//
// This program measures the GEMM PHASE ALONE under the fused tower's conditions (two 256-thread blocks per CU,
// weight fragments from L2 in fragment order, activation fragments from LDS, three products per accumulator,
// single register set reloaded after last use) -- no input transform, no output transform, no barriers, no
// LDS writes: an UPPER bound on what a Winograd layer could reach -- next to the direct convolution's k-loop in
// the same harness:
//   D        direct: 18 k-steps per layer, 4 x 4 tiles per wave, 8 weight + 8 activation fragments per 48 MFMAs
//   W<Tc,Tt> Winograd: 16 xi x 2 k-steps per layer, Tc channel tiles x Tt tile-tiles per wave, 2 Tc weight +
//            2 Tt transformed-activation fragments per 3 Tc Tt MFMAs, M folded into Y after each xi (VALU)
// Output: time per board and layer (CU-microseconds) of each pattern and the MFMA issue rate.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/wino_gemm.hip -o wino_gemm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f16x8 ld_g(const uint4 *p) { const uint4 q = *p; return *reinterpret_cast<const f16x8 *>(&q); }

// ---- direct pattern: the tower's k-loop without its hand-placed prefetch (4 x 4 tiles, 18 k-steps) --------------
__global__ __launch_bounds__(256, 2) void k_direct(const uint4 *wsrc, const uint4 *xsrc, float *out, int layers) {
    extern __shared__ uint4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += 256) lds[i] = xsrc[i];
    __syncthreads();
    f32x4 acc[4][4];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
    f16x8 wh[4], wl[4], xh[4], xl[4];
    auto load_w = [&](int st, int n) { wh[n] = ld_g(wsrc + ((size_t)st * 8 + n * 2) * 64 + lane); wl[n] = ld_g(wsrc + ((size_t)st * 8 + n * 2 + 1) * 64 + lane); };
    auto load_x = [&](int st, int m) {
        const uint4 a = lds[((st * 8 + m * 2 + wave * 3) * 64 + lane) & 4095], b = lds[((st * 8 + m * 2 + 1 + wave * 3) * 64 + lane) & 4095];
        xh[m] = *reinterpret_cast<const f16x8 *>(&a); xl[m] = *reinterpret_cast<const f16x8 *>(&b);
    };
    for (int n = 0; n < 4; ++n) load_w(0, n);
    for (int m = 0; m < 4; ++m) load_x(0, m);
    const int steps = layers * 18;
    for (int st = 0; st < steps; ++st) {
        const int ws = (st + 1) % (12 * 18);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[n], xh[m], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[n], xh[m], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[n], xl[m], acc[m][n], 0, 0, 0);
                if (n == 3) load_x(st + 1, m);               // tile m's activations are done for this step
                __builtin_amdgcn_sched_barrier(0);
            }
            load_w(ws, n);                                   // channel tile n's weights are done for this step
        }
    }
    float s = 0;
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    out[blockIdx.x * 256 + tid] = s;
}

// ---- Winograd GEMM phase: 16 xi x 2 k-steps, Tc x Tt register tile, M folded into Y[4] per xi -------------------
template <int TC, int TT>
__global__ __launch_bounds__(256, 2) void k_wino(const uint4 *usrc, const uint4 *vsrc, float *out, int layers) {
    extern __shared__ uint4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += 256) lds[i] = vsrc[i];
    __syncthreads();
    f32x4 Y[4][TC][TT], M[TC][TT];
    for (int o = 0; o < 4; ++o) for (int c = 0; c < TC; ++c) for (int t = 0; t < TT; ++t) Y[o][c][t] = f32x4{0, 0, 0, 0};
    f16x8 uh[TC], ul[TC], vh[TT], vl[TT];
    // U of one layer: [xi 16][ks 2][cout tile 4][hi,lo][lane]; a wave owns TC of the 4 channel tiles
    const int c0 = (wave * TC) & 3;
    const uint4 *ubase = usrc + lane;
    uint32_t uoff = 0, voff = (uint32_t)(wave * 5 * 64 + lane);      // running offsets (uint4 units), wrapped below
    auto load_u = [&](int, int c) {
        const uint32_t o = uoff + (uint32_t)(((c0 + c) & 3) * 128);
        uh[c] = ld_g(ubase + o); ul[c] = ld_g(ubase + o + 64);
    };
    auto load_v = [&](int, int t) {
        const uint4 a = lds[(voff + t * 128) & 4095], b = lds[(voff + t * 128 + 64) & 4095];
        vh[t] = *reinterpret_cast<const f16x8 *>(&a); vl[t] = *reinterpret_cast<const f16x8 *>(&b);
    };
    for (int c = 0; c < TC; ++c) load_u(0, c);
    for (int t = 0; t < TT; ++t) load_v(0, t);
    int q = 0;
    for (int layer = 0; layer < layers; ++layer) {
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            for (int c = 0; c < TC; ++c) for (int t = 0; t < TT; ++t) M[c][t] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks, ++q) {
                uoff += 512;                                  // next (xi, ks) step: 4 channel tiles x (hi, lo) x 64 lanes
                if (uoff >= 12u * 32u * 512u) uoff = 0;
                voff += TT * 128;
                asm volatile("" : "+v"(voff), "+s"(uoff));
#pragma unroll
                for (int c = 0; c < TC; ++c) {
#pragma unroll
                    for (int t = 0; t < TT; ++t) {
                        M[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(uh[c], vh[t], M[c][t], 0, 0, 0);
                        M[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ul[c], vh[t], M[c][t], 0, 0, 0);
                        M[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(uh[c], vl[t], M[c][t], 0, 0, 0);
                        if (c == TC - 1) load_v(q + 1, t);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    load_u(q + 1, c);
                }
            }
            // A^T m A: xi = (a, b) contributes +-m to Y[i][j] for the non-zero A[a][i] A[b][j]
            const int a = xi >> 2, b = xi & 3;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int ca = i == 0 ? (a < 3 ? 1 : 0) : (a == 0 ? 0 : (a == 1 ? 1 : -1));
                    const int cb = j == 0 ? (b < 3 ? 1 : 0) : (b == 0 ? 0 : (b == 1 ? 1 : -1));
                    const int sgn = ca * cb;
                    if (sgn != 0)
                        for (int c = 0; c < TC; ++c) for (int t = 0; t < TT; ++t)
                            Y[i * 2 + j][c][t] = sgn > 0 ? Y[i * 2 + j][c][t] + M[c][t] : Y[i * 2 + j][c][t] - M[c][t];
                }
        }
    }
    float s = 0;
    for (int o = 0; o < 4; ++o) for (int c = 0; c < TC; ++c) for (int t = 0; t < TT; ++t) s += Y[o][c][t][0] + Y[o][c][t][1] + Y[o][c][t][2] + Y[o][c][t][3];
    out[blockIdx.x * 256 + tid] = s;
}

// ---- Winograd input transform alone: X (fp32, LDS) -> V = B^T d B -> hi/lo f16 -> LDS ---------------------------
// 2 boards per block (72 tiles).  A thread takes (tile, 4 channels): 16 window reads of 16 B, the 4x4 transform
// (32 packed adds), 16 hi/lo splits of 4 values, 32 writes of 8 B.  No MFMA, no GEMM: the VALU + LDS cost the
// direct convolution does not have (it splits 121 x 64 values per layer, this splits 36 x 16 x 64).
__global__ __launch_bounds__(256, 2) void k_wino_in(const float4 *xsrc, float *out, int layers) {
    extern __shared__ uint4 lds[];
    float4 *X = reinterpret_cast<float4 *>(lds);                      // [2][13*13 (zero halo)][16 float4]  = 86.5 KB... use 12x... see below
    const int tid = threadIdx.x;
    constexpr int NPAD = 14;                                          // rows/cols -1..12 of an 11x11 board
    for (int i = tid; i < 2 * NPAD * NPAD * 16; i += 256) X[i] = xsrc[i & 4095];
    uint2 *V = reinterpret_cast<uint2 *>(X + 2 * NPAD * NPAD * 16);   // 16 KB window of the V image (wrapped)
    __syncthreads();
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    float acc = 0.f;
    for (int layer = 0; layer < layers; ++layer) {
        for (int task = tid; task < 72 * 16; task += 256) {
            const int tile = task >> 4, cg = task & 15;
            const int b = tile / 36, t = tile - 36 * b, ty = t / 6, tx = t - 6 * ty;
            const float4 *w0 = X + ((size_t)b * NPAD * NPAD + (2 * ty) * NPAD + 2 * tx) * 16 + cg;
            float4 d[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) d[r][c] = w0[(r * NPAD + c) * 16];
            float4 tt[4][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {     // B^T d: rows
                tt[0][c] = make_float4(d[0][c].x - d[2][c].x, d[0][c].y - d[2][c].y, d[0][c].z - d[2][c].z, d[0][c].w - d[2][c].w);
                tt[1][c] = make_float4(d[1][c].x + d[2][c].x, d[1][c].y + d[2][c].y, d[1][c].z + d[2][c].z, d[1][c].w + d[2][c].w);
                tt[2][c] = make_float4(d[2][c].x - d[1][c].x, d[2][c].y - d[1][c].y, d[2][c].z - d[1][c].z, d[2][c].w - d[1][c].w);
                tt[3][c] = make_float4(d[1][c].x - d[3][c].x, d[1][c].y - d[3][c].y, d[1][c].z - d[3][c].z, d[1][c].w - d[3][c].w);
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {     // (B^T d) B: columns, then split and store
                float4 v[4];
                v[0] = make_float4(tt[a][0].x - tt[a][2].x, tt[a][0].y - tt[a][2].y, tt[a][0].z - tt[a][2].z, tt[a][0].w - tt[a][2].w);
                v[1] = make_float4(tt[a][1].x + tt[a][2].x, tt[a][1].y + tt[a][2].y, tt[a][1].z + tt[a][2].z, tt[a][1].w + tt[a][2].w);
                v[2] = make_float4(tt[a][2].x - tt[a][1].x, tt[a][2].y - tt[a][1].y, tt[a][2].z - tt[a][1].z, tt[a][2].w - tt[a][1].w);
                v[3] = make_float4(tt[a][1].x - tt[a][3].x, tt[a][1].y - tt[a][3].y, tt[a][1].z - tt[a][3].z, tt[a][1].w - tt[a][3].w);
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    h4 hi = {(_Float16)v[bb].x, (_Float16)v[bb].y, (_Float16)v[bb].z, (_Float16)v[bb].w};
                    h4 lo = {(_Float16)(v[bb].x - (float)hi[0]), (_Float16)(v[bb].y - (float)hi[1]),
                             (_Float16)(v[bb].z - (float)hi[2]), (_Float16)(v[bb].w - (float)hi[3])};
                    const int slot = (((a * 4 + bb) * 72 + tile) * 32 + cg * 2 + layer) & 2047;
                    V[slot] = *reinterpret_cast<uint2 *>(&hi);
                    V[(slot + 1024) & 2047] = *reinterpret_cast<uint2 *>(&lo);
                }
            }
        }
        __syncthreads();
        acc += reinterpret_cast<float *>(V)[tid];
    }
    out[blockIdx.x * 256 + tid] = acc;
}

template <typename F>
static float run(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch();                       // warm-up
    hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    const size_t wn = (size_t)12 * 32 * 8 * 64;      // >= both patterns' per-12-layer fragment sets (uint4 each)
    std::vector<_Float16> h(wn * 8), x(4096 * 8);
    srand(1);
    for (auto &v : h) v = (_Float16)((rand() % 2001 - 1000) / 8000.0f);
    for (auto &v : x) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    uint4 *dw, *dx; float *o;
    hipMalloc(&dw, wn * 16); hipMalloc(&dx, 4096 * 16); hipMalloc(&o, 512 * 256 * 4);
    hipMemcpy(dw, h.data(), wn * 16, hipMemcpyHostToDevice);
    hipMemcpy(dx, x.data(), 4096 * 16, hipMemcpyHostToDevice);
    const int blocks = 512, layers = 240;           // 2 blocks per CU, 20 towers' worth of layers
    const size_t ldsb = 65536;
    hipFuncSetAttribute((const void *)k_direct, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipFuncSetAttribute((const void *)k_wino<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipFuncSetAttribute((const void *)k_wino<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipFuncSetAttribute((const void *)k_wino<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipFuncSetAttribute((const void *)k_wino<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipFuncSetAttribute((const void *)k_wino<2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipFuncSetAttribute((const void *)k_wino<3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    struct Row { const char *name; float ms; double mfma_per_wave_layer; double boards_per_wave_layer; };
    std::vector<Row> rows;
    for (int rep = 0; rep < 2; ++rep) {
        rows.push_back({"D  direct 4x4 tiles, 18 k-steps", run([&] { hipLaunchKernelGGL(k_direct, dim3(blocks), dim3(256), ldsb, 0, dw, dx, o, layers); }), 18.0 * 48, 0.5});
        rows.push_back({"W<4,2> winograd GEMM (spills 71 VGPRs)", run([&] { hipLaunchKernelGGL((k_wino<4, 2>), dim3(blocks), dim3(256), ldsb, 0, dw, dx, o, layers); }), 32.0 * 24, 32.0 / 36.0});
        rows.push_back({"W<2,4> winograd GEMM (spills 71 VGPRs)", run([&] { hipLaunchKernelGGL((k_wino<2, 4>), dim3(blocks), dim3(256), ldsb, 0, dw, dx, o, layers); }), 32.0 * 24, 64.0 / 36.0 * 0.5});
        rows.push_back({"W<2,3> winograd GEMM phase", run([&] { hipLaunchKernelGGL((k_wino<2, 3>), dim3(blocks), dim3(256), ldsb, 0, dw, dx, o, layers); }), 32.0 * 18, 48.0 / 36.0 * 0.5});
        rows.push_back({"W<3,2> winograd GEMM phase", run([&] { hipLaunchKernelGGL((k_wino<3, 2>), dim3(blocks), dim3(256), ldsb, 0, dw, dx, o, layers); }), 32.0 * 18, 32.0 / 36.0 * 0.75});
        rows.push_back({"W<2,2> winograd GEMM phase", run([&] { hipLaunchKernelGGL((k_wino<2, 2>), dim3(blocks), dim3(256), ldsb, 0, dw, dx, o, layers); }), 32.0 * 12, 32.0 / 36.0 * 0.5});
        rows.push_back({"W<4,1> winograd GEMM phase", run([&] { hipLaunchKernelGGL((k_wino<4, 1>), dim3(blocks), dim3(256), ldsb, 0, dw, dx, o, layers); }), 32.0 * 12, 16.0 / 36.0});
    }
    {   // the input transform alone: 2 boards per block and layer
        const size_t ldsin = (size_t)2 * 14 * 14 * 16 * 16 + 16384;
        hipFuncSetAttribute((const void *)k_wino_in, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsin);
        for (int rep = 0; rep < 2; ++rep) {
            const float ms = run([&] { hipLaunchKernelGGL(k_wino_in, dim3(blocks), dim3(256), ldsin, 0, (const float4 *)dx, o, layers); });
            printf("winograd input transform alone (fp32 window reads, B^T d B, hi/lo split, LDS writes): %.2f ms = %.3f CU-us per board-layer\n",
                   ms, ms * 1e3 * 256.0 / (blocks * 2.0 * layers));
        }
    }
    printf("%-34s %9s %14s %22s %16s\n", "pattern", "ms", "issued PFLOP/s", "CU-us per board-layer", "vs direct");
    double direct = 0;
    for (const Row &r : rows) {
        const double waves = blocks * 4.0;
        const double flops = waves * layers * r.mfma_per_wave_layer * 16384.0;
        const double board_layers = waves * layers * r.boards_per_wave_layer;
        const double cu_us = r.ms * 1e3 * 256.0 / board_layers;
        if (r.name[0] == 'D') direct = cu_us;
        printf("%-34s %9.2f %14.3f %22.3f %16.2f\n", r.name, r.ms, flops / r.ms / 1e12, cu_us, direct / cu_us);
    }
    printf("(board-layer = one 64->64 3x3 layer of one 11x11 board: 1728 MFMAs direct, 864 as 36 Winograd tiles; the W rows are the GEMM\n"
           " phase only -- no input / output transform -- so 'vs direct' is an upper bound on a Winograd layer's gain)\n");
    return 0;
}
