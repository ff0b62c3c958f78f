// train_kernels.hip -- one optimizer step of the reference's trainer on gfx950 (MI355X), hand-written.
//
// Restates azalea/policy_trainer.py:123-142 (supervised_step: zero_grad, forward, loss, backward, SGD step) over
// azalea/network.py:68-102 (Network.forward in TRAIN mode -- BatchNorm normalises with the batch statistics and
// updates its running ones -- and the loss of Network.run: mse(value, reward) - sum(moves_prob * logprob) / B) and
// :134-152 (HexNetwork: embedding, policy FC, legal-move gather, log_softmax), with torch.optim.SGD's update
// (momentum, weight decay, no dampening / nesterov: d = g + wd p; buf = mu buf + d; p -= lr buf).
//
// Design.  At the reference's batch of 128 boards a step is 41 GFLOP of 3x3 convolutions in a
// chain of ~40 dependent layer passes: latency-bound, not throughput-bound.  So: fp32 MFMA (v_mfma_f32_32x32x2_f32,
// exact products -- gradients span ten orders of magnitude, no f16 range games), one workgroup per (board, 32 output
// channels) so that a layer pass fills all 256 CUs with 4 waves each, every elementwise stage fused into the
// producer's epilogue or the consumer's prologue, and the whole step captured once as a HIP graph:
//   forward  : k_trn_stem_fwd, L x k_trn_conv<FWD> (prologue: BN(batch stats) + residual + ReLU of the INPUT, written
//              out once for the backward pass; epilogue: raw output + per-channel sum / sum of squares), heads
//   backward : heads, L x k_trn_conv<BWD> (prologue: BatchNorm backward of the incoming gradient; implicit GEMM with
//              the flipped / transposed filters; epilogue: skip-connection add, ReLU mask, the next BatchNorm's two
//              reductions), L x k_trn_wgrad (split over boards, deterministic two-stage reduction), stem
//   update   : k_trn_finalize (BN gradients, running statistics, small reductions), k_trn_update (SGD on every
//              tensor IN PLACE in the trainer's torch tensors + the MFMA-order copies of the filters)
// Layouts: activations [B][cells][C] fp32; per BatchNorm layer four f64 sums per channel {x, x^2, g, g xhat}.
#include "train.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/azx.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define TRN_EPS 1e-5
#define TRN_WG_GROUPS 28          // k_trn_wgrad: board groups per tap (9 x 28 = 252 workgroups <= 256 CUs)

static thread_local std::string g_trn_err;
const char *azx_trn_error() { return g_trn_err.c_str(); }
static int tfail(int code, const std::string &msg) {
    g_trn_err = msg;
    return code;
}

// ---- device-side description of one step ---------------------------------------------------------------------
struct TrnDev {
    int N, cells, C, L, B;
    float invN;                    // 1 / (B * cells): BatchNorm's population
    // inputs
    const int32_t *board, *legal;  // [B][cells]
    const float *prob, *reward;    // [B][cells], [B]
    // parameters (torch tensors, updated in place)
    const float *emb, *w1;         // encoder.weight [3][4], conv1.weight [C][4][3][3]
    const float *const *bn_w, *const *bn_b;      // [L + 1] -> [C]   (device arrays of pointers)
    const float *vconv, *pconv;    // [2][C], [4][C]
    const float *hbn_w[2], *hbn_b[2];            // value_bn1 (2), move_bn1 (4)
    const float *fc2w, *fc2b, *fc3w, *fc3b, *mfw, *mfb;
    // work buffers
    float *const *raw, *const *act, *const *g;   // [L + 1] -> [B][cells][C]
    const float *const *Wf, *const *Wb;          // [L + 1] (index 1..L) MFMA-order filters: forward / backward-data
    const float *const *convw;                   // [L + 1] (index 1..L) the filters themselves (torch layout [co][ci][3][3])
    double *sums;                  // [(L + 1)][C][4]
    double *hsums;                 // [6][4]
    double *lossacc;               // [2]
    float *hraw, *hact, *g6;       // [B][6][cells]
    float *h2, *dh2;               // [B][64]
    float *dv3;                    // [B]
    float *dlogit;                 // [B][128] dense by tile
    float *value, *logprob;        // [B], [B][cells]
    float *loss3;
    float *wpart;                  // [L][G][C*C*9] weight-gradient partial sums (index l - 1)
    float *stem_part;              // [B][27*C]
    float *hconv_part;             // [B][6*C]
    float *grad;                   // flat gradient buffer (offsets in the segment table)
    const float *hp;               // lr, momentum, weight decay
};

__device__ __forceinline__ double dsum(const double *s, int l, int C, int c, int k) { return s[((size_t)l * C + c) * 4 + k]; }

// per-channel BatchNorm coefficients of layer l from its batch sums
__device__ __forceinline__ void bn_coeffs(const TrnDev &P, int l, int c, float &mean, float &inv) {
    const double m = dsum(P.sums, l, P.C, c, 0) * (double)P.invN;
    const double v = dsum(P.sums, l, P.C, c, 1) * (double)P.invN - m * m;
    mean = (float)m;
    inv = (float)(1.0 / sqrt((v > 0 ? v : 0) + TRN_EPS));
}

// =================================================================================================================
// stem forward: embedding (3 -> 4) o conv 3x3 (4 -> C) as a [tap][cell value][cout] table built per block
// =================================================================================================================
template <int C>
__global__ __launch_bounds__(256) void k_trn_stem_fwd(TrnDev P) {
    __shared__ float T[28 * C];
    __shared__ unsigned char cellv[128];
    __shared__ float red[2][256];
    const int b = blockIdx.x, tid = threadIdx.x, N = P.N, cells = P.cells;
    for (int i = tid; i < 27 * C; i += 256) {
        const int k = i / C, co = i - k * C, tap = k / 3, v = k - tap * 3;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s += P.emb[v * 4 + j] * P.w1[(co * 4 + j) * 9 + tap];
        T[i] = s;
    }
    for (int i = tid; i < cells; i += 256) cellv[i] = (unsigned char)P.board[(size_t)b * cells + i];
    __syncthreads();
    const int co = tid % C;
    float s1 = 0.f, s2 = 0.f;
    float *out = P.raw[0] + (size_t)b * cells * C;
    for (int pos = tid / C; pos < cells; pos += 256 / C) {
        const int y = pos / N, x = pos - y * N;
        float acc = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            if (yy >= 0 && yy < N && xx >= 0 && xx < N) acc += T[(tap * 3 + cellv[yy * N + xx]) * C + co];
        }
        out[(size_t)pos * C + co] = acc;
        s1 += acc;
        s2 += acc * acc;
    }
    red[0][tid] = s1;
    red[1][tid] = s2;
    __syncthreads();
    if (tid < C) {
        double a = 0, q = 0;
        for (int i = tid; i < 256; i += C) { a += red[0][i]; q += red[1][i]; }
        atomicAdd(&P.sums[((size_t)0 * C + tid) * 4 + 0], a);
        atomicAdd(&P.sums[((size_t)0 * C + tid) * 4 + 1], q);
    }
}

// =================================================================================================================
// 3x3 convolution as an implicit GEMM on v_mfma_f32_32x32x2_f32, forward and backward-data
//   grid (N tiles of 32 output channels, boards), 4 waves: wave w owns positions 32 w .. 32 w + 31
//   LDS: the board's INPUT operand [cells + 1 zero row][C + 4]
// FWD  (layer l: raw_l = conv(act_{l-1})):
//   prologue  act_{l-1} = relu(BN_{l-1}(raw_{l-1}) [+ act_{l-3}])  (batch statistics), staged and written to HBM
//   epilogue  raw_l and its per-channel sum / sum of squares
// BWD  (layer l: dL/dact_{l-1} = conv^T(draw_l)):
//   prologue  draw_l = gamma inv (g_l - mean(g_l) - xhat_l mean(g_l xhat_l))       (BatchNorm backward)
//   epilogue  g_{l-1} = (acc [+ g_{l+1}: the skip connection]) * (act_{l-1} > 0), and the two reductions BN_{l-1}'s
//             backward needs: sum g_{l-1}, sum g_{l-1} xhat_{l-1}
// =================================================================================================================
enum { ROLE_FWD = 0, ROLE_BWD = 1 };

template <int C, int ROLE>
__global__ __launch_bounds__(256) void k_trn_conv(TrnDev P, int l) {
    constexpr int NT = (C + 31) / 32, LDW = C + 4, Q = C / 8, C4 = C / 4;
    extern __shared__ __align__(16) float lds[];
    float *X = lds;                                    // [(cells + 1)][LDW]
    const int N = P.N, cells = P.cells;
    float *cA = X + (size_t)(cells + 1) * LDW;          // per input channel coefficients
    float *cB = cA + C, *cM = cB + C, *cI = cM + C, *cK = cI + C;     // cK: [2][C] (BWD)
    float *pM = cK + 2 * C, *pI = pM + C;               // BWD epilogue: mean / invstd of layer l - 1
    float *red = pI + C;                                // [4][32][2]
    const int nt = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;

    // ---- per-channel coefficients -----------------------------------------------------------------------
    if (tid < C) {
        const int c = tid;
        if (ROLE == ROLE_FWD) {
            float mean, inv;
            bn_coeffs(P, l - 1, c, mean, inv);
            const float a = P.bn_w[l - 1][c] * inv;
            cA[c] = a;
            cB[c] = P.bn_b[l - 1][c] - mean * a;
        } else {
            float mean, inv;
            bn_coeffs(P, l, c, mean, inv);
            cM[c] = mean;
            cI[c] = inv;
            cA[c] = P.bn_w[l][c] * inv;
            cK[c] = (float)(dsum(P.sums, l, C, c, 2) * (double)P.invN);
            cK[C + c] = (float)(dsum(P.sums, l, C, c, 3) * (double)P.invN);
            bn_coeffs(P, l - 1, c, mean, inv);
            pM[c] = mean;
            pI[c] = inv;
        }
    }
    for (int i = tid; i < LDW; i += 256) X[(size_t)cells * LDW + i] = 0.f;
    __syncthreads();

    // ---- stage the input operand ---------------------------------------------------------------------------
    {
        const size_t base = (size_t)b * cells * C;
        if (ROLE == ROLE_FWD) {
            const float4 *src = reinterpret_cast<const float4 *>(P.raw[l - 1] + base);
            const bool has_res = ((l - 1) & 1) == 0 && l - 1 >= 2;
            const float4 *res = has_res ? reinterpret_cast<const float4 *>(P.act[l - 3] + base) : nullptr;
            float4 *dst = reinterpret_cast<float4 *>(P.act[l - 1] + base);
            for (int i = tid; i < cells * C4; i += 256) {
                const int pos = i / C4, c = (i - pos * C4) * 4;
                float4 v = src[i];
                v.x = v.x * cA[c] + cB[c];
                v.y = v.y * cA[c + 1] + cB[c + 1];
                v.z = v.z * cA[c + 2] + cB[c + 2];
                v.w = v.w * cA[c + 3] + cB[c + 3];
                if (has_res) {
                    const float4 r = res[i];
                    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                }
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                *reinterpret_cast<float4 *>(X + (size_t)pos * LDW + c) = v;
                if (NT == 1 || (c >> 5) == nt) dst[i] = v;       // each of a board's blocks writes its channel half
            }
        } else {
            const float4 *gs = reinterpret_cast<const float4 *>(P.g[l] + base);
            const float4 *rs = reinterpret_cast<const float4 *>(P.raw[l] + base);
            for (int i = tid; i < cells * C4; i += 256) {
                const int pos = i / C4, c = (i - pos * C4) * 4;
                const float4 gv = gs[i], rv = rs[i];
                float4 v;
                v.x = cA[c] * (gv.x - cK[c] - (rv.x - cM[c]) * cI[c] * cK[C + c]);
                v.y = cA[c + 1] * (gv.y - cK[c + 1] - (rv.y - cM[c + 1]) * cI[c + 1] * cK[C + c + 1]);
                v.z = cA[c + 2] * (gv.z - cK[c + 2] - (rv.z - cM[c + 2]) * cI[c + 2] * cK[C + c + 2]);
                v.w = cA[c + 3] * (gv.w - cK[c + 3] - (rv.w - cM[c + 3]) * cI[c + 3] * cK[C + c + 3]);
                *reinterpret_cast<float4 *>(X + (size_t)pos * LDW + c) = v;
            }
        }
    }
    __syncthreads();

    // ---- k-loop: 9 taps x C / 8 steps of four 32x32x2 MFMAs ----------------------------------------------------
    const int r = wave * 32 + li;
    const bool rvalid = r < cells;
    const int ry = r / N, rx = r - ry * N;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float4 *wl = reinterpret_cast<const float4 *>(ROLE == ROLE_FWD ? P.Wf[l] : P.Wb[l]);
    for (int tap = 0; tap < 9; ++tap) {
        const int yy = ry + tap / 3 - 1, xx = rx + tap % 3 - 1;
        const bool ok = rvalid && yy >= 0 && yy < N && xx >= 0 && xx < N;
        const float *arow = X + (size_t)(ok ? yy * N + xx : cells) * LDW + 4 * lh;
        const float4 *wt = wl + ((size_t)tap * Q * NT + nt) * 64 + lane;
#pragma unroll 4
        for (int q = 0; q < Q; ++q) {
            const float4 bf = wt[(size_t)q * NT * 64];
            const float4 af = *reinterpret_cast<const float4 *>(arow + 8 * q);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc, 0, 0, 0);
        }
    }

    // ---- epilogue: C/D layout col = lane & 31 (output channel), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) ----
    const int co = nt * 32 + li;
    const bool cvalid = co < C;
    float s1 = 0.f, s2 = 0.f;
    const size_t base = (size_t)b * cells * C;
    if (ROLE == ROLE_FWD) {
        float *out = P.raw[l] + base;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
            if (row < cells && cvalid) {
                const float v = acc[i];
                out[(size_t)row * C + co] = v;
                s1 += v;
                s2 += v * v;
            }
        }
    } else {
        float *out = P.g[l - 1] + base;
        const float *pact = P.act[l - 1] + base, *praw = P.raw[l - 1] + base;
        const bool has_skip = ((l - 1) & 1) == 0 && l + 1 <= P.L;
        const float *skip = has_skip ? P.g[l + 1] + base : nullptr;
        const float pm = cvalid ? pM[co] : 0.f, pi = cvalid ? pI[co] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
            if (row < cells && cvalid) {
                const size_t o = (size_t)row * C + co;
                float v = acc[i];
                if (has_skip) v += skip[o];
                v = pact[o] > 0.f ? v : 0.f;
                out[o] = v;
                s1 += v;
                s2 += v * (praw[o] - pm) * pi;
            }
        }
    }
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    if (lh == 0) {
        red[(wave * 32 + li) * 2] = s1;
        red[(wave * 32 + li) * 2 + 1] = s2;
    }
    __syncthreads();
    if (tid < 32 && nt * 32 + tid < C) {
        double a = 0, q = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { a += red[(w * 32 + tid) * 2]; q += red[(w * 32 + tid) * 2 + 1]; }
        const int lay = ROLE == ROLE_FWD ? l : l - 1, k0 = ROLE == ROLE_FWD ? 0 : 2;
        atomicAdd(&P.sums[((size_t)lay * C + nt * 32 + tid) * 4 + k0], a);
        atomicAdd(&P.sums[((size_t)lay * C + nt * 32 + tid) * 4 + k0 + 1], q);
    }
}

// =================================================================================================================
// weight gradient of layer l: dW[co][ci][tap] = sum over boards and positions of draw_l[pos][co] act_{l-1}[pos + tap][ci]
//   grid (9 taps, G board groups); a block walks its boards, M = co, N = ci, K = the board's positions (padded to
//   an even count), A = draw^T and B = the shifted input, both staged in LDS; partial sums per group, reduced by
//   k_trn_update (two stages, fixed order: the step is reproducible)
// =================================================================================================================
template <int C>
__global__ __launch_bounds__(256) void k_trn_wgrad(TrnDev P, int l, int G) {
    constexpr int NT = (C + 31) / 32, T = NT * NT, WPT = 4 / T, LDD = C + 32, C4 = C / 4;
    extern __shared__ __align__(16) float lds[];
    const int N = P.N, cells = P.cells, KP = (cells + 1) & ~1;
    float *D = lds;                                     // [KP][LDD] draw
    float *A = D + (size_t)KP * LDD;                     // [KP][LDD] shifted input
    float *cA = A + (size_t)KP * LDD, *cM = cA + C, *cI = cM + C, *cK = cI + C;      // cK [2][C]
    const int tap = blockIdx.x, grp = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    if (tid < C) {
        const int c = tid;
        float mean, inv;
        bn_coeffs(P, l, c, mean, inv);
        cM[c] = mean;
        cI[c] = inv;
        cA[c] = P.bn_w[l][c] * inv;
        cK[c] = (float)(dsum(P.sums, l, C, c, 2) * (double)P.invN);
        cK[C + c] = (float)(dsum(P.sums, l, C, c, 3) * (double)P.invN);
    }
    // rows beyond the board and columns beyond C are read by the MFMA lanes: keep them zero
    for (int i = tid; i < 2 * KP * LDD; i += 256) lds[i] = 0.f;
    __syncthreads();
    const int tile = WPT == 1 ? wave : 0, sub = WPT == 1 ? 0 : wave;
    const int tm = tile / NT, tn = tile % NT;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int b = grp; b < P.B; b += G) {
        const size_t base = (size_t)b * cells * C;
        const float4 *gs = reinterpret_cast<const float4 *>(P.g[l] + base);
        const float4 *rs = reinterpret_cast<const float4 *>(P.raw[l] + base);
        const float4 *as = reinterpret_cast<const float4 *>(P.act[l - 1] + base);
        for (int i = tid; i < cells * C4; i += 256) {
            const int pos = i / C4, c = (i - pos * C4) * 4;
            const float4 gv = gs[i], rv = rs[i];
            float4 v;
            v.x = cA[c] * (gv.x - cK[c] - (rv.x - cM[c]) * cI[c] * cK[C + c]);
            v.y = cA[c + 1] * (gv.y - cK[c + 1] - (rv.y - cM[c + 1]) * cI[c + 1] * cK[C + c + 1]);
            v.z = cA[c + 2] * (gv.z - cK[c + 2] - (rv.z - cM[c + 2]) * cI[c + 2] * cK[C + c + 2]);
            v.w = cA[c + 3] * (gv.w - cK[c + 3] - (rv.w - cM[c + 3]) * cI[c + 3] * cK[C + c + 3]);
            *reinterpret_cast<float4 *>(D + (size_t)pos * LDD + c) = v;
            const int y = pos / N, x = pos - y * N, yy = y + dy, xx = x + dx;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (yy >= 0 && yy < N && xx >= 0 && xx < N) a = as[(size_t)(yy * N + xx) * C4 + (i - pos * C4)];
            *reinterpret_cast<float4 *>(A + (size_t)pos * LDD + c) = a;
        }
        __syncthreads();
        const float *dp = D + (size_t)lh * LDD + tm * 32 + li, *ap = A + (size_t)lh * LDD + tn * 32 + li;
        for (int s = sub; s < KP / 2; s += WPT)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(dp[(size_t)2 * s * LDD], ap[(size_t)2 * s * LDD], acc, 0, 0, 0);
        __syncthreads();
    }
    float *part = P.wpart + ((size_t)(l - 1) * G + grp) * ((size_t)C * C * 9);
    if (WPT == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = tm * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh, ci = tn * 32 + li;
            if (co < C && ci < C) part[((size_t)co * C + ci) * 9 + tap] = acc[i];
        }
    } else {
        float *redt = lds;                               // [4][1024] (the operand tiles are done with)
#pragma unroll
        for (int i = 0; i < 16; ++i) redt[(size_t)wave * 1024 + i * 64 + lane] = acc[i];
        __syncthreads();
        for (int e = tid; e < 1024; e += 256) {
            const int i = e >> 6, ln = e & 63;
            const int co = (i & 3) + 8 * (i >> 2) + 4 * (ln >> 5), ci = ln & 31;
            if (co < C && ci < C)
                part[((size_t)co * C + ci) * 9 + tap] = redt[e] + redt[1024 + e] + redt[2048 + e] + redt[3072 + e];
        }
    }
}

// =================================================================================================================
// heads, forward part 1: act_L = relu(BN_L(raw_L) + act_{L-2}); the two 1x1 convolutions; their batch sums
// =================================================================================================================
template <int C>
__global__ __launch_bounds__(256) void k_trn_heads_conv(TrnDev P) {
    constexpr int C4 = C / 4, LDX = C + 1;
    extern __shared__ __align__(16) float lds[];
    float *X = lds;                         // [cells][LDX]
    float *cA = X + (size_t)P.cells * LDX, *cB = cA + C;
    float *W = cB + C;                      // [6][C]
    float *red = W + 6 * C;                 // [6][2]
    const int b = blockIdx.x, tid = threadIdx.x, cells = P.cells, L = P.L;
    if (tid < C) {
        float mean, inv;
        bn_coeffs(P, L, tid, mean, inv);
        const float a = P.bn_w[L][tid] * inv;
        cA[tid] = a;
        cB[tid] = P.bn_b[L][tid] - mean * a;
    }
    for (int i = tid; i < 6 * C; i += 256) W[i] = i < 2 * C ? P.vconv[i] : P.pconv[i - 2 * C];
    if (tid < 12) red[tid] = 0.f;
    __syncthreads();
    const size_t base = (size_t)b * cells * C;
    const float4 *src = reinterpret_cast<const float4 *>(P.raw[L] + base);
    const bool has_res = L >= 2;
    const float4 *res = has_res ? reinterpret_cast<const float4 *>(P.act[L - 2] + base) : nullptr;
    float4 *dst = reinterpret_cast<float4 *>(P.act[L] + base);
    for (int i = tid; i < cells * C4; i += 256) {
        const int pos = i / C4, c = (i - pos * C4) * 4;
        float4 v = src[i];
        v.x = v.x * cA[c] + cB[c]; v.y = v.y * cA[c + 1] + cB[c + 1]; v.z = v.z * cA[c + 2] + cB[c + 2]; v.w = v.w * cA[c + 3] + cB[c + 3];
        if (has_res) { const float4 r = res[i]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        dst[i] = v;
        float *x = X + (size_t)pos * LDX + c;
        x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
    }
    __syncthreads();
    float *hraw = P.hraw + (size_t)b * 6 * cells;
    for (int i = tid; i < 6 * cells; i += 256) {
        const int o = i / cells, pos = i - o * cells;
        const float *x = X + (size_t)pos * LDX, *w = W + o * C;
        float s = 0.f;
#pragma unroll 8
        for (int c = 0; c < C; ++c) s += x[c] * w[c];
        hraw[i] = s;
        atomicAdd(&red[o * 2], s);
        atomicAdd(&red[o * 2 + 1], s * s);
    }
    __syncthreads();
    if (tid < 12) atomicAdd(&P.hsums[(tid >> 1) * 4 + (tid & 1)], (double)red[tid]);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// =================================================================================================================
// heads, part 2 (one block per board): BN + ReLU of the head planes, the FC layers, masked log-softmax, the loss and
// the gradient back to the head planes (network.py:77-102, :146-152)
// =================================================================================================================
__global__ __launch_bounds__(256) void k_trn_heads_fc(TrnDev P) {
    __shared__ float ha[6 * 128];           // activated head planes [o][pos] (value 0..1, policy 2..5), flat = the FC inputs
    __shared__ float xh[6 * 128];           // their xhat
    __shared__ float h2[64], dh2s[64], logit[128], dlog[128], gflat[6 * 128];
    __shared__ float cS[6], cT[6], cMn[6], cIv[6];
    __shared__ float sc[8];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cells = P.cells, B = P.B;
    if (tid < 6) {
        const double m = P.hsums[tid * 4] * (double)P.invN, v = P.hsums[tid * 4 + 1] * (double)P.invN - m * m;
        const float inv = (float)(1.0 / sqrt((v > 0 ? v : 0) + TRN_EPS));
        const float gmm = tid < 2 ? P.hbn_w[0][tid] : P.hbn_w[1][tid - 2], bt = tid < 2 ? P.hbn_b[0][tid] : P.hbn_b[1][tid - 2];
        cMn[tid] = (float)m;
        cIv[tid] = inv;
        cS[tid] = gmm * inv;
        cT[tid] = bt - (float)m * gmm * inv;
    }
    __syncthreads();
    const float *hraw = P.hraw + (size_t)b * 6 * cells;
    float *hact = P.hact + (size_t)b * 6 * cells;
    for (int i = tid; i < 6 * cells; i += 256) {
        const int o = i / cells;
        const float r = hraw[i], v = fmaxf(r * cS[o] + cT[o], 0.f);
        ha[i] = v;
        xh[i] = (r - cMn[o]) * cIv[o];
        hact[i] = v;
    }
    __syncthreads();
    // value_fc2 (2 cells -> 64) + ReLU: a wave per output, lanes along the input
    const int KV = 2 * cells, KPp = 4 * cells;
    for (int o = wave; o < 64; o += 4) {
        const float *w = P.fc2w + (size_t)o * KV;
        float s = 0.f;
        for (int i = lane; i < KV; i += 64) s += w[i] * ha[i];
        s = wave_sum(s);
        if (lane == 0) h2[o] = fmaxf(s + P.fc2b[o], 0.f);
    }
    // move_fc (4 cells -> cells)
    for (int t = wave; t < cells; t += 4) {
        const float *w = P.mfw + (size_t)t * KPp;
        float s = 0.f;
        for (int i = lane; i < KPp; i += 64) s += w[i] * ha[2 * cells + i];
        s = wave_sum(s);
        if (lane == 0) logit[t] = s + P.mfb[t];
    }
    __syncthreads();
    if (wave == 0) {
        // value_fc3 + tanh, value loss and its gradient
        float s = wave_sum(P.fc3w[lane] * h2[lane]);
        const float value = tanhf(s + P.fc3b[0]), rew = P.reward[b];
        const float dv = 2.f * (value - rew) / (float)B, d3 = dv * (1.f - value * value);
        dh2s[lane] = h2[lane] > 0.f ? P.fc3w[lane] * d3 : 0.f;
        P.h2[(size_t)b * 64 + lane] = h2[lane];
        P.dh2[(size_t)b * 64 + lane] = dh2s[lane];
        if (lane == 0) {
            P.value[b] = value;
            P.dv3[b] = d3;
            atomicAdd(&P.lossacc[0], (double)((value - rew) * (value - rew)));
        }
    } else if (wave == 1) {
        // gather the legal moves' logits, log-softmax over them (padding entries are -99 in the reference and carry
        // exp(-99 - max) ~ 1e-43 of the sum: below fp32's resolution), policy loss and dL/dlogit by tile
        const int32_t *lm = P.legal + (size_t)b * cells;
        const float *mp = P.prob + (size_t)b * cells;
        float *lp = P.logprob + (size_t)b * cells;
        const int j0 = lane, j1 = lane + 64;
        const int m0 = j0 < cells ? lm[j0] : 0, m1 = j1 < cells ? lm[j1] : 0;
        const float x0 = m0 > 0 ? logit[m0 - 1] : -INFINITY, x1 = m1 > 0 ? logit[m1 - 1] : -INFINITY;
        const float mx = wave_max(fmaxf(x0, x1));
        const float e0 = m0 > 0 ? expf(x0 - mx) : 0.f, e1 = m1 > 0 ? expf(x1 - mx) : 0.f;
        const float lse = mx + logf(wave_sum(e0 + e1));
        const float p0 = j0 < cells ? mp[j0] : 0.f, p1 = j1 < cells ? mp[j1] : 0.f;
        const float l0 = x0 - lse, l1 = x1 - lse;
        const float S = wave_sum((m0 > 0 ? p0 : 0.f) + (m1 > 0 ? p1 : 0.f));
        const float nll = wave_sum((m0 > 0 ? -p0 * l0 : 0.f) + (m1 > 0 ? -p1 * l1 : 0.f));
        dlog[lane] = 0.f;
        dlog[lane + 64] = 0.f;
        if (j0 < cells) lp[j0] = m0 > 0 ? l0 : -99.f - lse;
        if (j1 < cells) lp[j1] = m1 > 0 ? l1 : -99.f - lse;
        __builtin_amdgcn_wave_barrier();
        if (m0 > 0) dlog[m0 - 1] = (expf(l0) * S - p0) / (float)B;
        if (m1 > 0) dlog[m1 - 1] = (expf(l1) * S - p1) / (float)B;
        if (lane == 0) atomicAdd(&P.lossacc[1], (double)nll);
    }
    __syncthreads();
    if (tid < 128) P.dlogit[(size_t)b * 128 + tid] = tid < cells ? dlog[tid] : 0.f;
    // back through the FC layers to the head planes
    for (int i = tid; i < KV; i += 256) {
        float s = 0.f;
        for (int o = 0; o < 64; ++o) s += P.fc2w[(size_t)o * KV + i] * dh2s[o];
        gflat[i] = ha[i] > 0.f ? s : 0.f;
    }
    for (int i = tid; i < KPp; i += 256) {
        float s = 0.f;
        for (int t = 0; t < cells; ++t) s += P.mfw[(size_t)t * KPp + i] * dlog[t];
        gflat[2 * cells + i] = ha[2 * cells + i] > 0.f ? s : 0.f;
    }
    __syncthreads();
    float *g6 = P.g6 + (size_t)b * 6 * cells;
    for (int i = tid; i < 6 * cells; i += 256) g6[i] = gflat[i];
    // the two reductions the head BatchNorms' backward needs
    if (tid < 6 * 32) {
        const int o = tid >> 5, j = tid & 31;
        float a = 0.f, q = 0.f;
        for (int pos = j; pos < cells; pos += 32) { const float gv = gflat[o * cells + pos]; a += gv; q += gv * xh[o * cells + pos]; }
#pragma unroll
        for (int s = 16; s >= 1; s >>= 1) { a += __shfl_xor(a, s); q += __shfl_xor(q, s); }
        if (j == 0) {
            atomicAdd(&P.hsums[o * 4 + 2], (double)a);
            atomicAdd(&P.hsums[o * 4 + 3], (double)q);
        }
    }
    (void)sc;
}

// gradients of the FC layers: one thread per element, the batch is the reduction (fixed order)
struct HeadGradOffs { size_t fc2w, fc2b, fc3w, fc3b, mfw, mfb; };
__global__ __launch_bounds__(256) void k_trn_heads_wgrad(TrnDev P, HeadGradOffs O) {
    const int cells = P.cells, B = P.B, KV = 2 * cells, KPp = 4 * cells;
    const size_t n_mfw = (size_t)cells * KPp, n_fc2 = (size_t)64 * KV;
    const size_t total = n_mfw + n_fc2 + cells + 64 + 64 + 1;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        float s = 0.f;
        if (e < n_mfw) {
            const int t = (int)(e / KPp), i = (int)(e - (size_t)t * KPp);
            for (int b = 0; b < B; ++b) s += P.dlogit[(size_t)b * 128 + t] * P.hact[(size_t)b * 6 * cells + 2 * cells + i];
            P.grad[O.mfw + e] = s;
        } else if (e < n_mfw + n_fc2) {
            const size_t k = e - n_mfw;
            const int o = (int)(k / KV), i = (int)(k - (size_t)o * KV);
            for (int b = 0; b < B; ++b) s += P.dh2[(size_t)b * 64 + o] * P.hact[(size_t)b * 6 * cells + i];
            P.grad[O.fc2w + k] = s;
        } else if (e < n_mfw + n_fc2 + cells) {
            const int t = (int)(e - n_mfw - n_fc2);
            for (int b = 0; b < B; ++b) s += P.dlogit[(size_t)b * 128 + t];
            P.grad[O.mfb + t] = s;
        } else if (e < n_mfw + n_fc2 + cells + 64) {
            const int o = (int)(e - n_mfw - n_fc2 - cells);
            for (int b = 0; b < B; ++b) s += P.dh2[(size_t)b * 64 + o];
            P.grad[O.fc2b + o] = s;
        } else if (e < n_mfw + n_fc2 + cells + 128) {
            const int o = (int)(e - n_mfw - n_fc2 - cells - 64);
            for (int b = 0; b < B; ++b) s += P.dv3[b] * P.h2[(size_t)b * 64 + o];
            P.grad[O.fc3w + o] = s;
        } else {
            for (int b = 0; b < B; ++b) s += P.dv3[b];
            P.grad[O.fc3b] = s;
        }
    }
}

// =================================================================================================================
// heads, backward: BatchNorm backward of the two 1x1 convolutions, their weight gradients (per-board partials), the
// gradient into the tower's output with its ReLU mask and BN_L's two reductions
// =================================================================================================================
template <int C>
__global__ __launch_bounds__(256) void k_trn_heads_bwd(TrnDev P) {
    __shared__ float dh[6 * 128];
    __shared__ float W[6 * C];
    __shared__ float cA[6], cMn[6], cIv[6], cK1[6], cK2[6];
    __shared__ float pM[C], pI[C];
    __shared__ float red[2][256];
    const int b = blockIdx.x, tid = threadIdx.x, cells = P.cells, L = P.L;
    if (tid < 6) {
        const double m = P.hsums[tid * 4] * (double)P.invN, v = P.hsums[tid * 4 + 1] * (double)P.invN - m * m;
        const float inv = (float)(1.0 / sqrt((v > 0 ? v : 0) + TRN_EPS));
        cMn[tid] = (float)m;
        cIv[tid] = inv;
        cA[tid] = (tid < 2 ? P.hbn_w[0][tid] : P.hbn_w[1][tid - 2]) * inv;
        cK1[tid] = (float)(P.hsums[tid * 4 + 2] * (double)P.invN);
        cK2[tid] = (float)(P.hsums[tid * 4 + 3] * (double)P.invN);
    }
    if (tid < C) bn_coeffs(P, L, tid, pM[tid], pI[tid]);
    for (int i = tid; i < 6 * C; i += 256) W[i] = i < 2 * C ? P.vconv[i] : P.pconv[i - 2 * C];
    __syncthreads();
    const float *g6 = P.g6 + (size_t)b * 6 * cells, *hraw = P.hraw + (size_t)b * 6 * cells;
    for (int i = tid; i < 6 * cells; i += 256) {
        const int o = i / cells;
        dh[o * 128 + (i - o * cells)] = cA[o] * (g6[i] - cK1[o] - (hraw[i] - cMn[o]) * cIv[o] * cK2[o]);
    }
    __syncthreads();
    const size_t base = (size_t)b * cells * C;
    const float *act = P.act[L] + base, *raw = P.raw[L] + base;
    float *gL = P.g[L] + base;
    const int c = tid % C;
    float s1 = 0.f, s2 = 0.f;
    for (int pos = tid / C; pos < cells; pos += 256 / C) {
        float d = 0.f;
#pragma unroll
        for (int o = 0; o < 6; ++o) d += dh[o * 128 + pos] * W[o * C + c];
        const size_t idx = (size_t)pos * C + c;
        const float gv = act[idx] > 0.f ? d : 0.f;
        gL[idx] = gv;
        s1 += gv;
        s2 += gv * (raw[idx] - pM[c]) * pI[c];
    }
    red[0][tid] = s1;
    red[1][tid] = s2;
    __syncthreads();
    if (tid < C) {
        double a = 0, q = 0;
        for (int i = tid; i < 256; i += C) { a += red[0][i]; q += red[1][i]; }
        atomicAdd(&P.sums[((size_t)L * C + tid) * 4 + 2], a);
        atomicAdd(&P.sums[((size_t)L * C + tid) * 4 + 3], q);
    }
    // weight gradients of the 1x1 convolutions: this board's share
    float *part = P.hconv_part + (size_t)b * 6 * C;
    for (int i = tid; i < 6 * C; i += 256) {
        const int o = i / C, cc = i - o * C;
        float s = 0.f;
        for (int pos = 0; pos < cells; ++pos) s += dh[o * 128 + pos] * act[(size_t)pos * C + cc];
        part[i] = s;
    }
}

// =================================================================================================================
// stem backward: BN_0 backward, then dL/dT[tap][cell value][cout] (this board's share); k_trn_finalize turns the
// table's gradient into conv1.weight's and the embedding's
// =================================================================================================================
template <int C>
__global__ __launch_bounds__(256) void k_trn_stem_bwd(TrnDev P) {
    extern __shared__ __align__(16) float lds[];
    float *Dr = lds;                         // [cells][C]
    float *cA = Dr + (size_t)P.cells * C, *cM = cA + C, *cI = cM + C, *cK = cI + C;
    __shared__ unsigned char cellv[128];
    const int b = blockIdx.x, tid = threadIdx.x, N = P.N, cells = P.cells;
    if (tid < C) {
        float mean, inv;
        bn_coeffs(P, 0, tid, mean, inv);
        cM[tid] = mean;
        cI[tid] = inv;
        cA[tid] = P.bn_w[0][tid] * inv;
        cK[tid] = (float)(dsum(P.sums, 0, C, tid, 2) * (double)P.invN);
        cK[C + tid] = (float)(dsum(P.sums, 0, C, tid, 3) * (double)P.invN);
    }
    for (int i = tid; i < cells; i += 256) cellv[i] = (unsigned char)P.board[(size_t)b * cells + i];
    __syncthreads();
    const size_t base = (size_t)b * cells * C;
    const float *g0 = P.g[0] + base, *r0 = P.raw[0] + base;
    for (int i = tid; i < cells * C; i += 256) {
        const int c = i % C;
        Dr[i] = cA[c] * (g0[i] - cK[c] - (r0[i] - cM[c]) * cI[c] * cK[C + c]);
    }
    __syncthreads();
    float *part = P.stem_part + (size_t)b * 27 * C;
    for (int i = tid; i < 27 * C; i += 256) {
        const int k = i / C, co = i - k * C, tap = k / 3, v = k - tap * 3;
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        float s = 0.f;
        for (int pos = 0; pos < cells; ++pos) {
            const int y = pos / N, x = pos - y * N, yy = y + dy, xx = x + dx;
            if (yy >= 0 && yy < N && xx >= 0 && xx < N && cellv[yy * N + xx] == v) s += Dr[(size_t)pos * C + co];
        }
        part[i] = s;
    }
}

// =================================================================================================================
// finalize: everything small between the backward pass and the update
//   block 0      stem: dT = sum of the boards' shares -> conv1.weight / encoder.weight gradients
//   block 1      head 1x1 convolutions: sum of the boards' shares; the loss
//   blocks 2..   BatchNorm: dgamma / dbeta from the sums, running statistics, num_batches_tracked
// =================================================================================================================
struct FinalizeArgs {
    size_t g_emb, g_w1, g_vconv, g_pconv;           // offsets into the flat gradient buffer
    const size_t *g_bnw, *g_bnb;                    // [L + 1 + 2] (device): tower layers 0..L, then value_bn1, move_bn1
    float *const *run_mean, *const *run_var;        // [L + 3] (device)
    long long *const *tracked;                      // [L + 3] (device)
};

template <int C>
__global__ __launch_bounds__(256) void k_trn_finalize(TrnDev P, FinalizeArgs F) {
    __shared__ float dT[27 * C];
    const int tid = threadIdx.x, B = P.B, L = P.L;
    if (blockIdx.x == 0) {
        for (int i = tid; i < 27 * C; i += 256) {
            float s = 0.f;
            for (int b = 0; b < B; ++b) s += P.stem_part[(size_t)b * 27 * C + i];
            dT[i] = s;
        }
        __syncthreads();
        for (int i = tid; i < C * 36; i += 256) {       // conv1.weight [co][i4][tap]
            const int co = i / 36, r = i - co * 36, i4 = r / 9, tap = r - i4 * 9;
            float s = 0.f;
            for (int v = 0; v < 3; ++v) s += P.emb[v * 4 + i4] * dT[(tap * 3 + v) * C + co];
            P.grad[F.g_w1 + i] = s;
        }
        if (tid < 12) {                                  // encoder.weight [v][i4]
            const int v = tid / 4, i4 = tid - v * 4;
            float s = 0.f;
            for (int tap = 0; tap < 9; ++tap)
                for (int co = 0; co < C; ++co) s += P.w1[(co * 4 + i4) * 9 + tap] * dT[(tap * 3 + v) * C + co];
            P.grad[F.g_emb + tid] = s;
        }
    } else if (blockIdx.x == 1) {
        for (int i = tid; i < 6 * C; i += 256) {
            float s = 0.f;
            for (int b = 0; b < B; ++b) s += P.hconv_part[(size_t)b * 6 * C + i];
            if (i < 2 * C) P.grad[F.g_vconv + i] = s;
            else P.grad[F.g_pconv + i - 2 * C] = s;
        }
        if (tid == 0) {
            const float lv = (float)(P.lossacc[0] / (double)B), lm = (float)(P.lossacc[1] / (double)B);
            P.loss3[0] = lv + lm;
            P.loss3[1] = lv;
            P.loss3[2] = lm;
        }
    } else {
        // (layer, channel) pairs: tower BN layers have C channels, the two head BNs 2 and 4
        const int total = (L + 1) * C + 6;
        const double Nn = 1.0 / (double)P.invN;
        for (int e = (blockIdx.x - 2) * 256 + tid; e < total; e += (gridDim.x - 2) * 256) {
            int lay, c;
            const double *s;
            if (e < (L + 1) * C) { lay = e / C; c = e - lay * C; s = P.sums + ((size_t)lay * C + c) * 4; }
            else { const int h = e - (L + 1) * C; lay = h < 2 ? L + 1 : L + 2; c = h < 2 ? h : h - 2; s = P.hsums + (size_t)h * 4; }
            P.grad[F.g_bnw[lay] + c] = (float)s[3];
            P.grad[F.g_bnb[lay] + c] = (float)s[2];
            const double mean = s[0] / Nn, var = s[1] / Nn - mean * mean;
            float *rm = F.run_mean[lay] + c, *rv = F.run_var[lay] + c;
            *rm = (float)(0.9 * (double)*rm + 0.1 * mean);
            *rv = (float)(0.9 * (double)*rv + 0.1 * (var > 0 ? var : 0) * Nn / (Nn - 1.0));
            if (c == 0) *F.tracked[lay] += 1;
        }
    }
}

// =================================================================================================================
// SGD update of every tensor, in place (torch.optim.SGD: d = g + wd p; buf = mu buf + d; p -= lr buf).  A block
// handles 1024 consecutive elements of one segment; conv filters take their gradient from the wgrad partial sums.
// =================================================================================================================
struct Segment {
    float *p, *mom;
    size_t n, goff;
    int layer;          // >= 1: tower conv filter of that layer (gradient = sum of partials, packs refreshed); 0: plain
};

template <int C>
__global__ __launch_bounds__(256) void k_trn_update(TrnDev P, const Segment *segs, const int2 *blocks, int G) {
    const int2 bk = blocks[blockIdx.x];
    const Segment S = segs[bk.x];
    const float lr = P.hp[0], mu = P.hp[1], wd = P.hp[2];
    const size_t e0 = (size_t)bk.y * 1024;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t e = e0 + (size_t)k * 256 + threadIdx.x;
        if (e >= S.n) break;
        float gr;
        if (S.layer >= 1) {
            const float *part = P.wpart + (size_t)(S.layer - 1) * G * S.n + e;
            gr = 0.f;
            for (int g = 0; g < G; ++g) gr += part[(size_t)g * S.n];
            P.grad[S.goff + e] = gr;
        } else {
            gr = P.grad[S.goff + e];
        }
        const float p = S.p[e], d = gr + wd * p, buf = mu * S.mom[e] + d, np = p - lr * buf;
        S.mom[e] = buf;
        S.p[e] = np;
    }
}

__global__ void k_trn_set_hp(float *hp, float lr, float mu, float wd) {
    hp[0] = lr;
    hp[1] = mu;
    hp[2] = wd;
}

// The MFMA-order copies of every tower filter, first kernel of each step: whatever wrote the weights last -- this
// trainer's own update, an eager optimizer step on a ragged batch, load_state_dict, a weight broadcast -- the step
// convolves with what the tensors hold NOW.  grid (C C 9 / 256, L).
//   forward pack        [tap][q][ntile][lane = j + 32 h][t] = W[co = 32 ntile + j][ci = 8 q + 4 h + t][tap]
//   backward-data pack  the same order for the transposed, flipped filter W'[n = ci][k = co][8 - tap]
template <int C>
__global__ __launch_bounds__(256) void k_trn_pack(TrnDev P) {
    constexpr int NT = (C + 31) / 32, Q = C / 8;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)C * C * 9) return;
    const int l = blockIdx.y + 1;
    const int tap = (int)(e % 9), ci = (int)(e / 9 % C), co = (int)(e / 9 / C);
    const float v = P.convw[l][e];
    float *wf = const_cast<float *>(P.Wf[l]), *wb = const_cast<float *>(P.Wb[l]);
    wf[((((size_t)tap * Q + (ci >> 3)) * NT + (co >> 5)) * 64 + (co & 31) + 32 * ((ci >> 2) & 1)) * 4 + (ci & 3)] = v;
    wb[((((size_t)(8 - tap) * Q + (co >> 3)) * NT + (ci >> 5)) * 64 + (ci & 31) + 32 * ((co >> 2) & 1)) * 4 + (co & 3)] = v;
}

// =================================================================================================================
// host side
// =================================================================================================================
struct Bound {
    std::string name;
    void *ptr = nullptr;
    float *mom = nullptr;
    size_t n = 0, goff = 0;
};

struct AzxTrain {
    TrnDev d;
    int device = 0, G = 1;
    std::vector<void *> allocs;
    std::map<std::string, Bound> bound;
    bool is_bound = false;
    // device-side tables
    float **raw_h = nullptr, **act_h = nullptr, **g_h = nullptr, **Wf_h = nullptr, **Wb_h = nullptr;
    std::vector<float *> raw, act, g, Wf, Wb;
    Segment *segs = nullptr;
    int2 *blocks = nullptr;
    int n_blocks = 0;
    FinalizeArgs fin;
    HeadGradOffs hoffs;
    float *hp_dev = nullptr;
    int32_t *in_board = nullptr, *in_legal = nullptr;
    float *in_prob = nullptr, *in_reward = nullptr;
    size_t zero_bytes = 0;       // sums + hsums + lossacc, contiguous
    size_t grad_floats = 0;
    std::map<std::string, std::pair<void *, size_t>> dbg;      // name -> (device ptr, bytes)
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipStream_t cap = nullptr, side = nullptr;
    std::vector<hipEvent_t> events;
    bool use_graph = true;
};

template <typename T>
static T *talloc(AzxTrain *t, size_t count, bool zero = true) {
    void *p = nullptr;
    const size_t bytes = std::max<size_t>(count * sizeof(T), 16);
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    if (zero) (void)hipMemset(p, 0, bytes);
    t->allocs.push_back(p);
    return reinterpret_cast<T *>(p);
}

template <typename T>
static T *upload_table(AzxTrain *t, const std::vector<T> &v) {
    T *p = talloc<T>(t, v.size(), false);
    if (p && hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return p;
}

int azx_trn_create(AzxTrain **out, int N, int blocks, int chans, int batch, int device) {
    if (N < 2 || N > 11) return tfail(AZX_EINVAL, "train: the native step covers boards up to 11x11 (121 cells = four 32-row MFMA tiles)");
    if (chans != 16 && chans != 32 && chans != 64) return tfail(AZX_EINVAL, "train: base_chans must be 16, 32 or 64");
    if (blocks < 1 || batch < 1) return tfail(AZX_EINVAL, "train: num_blocks and batch must be positive");
    AzxTrain *t = new AzxTrain();
    memset(&t->d, 0, sizeof t->d);
    t->device = device;
    TrnDev &d = t->d;
    d.N = N; d.cells = N * N; d.C = chans; d.L = 2 * blocks; d.B = batch;
    d.invN = (float)(1.0 / ((double)batch * d.cells));
    t->G = std::min(batch, TRN_WG_GROUPS);
    const int L = d.L, C = chans, cells = d.cells, B = batch;
    const size_t A = (size_t)B * cells * C;
    t->raw.resize(L + 1); t->act.resize(L + 1); t->g.resize(L + 1); t->Wf.assign(L + 1, nullptr); t->Wb.assign(L + 1, nullptr);
    bool ok = true;
    for (int l = 0; l <= L; ++l) {
        ok = ok && (t->raw[l] = talloc<float>(t, A)) && (t->act[l] = talloc<float>(t, A)) && (t->g[l] = talloc<float>(t, A));
        if (l >= 1) {
            const size_t wn = (size_t)9 * (C / 8) * ((C + 31) / 32) * 64 * 4;
            ok = ok && (t->Wf[l] = talloc<float>(t, wn)) && (t->Wb[l] = talloc<float>(t, wn));
        }
    }
    // sums | hsums | lossacc contiguous: one memset per step
    const size_t nsum = (size_t)(L + 1) * C * 4 + 6 * 4 + 2;
    double *z = ok ? talloc<double>(t, nsum) : nullptr;
    ok = ok && z;
    if (ok) {
        d.sums = z; d.hsums = z + (size_t)(L + 1) * C * 4; d.lossacc = d.hsums + 24;
        t->zero_bytes = nsum * sizeof(double);
    }
    ok = ok && (d.hraw = talloc<float>(t, (size_t)B * 6 * cells)) && (d.hact = talloc<float>(t, (size_t)B * 6 * cells)) &&
         (d.g6 = talloc<float>(t, (size_t)B * 6 * cells)) && (d.h2 = talloc<float>(t, (size_t)B * 64)) &&
         (d.dh2 = talloc<float>(t, (size_t)B * 64)) && (d.dv3 = talloc<float>(t, B)) &&
         (d.dlogit = talloc<float>(t, (size_t)B * 128)) && (d.value = talloc<float>(t, B)) &&
         (d.logprob = talloc<float>(t, (size_t)B * cells)) && (d.loss3 = talloc<float>(t, 4)) &&
         (d.wpart = talloc<float>(t, (size_t)L * t->G * C * C * 9)) && (d.stem_part = talloc<float>(t, (size_t)B * 27 * C)) &&
         (d.hconv_part = talloc<float>(t, (size_t)B * 6 * C)) &&
         (t->in_board = talloc<int32_t>(t, (size_t)B * cells)) && (t->in_legal = talloc<int32_t>(t, (size_t)B * cells)) &&
         (t->in_prob = talloc<float>(t, (size_t)B * cells)) && (t->in_reward = talloc<float>(t, B)) &&
         (t->hp_dev = talloc<float>(t, 4));
    if (!ok) {
        azx_trn_destroy(t);
        return tfail(AZX_ENOMEM, "train: hipMalloc failed");
    }
    d.board = t->in_board; d.legal = t->in_legal; d.prob = t->in_prob; d.reward = t->in_reward; d.hp = t->hp_dev;
    {
        std::vector<float *> v;
        v = t->raw; d.raw = upload_table(t, v);
        v = t->act; d.act = upload_table(t, v);
        v = t->g; d.g = upload_table(t, v);
        std::vector<const float *> w(t->Wf.begin(), t->Wf.end());
        d.Wf = upload_table(t, w);
        w.assign(t->Wb.begin(), t->Wb.end());
        d.Wb = upload_table(t, w);
        if (!d.raw || !d.act || !d.g || !d.Wf || !d.Wb) {
            azx_trn_destroy(t);
            return tfail(AZX_ENOMEM, "train: uploading the buffer tables failed");
        }
    }
    for (int l = 0; l <= L; ++l) {
        char nm[32];
        snprintf(nm, sizeof nm, "raw%d", l); t->dbg[nm] = {t->raw[l], A * 4};
        snprintf(nm, sizeof nm, "act%d", l); t->dbg[nm] = {t->act[l], A * 4};
        snprintf(nm, sizeof nm, "g%d", l); t->dbg[nm] = {t->g[l], A * 4};
    }
    t->dbg["sums"] = {d.sums, (size_t)(L + 1) * C * 4 * 8};
    t->dbg["hsums"] = {d.hsums, 24 * 8};
    t->dbg["hraw"] = {d.hraw, (size_t)B * 6 * cells * 4};
    t->dbg["hact"] = {d.hact, (size_t)B * 6 * cells * 4};
    t->dbg["hg"] = {d.g6, (size_t)B * 6 * cells * 4};
    t->dbg["dlogit"] = {d.dlogit, (size_t)B * 128 * 4};
    t->use_graph = !(getenv("AZX_TRAIN_GRAPH") && !strcmp(getenv("AZX_TRAIN_GRAPH"), "0"));
    if (hipDeviceSynchronize() != hipSuccess) {
        azx_trn_destroy(t);
        return tfail(AZX_EHIP, "train: device sync after the allocations failed");
    }
    *out = t;
    return AZX_OK;
}

void azx_trn_destroy(AzxTrain *t) {
    if (!t) return;
    if (t->exec) (void)hipGraphExecDestroy(t->exec);
    if (t->graph) (void)hipGraphDestroy(t->graph);
    for (hipEvent_t e : t->events) (void)hipEventDestroy(e);
    if (t->cap) (void)hipStreamDestroy(t->cap);
    if (t->side) (void)hipStreamDestroy(t->side);
    for (void *p : t->allocs) (void)hipFree(p);
    delete t;
}

int azx_trn_bind(AzxTrain *t, int n, const char *const *names, void *const *ptrs, const int64_t *counts,
                 void *const *momentum) {
    TrnDev &d = t->d;
    const int L = d.L, C = d.C, cells = d.cells;
    if (t->exec) { (void)hipGraphExecDestroy(t->exec); t->exec = nullptr; }
    if (t->graph) { (void)hipGraphDestroy(t->graph); t->graph = nullptr; }
    t->bound.clear();
    for (int i = 0; i < n; ++i) {
        Bound b;
        b.name = names[i];
        b.ptr = ptrs[i];
        b.mom = momentum ? static_cast<float *>(momentum[i]) : nullptr;
        b.n = (size_t)counts[i];
        t->bound[b.name] = b;
    }
    std::string err;
    auto need = [&](const std::string &name, size_t count, bool param) -> Bound * {
        auto it = t->bound.find(name);
        if (it == t->bound.end() || it->second.n != count || !it->second.ptr || (param && !it->second.mom)) {
            if (err.empty()) err = "train: tensor '" + name + "' missing, of the wrong size, or (a parameter) without a momentum buffer";
            return nullptr;
        }
        return &it->second;
    };
    // parameters in a fixed order: their gradients live at goff in one flat buffer
    std::vector<Bound *> params;
    auto P = [&](const std::string &name, size_t count) -> Bound * {
        Bound *b = need(name, count, true);
        if (b) params.push_back(b);
        return b;
    };
    Bound *emb = P("encoder.weight", 12), *w1 = P("conv1.weight", (size_t)C * 36);
    std::vector<Bound *> bnw(L + 3), bnb(L + 3), rmean(L + 3), rvar(L + 3), trk(L + 3), conv(L + 1, nullptr);
    auto BN = [&](int idx, const std::string &pre, int c) {
        bnw[idx] = P(pre + ".weight", c);
        bnb[idx] = P(pre + ".bias", c);
        rmean[idx] = need(pre + ".running_mean", c, false);
        rvar[idx] = need(pre + ".running_var", c, false);
        trk[idx] = need(pre + ".num_batches_tracked", 1, false);
    };
    BN(0, "bn1", C);
    for (int l = 1; l <= L; ++l) {
        char nm[96];
        snprintf(nm, sizeof nm, "resblocks.%d.conv%d.weight", (l - 1) / 2, (l - 1) % 2 + 1);
        conv[l] = P(nm, (size_t)C * C * 9);
        snprintf(nm, sizeof nm, "resblocks.%d.bn%d", (l - 1) / 2, (l - 1) % 2 + 1);
        BN(l, nm, C);
    }
    Bound *vconv = P("value_conv1.weight", (size_t)2 * C);
    BN(L + 1, "value_bn1", 2);
    Bound *fc2w = P("value_fc2.weight", (size_t)64 * 2 * cells), *fc2b = P("value_fc2.bias", 64);
    Bound *fc3w = P("value_fc3.weight", 64), *fc3b = P("value_fc3.bias", 1);
    Bound *pconv = P("move_conv1.weight", (size_t)4 * C);
    BN(L + 2, "move_bn1", 4);
    Bound *mfw = P("move_fc.weight", (size_t)cells * 4 * cells), *mfb = P("move_fc.bias", cells);
    if (!err.empty()) return tfail(AZX_EINVAL, err);
    size_t off = 0;
    for (Bound *b : params) { b->goff = off; off += (b->n + 3) & ~(size_t)3; }
    if (off > t->grad_floats) {
        d.grad = talloc<float>(t, off);
        if (!d.grad) return tfail(AZX_ENOMEM, "train: allocating the gradient buffer failed");
        t->grad_floats = off;
    }
    for (Bound *b : params) t->dbg["grad:" + b->name] = {d.grad + b->goff, b->n * 4};
    // device-side views
    d.emb = (const float *)emb->ptr; d.w1 = (const float *)w1->ptr;
    d.vconv = (const float *)vconv->ptr; d.pconv = (const float *)pconv->ptr;
    d.hbn_w[0] = (const float *)bnw[L + 1]->ptr; d.hbn_b[0] = (const float *)bnb[L + 1]->ptr;
    d.hbn_w[1] = (const float *)bnw[L + 2]->ptr; d.hbn_b[1] = (const float *)bnb[L + 2]->ptr;
    d.fc2w = (const float *)fc2w->ptr; d.fc2b = (const float *)fc2b->ptr; d.fc3w = (const float *)fc3w->ptr;
    d.fc3b = (const float *)fc3b->ptr; d.mfw = (const float *)mfw->ptr; d.mfb = (const float *)mfb->ptr;
    {
        std::vector<const float *> w(L + 1), bb(L + 1);
        for (int l = 0; l <= L; ++l) { w[l] = (const float *)bnw[l]->ptr; bb[l] = (const float *)bnb[l]->ptr; }
        d.bn_w = upload_table(t, w);
        d.bn_b = upload_table(t, bb);
        std::vector<size_t> gw(L + 3), gb(L + 3);
        std::vector<float *> rm(L + 3), rv(L + 3);
        std::vector<long long *> tk(L + 3);
        for (int i = 0; i < L + 3; ++i) {
            gw[i] = bnw[i]->goff; gb[i] = bnb[i]->goff;
            rm[i] = (float *)rmean[i]->ptr; rv[i] = (float *)rvar[i]->ptr; tk[i] = (long long *)trk[i]->ptr;
        }
        t->fin.g_emb = emb->goff; t->fin.g_w1 = w1->goff; t->fin.g_vconv = vconv->goff; t->fin.g_pconv = pconv->goff;
        t->fin.g_bnw = upload_table(t, gw); t->fin.g_bnb = upload_table(t, gb);
        t->fin.run_mean = upload_table(t, rm); t->fin.run_var = upload_table(t, rv); t->fin.tracked = upload_table(t, tk);
        if (!d.bn_w || !d.bn_b || !t->fin.g_bnw || !t->fin.g_bnb || !t->fin.run_mean || !t->fin.run_var || !t->fin.tracked)
            return tfail(AZX_ENOMEM, "train: uploading the parameter tables failed");
    }
    t->hoffs = {fc2w->goff, fc2b->goff, fc3w->goff, fc3b->goff, mfw->goff, mfb->goff};
    // update segments and the block table
    {
        std::vector<Segment> segs;
        std::vector<int2> blocks;
        for (Bound *b : params) {
            Segment s;
            s.p = (float *)b->ptr; s.mom = b->mom; s.n = b->n; s.goff = b->goff; s.layer = 0;
            for (int l = 1; l <= L; ++l) if (conv[l] == b) s.layer = l;
            const int si = (int)segs.size();
            segs.push_back(s);
            for (size_t e = 0; e < b->n; e += 1024) blocks.push_back(make_int2(si, (int)(e / 1024)));
        }
        t->segs = upload_table(t, segs);
        t->blocks = upload_table(t, blocks);
        t->n_blocks = (int)blocks.size();
        if (!t->segs || !t->blocks) return tfail(AZX_ENOMEM, "train: uploading the update tables failed");
    }
    {
        std::vector<const float *> cw(L + 1, nullptr);
        for (int l = 1; l <= L; ++l) cw[l] = (const float *)conv[l]->ptr;
        d.convw = upload_table(t, cw);
        if (!d.convw) return tfail(AZX_ENOMEM, "train: uploading the filter table failed");
    }
    if (hipDeviceSynchronize() != hipSuccess) return tfail(AZX_EHIP, "train: device sync after the bind failed");
    t->is_bound = true;
    return AZX_OK;
}

int azx_trn_inputs(AzxTrain *t, int32_t **board, int32_t **legal_moves, float **moves_prob, float **reward) {
    if (board) *board = t->in_board;
    if (legal_moves) *legal_moves = t->in_legal;
    if (moves_prob) *moves_prob = t->in_prob;
    if (reward) *reward = t->in_reward;
    return AZX_OK;
}

int azx_trn_outputs(AzxTrain *t, float **loss3, float **value, float **logprob) {
    if (loss3) *loss3 = t->d.loss3;
    if (value) *value = t->d.value;
    if (logprob) *logprob = t->d.logprob;
    return AZX_OK;
}

// the step as a sequence of launches on `st`, the weight-gradient passes forked onto `side`
template <int C>
static int enqueue_step(AzxTrain *t, hipStream_t st, hipStream_t side, bool fork) {
    const TrnDev &d = t->d;
    const int L = d.L, B = d.B, cells = d.cells, G = t->G;
    constexpr int NT = (C + 31) / 32;
    if (hipMemsetAsync(d.sums, 0, t->zero_bytes, st) != hipSuccess) return tfail(AZX_EHIP, "train: memset failed");
    hipLaunchKernelGGL(k_trn_pack<C>, dim3((C * C * 9 + 255) / 256, L), dim3(256), 0, st, d);
    hipLaunchKernelGGL(k_trn_stem_fwd<C>, dim3(B), dim3(256), 0, st, d);
    const size_t conv_lds = ((size_t)(cells + 1) * (C + 4) + 9 * C + 256) * sizeof(float);
    for (int l = 1; l <= L; ++l)
        hipLaunchKernelGGL((k_trn_conv<C, ROLE_FWD>), dim3(NT, B), dim3(256), conv_lds, st, d, l);
    const size_t hc_lds = ((size_t)cells * (C + 1) + 2 * C + 6 * C + 16) * sizeof(float);
    hipLaunchKernelGGL(k_trn_heads_conv<C>, dim3(B), dim3(256), hc_lds, st, d);
    hipLaunchKernelGGL(k_trn_heads_fc, dim3(B), dim3(256), 0, st, d);
    size_t ev = 0;
    auto next_event = [&]() -> hipEvent_t {
        if (ev == t->events.size()) {
            hipEvent_t e;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
            t->events.push_back(e);
        }
        return t->events[ev++];
    };
    hipStream_t ws = fork ? side : st;
    auto fork_side = [&]() -> bool {       // what `side` launches next depends on everything queued on `st` so far
        if (!fork) return true;
        hipEvent_t e = next_event();
        return e && hipEventRecord(e, st) == hipSuccess && hipStreamWaitEvent(side, e, 0) == hipSuccess;
    };
    if (!fork_side()) return tfail(AZX_EHIP, "train: forking the weight-gradient stream failed");
    hipLaunchKernelGGL(k_trn_heads_wgrad, dim3(256), dim3(256), 0, ws, d, t->hoffs);
    hipLaunchKernelGGL(k_trn_heads_bwd<C>, dim3(B), dim3(256), 0, st, d);
    const int KP = (cells + 1) & ~1;
    const size_t wg_lds = std::max((size_t)2 * KP * (C + 32) + 5 * C, (size_t)4096) * sizeof(float);
    for (int l = L; l >= 1; --l) {
        // g_l and BN_l's sums are complete here: the weight gradient of layer l runs beside the data chain
        if (!fork_side()) return tfail(AZX_EHIP, "train: forking the weight-gradient stream failed");
        hipLaunchKernelGGL(k_trn_wgrad<C>, dim3(9, G), dim3(256), wg_lds, ws, d, l, G);
        hipLaunchKernelGGL((k_trn_conv<C, ROLE_BWD>), dim3(NT, B), dim3(256), conv_lds, st, d, l);
    }
    const size_t sb_lds = ((size_t)cells * C + 5 * C) * sizeof(float);
    hipLaunchKernelGGL(k_trn_stem_bwd<C>, dim3(B), dim3(256), sb_lds, st, d);
    if (fork) {
        hipEvent_t e = next_event();
        if (!e || hipEventRecord(e, side) != hipSuccess || hipStreamWaitEvent(st, e, 0) != hipSuccess)
            return tfail(AZX_EHIP, "train: joining the weight-gradient stream failed");
    }
    const int fin_blocks = 2 + ((L + 1) * C + 6 + 255) / 256;
    hipLaunchKernelGGL(k_trn_finalize<C>, dim3(fin_blocks), dim3(256), 0, st, d, t->fin);
    hipLaunchKernelGGL(k_trn_update<C>, dim3(t->n_blocks), dim3(256), 0, st, d, (const Segment *)t->segs, (const int2 *)t->blocks, G);
    if (hipGetLastError() != hipSuccess) return tfail(AZX_EHIP, "train: a kernel of the step failed to launch");
    return AZX_OK;
}

static int raise_limits(int C) {
    const int cap = 128 * 1024;     // the largest user (k_trn_wgrad<64>) takes 94 KB; some kernels add static LDS on top
    const void *f64[] = {(const void *)k_trn_conv<64, ROLE_FWD>, (const void *)k_trn_conv<64, ROLE_BWD>, (const void *)k_trn_wgrad<64>,
                         (const void *)k_trn_heads_conv<64>, (const void *)k_trn_stem_bwd<64>};
    const void *f32[] = {(const void *)k_trn_conv<32, ROLE_FWD>, (const void *)k_trn_conv<32, ROLE_BWD>, (const void *)k_trn_wgrad<32>,
                         (const void *)k_trn_heads_conv<32>, (const void *)k_trn_stem_bwd<32>};
    const void *f16[] = {(const void *)k_trn_conv<16, ROLE_FWD>, (const void *)k_trn_conv<16, ROLE_BWD>, (const void *)k_trn_wgrad<16>,
                         (const void *)k_trn_heads_conv<16>, (const void *)k_trn_stem_bwd<16>};
    const void **f = C == 64 ? f64 : C == 32 ? f32 : f16;
    for (int i = 0; i < 5; ++i)
        if (hipFuncSetAttribute(f[i], hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess)
            return tfail(AZX_EHIP, "train: raising a kernel's dynamic LDS limit failed");
    return AZX_OK;
}

static int enqueue_any(AzxTrain *t, hipStream_t st, hipStream_t side, bool fork) {
    switch (t->d.C) {
        case 64: return enqueue_step<64>(t, st, side, fork);
        case 32: return enqueue_step<32>(t, st, side, fork);
        default: return enqueue_step<16>(t, st, side, fork);
    }
}

int azx_trn_step(AzxTrain *t, float lr, float momentum, float weight_decay, hipStream_t st) {
    if (!t->is_bound) return tfail(AZX_ESTATE, "train: azx_train_bind has not been called");
    if (!t->cap) {
        if (int rc = raise_limits(t->d.C)) return rc;
        if (hipStreamCreateWithFlags(&t->cap, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&t->side, hipStreamNonBlocking) != hipSuccess)
            return tfail(AZX_EHIP, "train: creating the capture streams failed");
    }
    // the hyper-parameters travel through device memory, so one captured graph serves every learning rate
    hipLaunchKernelGGL(k_trn_set_hp, dim3(1), dim3(1), 0, st, t->hp_dev, lr, momentum, weight_decay);
    if (!t->use_graph) {
        // plain launches: the weight-gradient passes fork onto the side stream after an event on `st`
        return enqueue_any(t, st, t->side, true);
    }
    if (!t->exec) {
        if (hipStreamBeginCapture(t->cap, hipStreamCaptureModeRelaxed) != hipSuccess)
            return tfail(AZX_EHIP, "train: hipStreamBeginCapture failed");
        int rc = enqueue_any(t, t->cap, t->side, true);
        hipGraph_t graph = nullptr;
        const hipError_t ce = hipStreamEndCapture(t->cap, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (ce != hipSuccess || !graph) return tfail(AZX_EHIP, std::string("train: hipStreamEndCapture failed: ") + hipGetErrorString(ce));
        t->graph = graph;
        if (hipGraphInstantiate(&t->exec, graph, nullptr, nullptr, 0) != hipSuccess)
            return tfail(AZX_EHIP, "train: hipGraphInstantiate failed");
    }
    if (hipGraphLaunch(t->exec, st) != hipSuccess) return tfail(AZX_EHIP, "train: hipGraphLaunch failed");
    return AZX_OK;
}

int azx_trn_debug(AzxTrain *t, const char *name, void *out, int64_t cap, int64_t *nbytes) {
    auto it = t->dbg.find(name);
    if (it == t->dbg.end()) return tfail(AZX_EINVAL, std::string("train: no buffer named '") + name + "'");
    *nbytes = (int64_t)it->second.second;
    if (out && cap >= *nbytes) {
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out, it->second.first, it->second.second, hipMemcpyDeviceToHost) != hipSuccess)
            return tfail(AZX_EHIP, "train: copying a buffer to the host failed");
    }
    return AZX_OK;
}
