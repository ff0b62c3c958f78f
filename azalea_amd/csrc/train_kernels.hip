// train_kernels.hip -- one optimizer step of the reference's trainer on gfx950 (MI355X), hand-written.
//
// Restates azalea/policy_trainer.py:123-142 (supervised_step: zero_grad, forward, loss, backward, SGD step) over
// azalea/network.py:68-102 (Network.forward in TRAIN mode -- BatchNorm normalises with the batch statistics and
// updates its running ones -- and the loss of Network.run: mse(value, reward) - sum(moves_prob * logprob) / B) and
// :134-152 (HexNetwork: embedding, policy FC, legal-move gather, log_softmax), with torch.optim.SGD's update
// (momentum, weight decay, no dampening / nesterov: d = g + wd p; buf = mu buf + d; p -= lr buf).
//
// Design.  At the reference's batch of 128 boards a step is 41 GFLOP of 3x3 convolutions in a
// chain of ~40 dependent layer passes: latency-bound, not throughput-bound.  So: one workgroup per (board, 32 output
// channels) so that a layer pass fills all 256 CUs with 4 waves each, every elementwise stage fused into the
// producer's epilogue or the consumer's prologue, the filter gradients on a stream of their own beside the data chain,
// and the convolutions of all three passes on the split-f16 arithmetic of the self-play tower (hi + lo f16 operands,
// three v_mfma_f32_32x32x16_f16 per k-step, fp32 accumulate) with every operand scaled per layer by a power of two
// (ROLE_FWD16 / ROLE_BWD16 / k_trn_wgrad16 below); the exact-fp32 kernels (v_mfma_f32_32x32x2_f32) stay selectable
// per pass (AZX_TRAIN_FWD / _BWD / _WGRAD=fp32):
//   prep     : k_trn_prep (max |filter| per layer, fp32 MFMA-order filter copies for the fp32 roles, accumulators
//              zeroed, stem table, hyper-parameters from a pinned ring)
//   forward  : k_trn_stem_fwd (+ the split-f16 filter fragments and the per-layer scales), L x k_trn_conv<FWD16>
//              (prologue: BN(batch stats) + residual + ReLU of the INPUT, written out once for the backward pass;
//              epilogue: raw output + per-board per-channel sum / sum of squares), heads
//   backward : heads, L x k_trn_conv<BWD16> (prologue: BatchNorm backward of the incoming gradient; implicit GEMM with
//              the flipped / transposed filters; epilogue: skip-connection add, ReLU mask, the next BatchNorm's two
//              reductions, max |g|), L x k_trn_wgrad16 (split over boards and channel-tile pairs, partial copies), stem
//   update   : k_trn_update twice (the tower filters -- second stage of the filter-gradient reduction in a fixed order --
//              behind the last filter-gradient kernel on its stream; everything else at the end of the data chain, each
//              element's gradient taken from where the backward pass left it -- BatchNorm sums, the stem table's and the
//              head convolutions' accumulators -- with the running statistics and the loss on the way): SGD IN PLACE in
//              the trainer's torch tensors
// Layouts: activations [B][cells][C] fp32; per BatchNorm layer per-board partial pairs (x, x^2) and (g, g xhat), summed
// in a fixed order by the kernels that consume them (sum_partials), the totals filed as four f64 per channel.
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
#define TRN_MAXL 38               // tower conv layers (19 blocks): the per-layer pointers live IN the kernel argument struct --
                                  // a pointer fetched from a table in memory is a generic pointer to the compiler, every
                                  // access through it a FLAT instruction whose completion the LDS waits then also wait for
#define TRN_HP_SLOTS 256          // hyper-parameter ring: the host may run this many steps ahead of the device
#define TRN_SMALL_THREADS 1024    // the elementwise kernels around the tower: 16 waves per CU hide their loads' latency
// ... except the two that open a chain of them (k_trn_heads_conv behind the forward tower, k_trn_stem_bwd behind the backward
// one): 512 threads.  A 1024-thread block of theirs is 4 waves x 80-88 registers per SIMD -- more than the HALF of a CU's
// register file one retiring self-play tower block frees -- so beside self-play (azalea_amd/play_ahead.py) it waited until
// BOTH tower blocks of some CU happened to have gone: 690 us and 166 us a step where the kernels take 12 and 13 alone
// (profiles/r6_train_loop_overlap_timeline.txt).  With 8 waves a block fits the freed half; alone they cost +1-2 us.
#define TRN_MID_THREADS 512
#define TW_GSLOTS 32
#define TRN_REP 8                  // copies of the small f64-atomic accumulators
#define TRN_WG_GROUPS 64          // k_trn_wgrad: board groups (x 4 channel-tile pairs at C = 64: 256 workgroups)
#define TRN_PRESUM_BATCH 256      // above this batch a totals kernel sums a layer's per-board partial pairs once (see k_trn_totals)

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
    float *embw1_old;              // [12 + C 36] their values at the start of the step (k_trn_prep): each one's gradient
                                   // is made from the other while k_trn_update is already moving both
    const float *bn_w[TRN_MAXL + 1], *bn_b[TRN_MAXL + 1];      // [L + 1] -> [C]
    const float *vconv, *pconv;    // [2][C], [4][C]
    const float *hbn_w[2], *hbn_b[2];            // value_bn1 (2), move_bn1 (4)
    const float *fc2w, *fc2b, *fc3w, *fc3b, *mfw, *mfb;
    // work buffers
    float *raw[TRN_MAXL + 1], *act[TRN_MAXL + 1], *g[TRN_MAXL + 1];   // [L + 1] -> [B][cells][C]
    float *Wf[TRN_MAXL + 1], *Wb[TRN_MAXL + 1];  // (index 1..L) MFMA-order filters: forward / backward-data
    const float *convw[TRN_MAXL + 1];            // (index 1..L) the filters themselves (torch layout [co][ci][3][3])
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
    // (TRN_REP copies each, board b adds into copy b % TRN_REP: 2 K double atomics per cache line and launch cost a
    // kernel ~4 us -- see the BatchNorm sums --, 256 do not; k_trn_update adds the copies up)
    double *stem_dT;               // [TRN_REP][27][C]  dL/d(stem table), summed over the boards (f64 atomics)
    double *hconv_acc;             // [TRN_REP][6][C]   gradient of the two 1x1 head convolutions
    float *grad;                   // flat gradient buffer (offsets in the segment table)
    float *hp;                     // lr, momentum, weight decay (written by k_trn_prep from the host's ring)
    float *stemT;                  // [27][C] embedding folded through conv1: table[tap * 3 + cell value][cout]
    const float *hp_ring;          // pinned host memory, TRN_HP_SLOTS x 4 floats: the host writes slot (step % slots)
    unsigned int *step_ctr;        // steps run so far (device side of the same count)
    unsigned short *Wf16[TRN_MAXL + 1];   // (index 1..L) forward filters as hi / lo f16 MFMA fragments (ROLE_FWD16)
    unsigned short *Wb16[TRN_MAXL + 1];   // the same for the backward-data pass (transposed, taps flipped)
    unsigned int *gmax;                   // [L + 1] bits of max |g_l| (non-negative floats order as their bits)
    unsigned int *gslots;                 // wide step: [L + 1][TW_GSLOTS] -- the backward convolution's blocks file max |g_{l-1}| by
                                          // atomicMax on slot (block & 31) instead of all on one word; k_tw_bnbwd folds them
    // the last tower layer's tensors once more, as plain members: P.raw[P.L] is two dependent scalar loads
    float *rawL, *actL, *actLm2, *gL;
    const float *bnwL, *bnbL;
    float4 *fsc;                          // [L + 1] (act scale, 1 / (act scale x filter scale), filter scale, 1 / filter scale): k_trn_stem_fwd
    float2 *pstat, *pgsum;                // [L + 1][B][C] per-board (sum, sum of squares) of raw_l / (sum g_l, sum g_l xhat_l)
    float *wpmax;                         // [L + 1][C C 9 / 256] max |filter| per k_trn_prep block
    unsigned int *wmax;                   // [L + 1] bits of max |filter of layer l| (k_trn_stem_fwd, from wpmax)
    double *zero_base;             // the per-step accumulators (sums | hsums | lossacc | stem_dT | hconv_acc), zero_count doubles
    int zero_count;
    // wide towers (C = 128 / 256: the "wide tower" section below)
    float2 *bsc;                   // [L + 1] (scale of layer l's BatchNorm-backward image, 1 / (that x the filter scale))
    int dl_stride;                 // row stride of dlogit: 128 (boards up to 11x11), 192 beyond
    int presum;                    // batches beyond TRN_PRESUM_BATCH: the batch sums come from k_trn_totals, not from every consumer
};

__device__ __forceinline__ double dsum(const double *s, int l, int C, int c, int k) { return s[((size_t)l * C + c) * 4 + k]; }

// per-channel BatchNorm coefficients of layer l from its batch sums
__device__ __forceinline__ void bn_coeffs(const TrnDev &P, int l, int c, float &mean, float &inv) {
    const double m = dsum(P.sums, l, P.C, c, 0) * (double)P.invN;
    const double v = dsum(P.sums, l, P.C, c, 1) * (double)P.invN - m * m;
    mean = (float)m;
    inv = (float)(1.0 / sqrt((v > 0 ? v : 0) + TRN_EPS));
}

// The batch sums of a layer are kept as per-board partial pairs (plain stores by the kernel that produces the tensor:
// 32 K double atomics on 16 cache lines cost a convolution launch ~4.5 us of its ~17) and summed over the boards, in
// a fixed order, by the kernels that consume them: thread (c = tid % CW, part = tid / CW) takes the boards part,
// part + PARTS, ...; the parts meet in LDS (`sh`: NTH double2) and threads tid < CW return the totals of channel
// c0 + tid.  The first consumer's workgroup 0 files the totals in P.sums for everything later (k_trn_update, the
// backward kernels' bn_coeffs).  Contains a barrier.
// In two halves so that a kernel can request the partials before its bulk input and add them up when it needs them.
// The loads are unconditional (a clamped board index, the surplus multiplied away): behind a branch the compiler waits
// for each load in turn -- 32 round trips.  U = loads in flight per thread (a batch of 128 is one round: 32 x 4 parts at
// 64 channels, 16 x 8 at 32).
template <int CW, int NTH, int U>
__device__ __forceinline__ void sum_partials_request(const float2 *ps, int Cs, int c0, int B, int tid, float2 (&v)[U]) {
    constexpr int PARTS = NTH / CW;
    const int c = tid % CW, part = tid / CW;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int b = part + u * PARTS;
        const float2 x = ps[(size_t)min(b, max(B - 1, 0)) * Cs + c0 + c];      // (B = 0: presummed, nothing is taken)
        const float on = b < B ? 1.f : 0.f;           // (a multiplication, not a select: the compiler turns the select
        v[u] = make_float2(x.x * on, x.y * on);       // back into a branch around the load)
    }
}
template <int CW, int NTH, int U>
__device__ __forceinline__ void sum_partials_finish(const float2 *ps, int Cs, int c0, int B, double2 *sh, int tid, const float2 (&v)[U],
                                                    double &a, double &q) {
    constexpr int PARTS = NTH / CW;
    const int c = tid % CW, part = tid / CW;
    double sa = 0, sq = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        sa += (double)v[u].x;
        sq += (double)v[u].y;
    }
    for (int b0 = part + U * PARTS; b0 < B; b0 += U * PARTS) {       // batches beyond U boards per part
        float2 w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = b0 + u * PARTS;
            const float2 x = ps[(size_t)min(b, B - 1) * Cs + c0 + c];
            const float on = b < B ? 1.f : 0.f;
            w[u] = make_float2(x.x * on, x.y * on);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            sa += (double)w[u].x;
            sq += (double)w[u].y;
        }
    }
    sh[part * CW + c] = make_double2(sa, sq);
    __syncthreads();
    a = 0;
    q = 0;
    if (tid < CW)
        for (int pp = 0; pp < PARTS; ++pp) { a += sh[pp * CW + tid].x; q += sh[pp * CW + tid].y; }
}
template <int CW, int NTH>
__device__ __forceinline__ void sum_partials(const float2 *ps, int Cs, int c0, int B, double2 *sh, int tid, double &a, double &q) {
    float2 v[16];
    sum_partials_request<CW, NTH, 16>(ps, Cs, c0, B, tid, v);
    sum_partials_finish<CW, NTH, 16>(ps, Cs, c0, B, sh, tid, v, a, q);
}
// Every consumer block summing every board's partial pair is what keeps a kernel boundary out of the 6x64 step at the
// reference's batch (64 KB of L2 reads per block at 128 boards) -- and grows with the SQUARE of the batch over a launch
// (2 MB per block at 4 096).  Beyond TRN_PRESUM_BATCH boards a launch of this kernel behind each producer sums a layer's
// pairs once, in a fixed order, into P.sums, and the consumers take the totals from there (P.presum; they pass B = 0
// to their own summation).  grid C / 16, 256 threads = 16 channels x 16 board phases.
__global__ __launch_bounds__(256) void k_trn_totals(const float2 *part, int B, int C, double *dst) {
    __shared__ double2 sh[256];
    const int tid = threadIdx.x, c = blockIdx.x * 16 + (tid & 15), ph = tid >> 4;
    double a = 0, q = 0;
    if (c < C)
        for (int b = ph; b < B; b += 16) {
            const float2 x = part[(size_t)b * C + c];
            a += (double)x.x;
            q += (double)x.y;
        }
    sh[tid] = make_double2(a, q);
    __syncthreads();
    if (tid < 16 && c < C) {
        double sa = 0, sq = 0;
        for (int p = 0; p < 16; ++p) { sa += sh[p * 16 + tid].x; sq += sh[p * 16 + tid].y; }
        dst[(size_t)c * 4] = sa;
        dst[(size_t)c * 4 + 1] = sq;
    }
}
// the presummed totals of channel c (k = 0: raw's sum / sum of squares; 2: the gradient's two sums)
__device__ __forceinline__ void presummed(const double *sums, int l, int C, int c, int k, double &t0, double &t1) {
    const double *sm = sums + ((size_t)l * C + c) * 4 + k;
    t0 = sm[0];
    t1 = sm[1];
}

__device__ __forceinline__ void bn_from_sums(double s0, double s1, float invN, float &mean, float &inv) {
    const double m = s0 * (double)invN, v = s1 * (double)invN - m * m;
    mean = (float)m;
    inv = (float)(1.0 / sqrt((v > 0 ? v : 0) + TRN_EPS));
}

// the power of two that brings `bound` to [2^13, 2^14) (1 for a zero or non-finite bound)
__device__ __forceinline__ float pow2_scale(float bound) {
    if (!(bound > 0.f) || !(bound < INFINITY)) return 1.f;
    int e = (int)((__float_as_uint(bound) >> 23) & 0xff) - 127;           // bound = m 2^e, 1 <= m < 2 (denormals: e = -127)
    e = 13 - e;
    e = e > 60 ? 60 : e < -60 ? -60 : e;      // (two such scales multiply in an epilogue: 2^+-120 stays a float)
    return __uint_as_float((unsigned int)(e + 127) << 23);
}
// |gamma inv| (max|g| + |mean g| + sqrt(n) |mean g xhat|): what a channel's BatchNorm-backward image cannot exceed
__device__ __forceinline__ float draw_bound(float a, float k0, float k1, float gmax, float sqrt_n) {
    return fabsf(a) * (gmax + fabsf(k0) + sqrt_n * fabsf(k1));
}

// =================================================================================================================
// stem forward: embedding (3 -> 4) o conv 3x3 (4 -> C) as a [tap][cell value][cout] table built per block
// =================================================================================================================
template <int C>
__global__ __launch_bounds__(TRN_SMALL_THREADS) void k_trn_stem_fwd(TrnDev P) {
    constexpr int NTH = TRN_SMALL_THREADS;
    __shared__ float T[27 * C];
    __shared__ unsigned char cellv[176];
    __shared__ float red[2][NTH];
    const int b = blockIdx.x, tid = threadIdx.x, N = P.N, cells = P.cells;
    if (C <= 64 && (P.Wf16[1] != nullptr || P.Wb16[1] != nullptr)) {
        // The tower filters as hi / lo f16 fragments of v_mfma_f32_32x32x16_f16, scaled by the layer's power of two.
        // Every block first reduces k_trn_prep's per-block maxima (and the BatchNorm bounds) -- block 0 publishes them
        // for the convolution kernels --, then the grid shares the fragments: an item is one lane's 8 halves of one
        // k-step (8 gathered filter entries in, 16 bytes of hi and 16 of lo out, coalesced):
        //   forward pack        [tap][q = ci >> 4][ntile = co >> 5][hi, lo][lane = (co & 31) + 32 ((ci >> 3) & 1)][ci & 7]
        //   backward-data pack  the same order for the transposed, flipped filter W'[n = ci][k = co][8 - tap]
        constexpr int NTl = (C + 31) / 32, Q16 = C / 16, NB = (C * C * 9 + 255) / 256, NW = NTH / 64;
        __shared__ float sW[TRN_MAXL + 1], sBn[TRN_MAXL + 1];
        const int lane = tid & 63, wave = tid >> 6;
        for (int l = 1 + wave; l <= P.L; l += NW) {
            float m = 0.f, pm[(NB + 63) / 64];
#pragma unroll
            for (int u = 0; u < (NB + 63) / 64; ++u) pm[u] = P.wpmax[l * NB + min(lane + 64 * u, NB - 1)];     // (unconditional: all in flight)
#pragma unroll
            for (int u = 0; u < (NB + 63) / 64; ++u) m = fmaxf(m, pm[u]);
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            if (lane == 0) {
                sW[l] = pow2_scale(m);
                if (b == 0) P.wmax[l] = __float_as_uint(m);
            }
        }
        if (b == 0) {
            const float sq = sqrtf((float)P.B * cells);
            for (int l = wave; l <= P.L; l += NW) {
                float m = lane < C ? fabsf(P.bn_w[l][lane]) * sq + fabsf(P.bn_b[l][lane]) : 0.f;
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
                if (lane == 0) sBn[l] = m;
            }
        }
        __syncthreads();
        if (b == 0 && tid >= 1 && tid <= P.L) {
            // layer l = tid stages act_{l-1}: bounded by its BatchNorm's bound plus, where it has a skip input, the
            // skip's (a chain of them down to layer 0)
            const int l = tid;
            float bd = sBn[l - 1];
            if (((l - 1) & 1) == 0 && l - 1 >= 2)
                for (int j = l - 3; j >= 0; j -= 2) bd += sBn[j];
            const float sa = pow2_scale(bd), sw = sW[l];
            P.fsc[l] = make_float4(sa, 1.f / (sa * sw), sw, 1.f / sw);
        }
        const unsigned per = 9u * Q16 * NTl * 64u, items = per * P.L;
        for (unsigned idx = b * NTH + tid; idx < 2 * items; idx += gridDim.x * NTH) {
            const bool bwd = idx >= items;
            const unsigned it = bwd ? idx - items : idx;
            // (a wave's 64 items are one layer's -- `per` is a multiple of 64 --: the layer's pointers are scalar loads)
            const int l = __builtin_amdgcn_readfirstlane((int)(it / per) + 1);
            unsigned r = it - (unsigned)(l - 1) * per;
            const int ln = (int)(r & 63u); r >>= 6;
            const int nt = (int)(r % NTl); r /= NTl;
            const int q = (int)(r % Q16), tp = (int)(r / Q16);            // tp: the PACK's tap index
            unsigned short *dst = bwd ? P.Wb16[l] : P.Wf16[l];
            if (dst == nullptr) continue;
            const int n = 32 * nt + (ln & 31), k0 = 16 * q + 8 * (ln >> 5), tap = bwd ? 8 - tp : tp;
            const float sc = sW[l];
            const float *w = P.convw[l];
            _Float16 hi[8], lo[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                // forward: n = co, k = ci; backward-data: n = ci, k = co
                const int co = bwd ? k0 + t : n, ci = bwd ? n : k0 + t;
                const float v = (NTl * 32 == C || n < C) ? w[((size_t)min(co, C - 1) * C + min(ci, C - 1)) * 9 + tap] * sc : 0.f;
                hi[t] = (_Float16)v;
                lo[t] = (_Float16)(v - (float)hi[t]);
            }
            uint4 *o = reinterpret_cast<uint4 *>(dst) + ((((size_t)tp * Q16 + q) * NTl + nt) * 2) * 64 + ln;
            o[0] = *reinterpret_cast<const uint4 *>(hi);
            o[64] = *reinterpret_cast<const uint4 *>(lo);
        }
    }
    for (int i = tid; i < 27 * C; i += NTH) T[i] = P.stemT[i];          // built once per step by k_trn_prep
    for (int i = tid; i < cells; i += NTH) cellv[i] = (unsigned char)P.board[(size_t)b * cells + i];
    __syncthreads();
    const int co = tid % C;
    float s1 = 0.f, s2 = 0.f;
    float *out = P.raw[0] + (size_t)b * cells * C;
    for (int pos = tid / C; pos < cells; pos += NTH / C) {
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
        for (int i = tid; i < NTH; i += C) { a += red[0][i]; q += red[1][i]; }
        P.pstat[(size_t)b * C + tid] = make_float2((float)a, (float)q);
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
// ROLE_FWD16: the forward pass on the split-f16 arithmetic of the self-play tower (every fp32 operand as hi = f16(x),
// lo = f16(x - hi); hi hi + hi lo + lo hi accumulated in fp32: 22 significant bits, three v_mfma_f32_32x32x16_f16 per
// 16-channel k-step at 16x the fp32 instruction's rate) -- activations behind a BatchNorm are O(1..100), the range the
// engine's own evaluations live in.  The backward pass stays fp32: gradients span ten orders of magnitude.
// ROLE_BWD16 / k_trn_wgrad16: the gradients the same way.  Their range is managed per layer: every kernel that writes a
// g_l keeps max |g_l| (one atomicMax of the float's bits per block), the consumers turn it into a bound of the
// BatchNorm-backward image they stage -- |draw| <= |gamma inv| (max|g| + |mean g| + sqrt(n) |mean g xhat|), since
// |xhat| <= sqrt(n) for a batch's own statistics -- and scale by the power of two that puts the bound at 2^13..2^14:
// exact, nothing can overflow, and what the split resolves (an absolute 2^-25 of the scaled values) is 2^-38 of the
// bound: far below the fp32 accumulation's own rounding.
enum { ROLE_FWD = 0, ROLE_BWD = 1, ROLE_FWD16 = 2, ROLE_BWD16 = 3 };
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// Diagnostic build only (-DAZX_TRN_STAMP): shader-clock stamps of k_trn_conv's phases (every wave's lane 0), summed per
// role; azx_trn_destroy prints them.  The shipped kernels execute no stamp.
#ifdef AZX_TRN_STAMP
// every wave of the LAST launch of a role leaves its phase times in its own slot (no atomics: a stamp must not wait on
// other waves' stamps); the host averages
#define TS_WAVES 2048
__device__ unsigned long long g_trn_stamp[3][TS_WAVES][8];
#define TS_DECL unsigned long long ts_last = __builtin_amdgcn_s_memtime(), ts_acc[6] = {0, 0, 0, 0, 0, 0}; const unsigned long long ts_rt0 = __builtin_amdgcn_s_memrealtime();
#define TS_MARK(r) { const unsigned long long ts_now = __builtin_amdgcn_s_memtime(); ts_acc[r] += ts_now - ts_last; ts_last = ts_now; }
#define TS_END { const int ts_w = ((blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)); \
    if ((threadIdx.x & 63) == 0 && ts_w < TS_WAVES) { for (int k_ = 0; k_ < 6; ++k_) g_trn_stamp[TS_SLOT][ts_w][k_] = ts_acc[k_]; \
        g_trn_stamp[TS_SLOT][ts_w][6] = 1; g_trn_stamp[TS_SLOT][ts_w][7] = __builtin_amdgcn_s_memrealtime() - ts_rt0; } }
#else
#define TS_DECL
#define TS_MARK(r)
#define TS_END
#endif

// four channels of a position into the split-f16 LDS image: a row is [C hi halves | C lo halves | 16 B pad] (the same
// 4 C + 16 bytes as the fp32 row)
template <int C>
__device__ __forceinline__ void split_store(float *X, int pos, int c, float4 v) {
    const float f[4] = {v.x, v.y, v.z, v.w};
    _Float16 hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { hi[j] = (_Float16)f[j]; lo[j] = (_Float16)(f[j] - (float)hi[j]); }
    unsigned char *row = reinterpret_cast<unsigned char *>(X) + (size_t)pos * ((C + 4) * 4);
    *reinterpret_cast<uint2 *>(row + 2 * c) = *reinterpret_cast<const uint2 *>(hi);
    *reinterpret_cast<uint2 *>(row + 2 * C + 2 * c) = *reinterpret_cast<const uint2 *>(lo);
}

// The layer's tensors as DIRECT kernel arguments (the host indexes the per-layer arrays): fetched out of TrnDev's arrays
// inside the kernel they are a second scalar load that waits for the first (the layer index), ahead of every vector
// load of the prologue -- one round trip of the scalar cache per launch, 24 launches per step.
struct ConvPtrs {
    const float *in0, *in1;          // FWD: raw_{l-1}, act_{l-3} (or in0 again); BWD: g_l, raw_l
    const float *bnw, *bnb;          // the affine pair of the BatchNorm the prologue applies (FWD: l - 1; BWD: l)
    const void *w;                   // the layer's filter fragments for this role
    float *act_out, *out;            // FWD: act_{l-1} (written for the backward pass), raw_l; BWD: -, g_{l-1}
    const float *e_act, *e_raw, *e_skip;   // BWD epilogue: act_{l-1}, raw_{l-1}, g_{l+1} (or e_act again)
};

template <int C, int ROLE>
__device__ __forceinline__ void trn_conv_body(const ConvPtrs &A, const TrnDev &P, const int l, const int nt, const int b, float *lds) {
    constexpr int NT = (C + 31) / 32, LDW = C + 4, Q = C / 8, C4 = C / 4;
    float *X = lds;                                    // [(cells + 1)][LDW]
    const int N = P.N, cells = P.cells;
    // (the backward epilogue re-uses X as a [cells][36] output tile, then as [256][8] running sums: the region is the
    // largest of the three)
    float *cA = X + (size_t)max(max((cells + 1) * LDW, cells * 36), 2048);      // per input channel coefficients
    float *cB = cA + C, *cM = cB + C, *cI = cM + C, *cK = cI + C;     // cK: [2][C] (BWD)
    float *pM = cK + 2 * C, *pI = pM + C;               // BWD epilogue: mean / invstd of layer l - 1
    float *red = pI + C;                                // [4][32][2]
    float *cS = red + 256;                              // the f16 roles: the operand's scale, the epilogue's factor
    double2 *sh = reinterpret_cast<double2 *>(red + 264);    // [256] sum_partials
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    constexpr bool FORWARD = ROLE == ROLE_FWD || ROLE == ROLE_FWD16, F16 = ROLE == ROLE_FWD16 || ROLE == ROLE_BWD16;
    constexpr int TS_SLOT = FORWARD ? 0 : 1;
    (void)TS_SLOT;
    TS_DECL

    // ---- the block's two input tensors are requested first: they travel while the coefficients are computed ----
    // (FWD: raw_{l-1} and the residual act_{l-3}; BWD: g_l and raw_l; every load before the first use -- a dependent
    // load -> use loop would pay the L2 / HBM latency once per iteration, eight times per block)
    constexpr int ITER = (121 * C4 + 255) / 256;
    const size_t base0 = (size_t)b * cells * C;
    const int total = cells * C4;
    const bool has_res = FORWARD && ((l - 1) & 1) == 0 && l - 1 >= 2;
    const float4 *src0 = reinterpret_cast<const float4 *>(A.in0 + base0);
    // (a layer without a skip input reads its own input twice and multiplies the copy away: a null pointer would put
    // the loads behind a branch whose join the compiler waits at)
    const float4 *src1 = reinterpret_cast<const float4 *>(A.in1 + base0);
    const float res_on = has_res ? 1.f : 0.f;
    // (the per-board partial sums this kernel is the first to need go out before them: the coefficients are computed
    // while the bulk is still on its way)
    // Before everything, the few words the coefficient phase works with: the BatchNorm's affine pair, the finished
    // sums of the layers this kernel is not the first to look at, the layer's scales -- requested behind the barrier
    // they would each cost a trip to L2 with nothing to hide it.
    const int cc = min(tid, C - 1);
    const float e_w = A.bnw[cc], e_b = FORWARD ? A.bnb[cc] : 0.f;
    const double2 e_sl = FORWARD ? make_double2(0, 0) : *reinterpret_cast<const double2 *>(P.sums + ((size_t)l * C + cc) * 4);
    const double2 e_sp = FORWARD ? make_double2(0, 0) : *reinterpret_cast<const double2 *>(P.sums + ((size_t)(l - 1) * C + cc) * 4);
    const float4 e_fs = F16 ? P.fsc[l] : make_float4(1.f, 1.f, 1.f, 1.f);
    const float e_gmax = ROLE == ROLE_BWD16 ? __uint_as_float(P.gmax[l]) : 0.f;
    const float2 *psrc = FORWARD ? P.pstat + (size_t)(l - 1) * P.B * C : P.pgsum + (size_t)l * P.B * C;
    constexpr int PSU = C >= 64 ? 32 : 16;
    float2 pv[PSU];
    const int Bsum = P.presum ? 0 : P.B;
    sum_partials_request<C, 256, PSU>(psrc, C, 0, Bsum, tid, pv);
    // (unconditional loads at a clamped index -- the items beyond `total` are never looked at: a load behind a per-lane
    // branch makes the compiler wait at the join)
    float4 v0[ITER], v1[ITER];
#pragma unroll
    for (int k = 0; k < ITER; ++k) v0[k] = src0[min(tid + 256 * k, total - 1)];
#pragma unroll
    for (int k = 0; k < ITER; ++k) v1[k] = src1[min(tid + 256 * k, total - 1)];
    // ---- BWD: what the epilogue reads per output element (the ReLU mask's activation, the BatchNorm input of layer
    // l - 1 and the skip gradient, this block's 32 channels), requested when the k-loop is done.  Requesting them earlier
    // was measured three times and lost three times: before the fp32 k-loop (0.884 vs 0.863 ms per step: they queue
    // ahead of the first taps' filter fragments in the in-order vmcnt); with the split-f16 k-loop, together with the
    // inputs (18.3 vs 17.1 us per launch: 12 more 16-byte loads per thread ahead of the staging's); and behind the last
    // tap's filter fragments, to travel under the last two taps' MFMAs (15.8 vs 15.6 us, 0.496 vs 0.487 ms per step).
    constexpr int LDO = 36, CHo = C < 32 ? C : 32, O4 = CHo / 4, ITERO = (121 * O4 + 255) / 256;
    const int totalo = cells * O4;
    float4 ea[ITERO], er[ITERO], es[ITERO];
    const bool has_skip = ((l - 1) & 1) == 0 && l + 1 <= P.L;
    const float skip_on = has_skip ? 1.f : 0.f;
    auto epi_loads = [&]() {
        const float *pact = A.e_act + base0 + nt * 32, *praw = A.e_raw + base0 + nt * 32;
        const float *skip = A.e_skip + base0 + nt * 32;       // (no skip: act again, a copy that is multiplied away)
#pragma unroll
        for (int k = 0; k < ITERO; ++k) {
            const int i = min(tid + 256 * k, totalo - 1), pos = i / O4, c = (i - pos * O4) * 4;
            const size_t o = (size_t)pos * C + c;
            ea[k] = *reinterpret_cast<const float4 *>(pact + o);
            er[k] = *reinterpret_cast<const float4 *>(praw + o);
        }
#pragma unroll
        for (int k = 0; k < ITERO; ++k) {
            const int i = min(tid + 256 * k, totalo - 1), pos = i / O4, c = (i - pos * O4) * 4;
            es[k] = *reinterpret_cast<const float4 *>(skip + (size_t)pos * C + c);
        }
    };

    // ---- per-channel coefficients -----------------------------------------------------------------------
    // the sums this kernel is the first to need -- FWD: raw_{l-1}'s (sum, sum of squares); BWD: (sum g_l, sum g_l xhat_l)
    // -- are taken from the per-board partials
    double t0, t1;
    TS_MARK(5)
    sum_partials_finish<C, 256, PSU>(psrc, C, 0, Bsum, sh, tid, pv, t0, t1);
    TS_MARK(4)
    if (tid < C) {
        const int c = tid;
        if (P.presum) presummed(P.sums, FORWARD ? l - 1 : l, C, c, FORWARD ? 0 : 2, t0, t1);
        if (nt == 0 && b == 0 && !P.presum) {
            double *dst = P.sums + ((size_t)(FORWARD ? l - 1 : l) * C + c) * 4 + (FORWARD ? 0 : 2);
            dst[0] = t0;
            dst[1] = t1;
        }
        if (FORWARD) {
            float mean, inv;
            bn_from_sums(t0, t1, P.invN, mean, inv);
            const float a = e_w * inv;
            cA[c] = a;
            cB[c] = e_b - mean * a;
        } else {
            float mean, inv;
            bn_from_sums(e_sl.x, e_sl.y, P.invN, mean, inv);
            cM[c] = mean;
            cI[c] = inv;
            cA[c] = e_w * inv;
            cK[c] = (float)(t0 * (double)P.invN);
            cK[C + c] = (float)(t1 * (double)P.invN);
            bn_from_sums(e_sp.x, e_sp.y, P.invN, mean, inv);
            pM[c] = mean;
            pI[c] = inv;
        }
    }
    if (ROLE == ROLE_BWD16 && wave == 0) {
        float bd = tid < C ? draw_bound(cA[tid], cK[tid], cK[C + tid], e_gmax, sqrtf((float)P.B * cells)) : 0.f;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) bd = fmaxf(bd, __shfl_xor(bd, o));
        if (tid == 0) {
            const float sc = pow2_scale(bd);
            cS[0] = sc;
            cS[1] = e_fs.w / sc;           // (powers of two: exact)
        }
    }
    if (ROLE == ROLE_FWD16 && tid == 0) {
        cS[0] = e_fs.x;                    // the activation's and the filter's scales: k_trn_stem_fwd made them
        cS[1] = e_fs.y;
    }
    for (int i = tid; i < LDW; i += 256) X[(size_t)cells * LDW + i] = 0.f;
    __syncthreads();
    TS_MARK(0)

    // ---- stage the input operand ---------------------------------------------------------------------------
    if (FORWARD) {
        float4 *dst = reinterpret_cast<float4 *>(A.act_out + base0);
#pragma unroll
        for (int k = 0; k < ITER; ++k) {
            const int i = tid + 256 * k;
            if (i >= total) break;
            const int pos = i / C4, c = (i - pos * C4) * 4;
            float4 v = v0[k];
            v.x = fmaxf(v.x * cA[c] + cB[c] + v1[k].x * res_on, 0.f);
            v.y = fmaxf(v.y * cA[c + 1] + cB[c + 1] + v1[k].y * res_on, 0.f);
            v.z = fmaxf(v.z * cA[c + 2] + cB[c + 2] + v1[k].z * res_on, 0.f);
            v.w = fmaxf(v.w * cA[c + 3] + cB[c + 3] + v1[k].w * res_on, 0.f);
            if (F16) {
                const float sc = cS[0];
                split_store<C>(X, pos, c, make_float4(v.x * sc, v.y * sc, v.z * sc, v.w * sc));
            } else {
                *reinterpret_cast<float4 *>(X + (size_t)pos * LDW + c) = v;
            }
            if (NT == 1 || (c >> 5) == nt) dst[i] = v;       // each of a board's blocks writes its channel half
        }
    } else {
#pragma unroll
        for (int k = 0; k < ITER; ++k) {
            const int i = tid + 256 * k;
            if (i >= total) break;
            const int pos = i / C4, c = (i - pos * C4) * 4;
            const float4 gv = v0[k], rv = v1[k];
            float4 v;
            v.x = cA[c] * (gv.x - cK[c] - (rv.x - cM[c]) * cI[c] * cK[C + c]);
            v.y = cA[c + 1] * (gv.y - cK[c + 1] - (rv.y - cM[c + 1]) * cI[c + 1] * cK[C + c + 1]);
            v.z = cA[c + 2] * (gv.z - cK[c + 2] - (rv.z - cM[c + 2]) * cI[c + 2] * cK[C + c + 2]);
            v.w = cA[c + 3] * (gv.w - cK[c + 3] - (rv.w - cM[c + 3]) * cI[c + 3] * cK[C + c + 3]);
            if (F16) {
                const float sc = cS[0];
                split_store<C>(X, pos, c, make_float4(v.x * sc, v.y * sc, v.z * sc, v.w * sc));
            } else {
                *reinterpret_cast<float4 *>(X + (size_t)pos * LDW + c) = v;
            }
        }
    }
    __syncthreads();
    TS_MARK(1)


    // ---- k-loop: 9 taps x C / 8 steps of four 32x32x2 MFMAs ----------------------------------------------------
    // One wave per SIMD and a dependent MFMA chain: nothing hides a load but the loop itself.  So the filter
    // fragments of tap t + 1 (C / 8 16-byte loads from L2) are requested before tap t's MFMAs start, and a tap's
    // activation fragments (LDS) are all requested at its top; two accumulators alternate so that consecutive
    // MFMAs do not wait on each other's result.
    const int r = wave * 32 + li;
    const bool rvalid = r < cells;
    const int ry = r / N, rx = r - ry * N;
    f32x16 acc, acc2;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc2[i] = 0.f; }
    if (F16) {
        // 9 taps x C / 16 k-steps x 3 MFMAs (hi hi, hi lo, lo hi) of 32 cycles; fragments: A = the positions' 8
        // consecutive channels 16 q + 8 (lane >> 5) .. (one 16-byte LDS read per plane; 272-byte rows: conflict-free),
        // B = the filter's, packed by k_trn_stem_fwd as [tap][q][ntile][hi, lo][lane][8]; a tap's filter fragments are
        // requested a tap ahead.  With the LDS reads pinned ahead of the MFMAs (below) what bounds the k-loop is
        // operand delivery: each of the block's four waves pulls the same 72 KB of filter fragments through the CU's
        // one 64 B/clk vector-memory path (4.6 K cycles against 3.5 K of MFMA).  Measured and dropped: the filter
        // fragments requested two or three taps ahead (with the pinned schedule: 0.479-0.484 vs 0.480 ms per step;
        // before it the wait only moved from the k-loop into the staging) and the fragments staged ONCE per block
        // through LDS (k-loop 5.8 K, LDS-bound -- A and B fragments are 170 B/clk of reads per CU against 128 --, same
        // launch time, and at 116 KB of LDS the block no longer shares a CU with the filter-gradient kernel beside it:
        // 0.590 vs 0.559 ms per step).
        constexpr int Q16 = C / 16, ROWB = LDW * 4;
        const uint4 *w16 = reinterpret_cast<const uint4 *>(A.w) + (size_t)nt * 128 + lane;
        uint4 wc[Q16][2], wn[Q16][2];
#pragma unroll
        for (int q = 0; q < Q16; ++q) { wc[q][0] = w16[(size_t)q * NT * 128]; wc[q][1] = w16[(size_t)q * NT * 128 + 64]; }
        const unsigned char *Xb = reinterpret_cast<const unsigned char *>(X);
        // The activation fragments of tap t + 1 (8 LDS reads) are requested BEFORE tap t's MFMAs, like the filter
        // fragments: left to itself the compiler sinks every LDS read to just ahead of the MFMA that uses it and the
        // k-loop runs at one LDS latency per k-step (8.4 K cycles for 3.5 K of MFMA).  The scheduling barriers are what
        // keeps the requests where they are written.
        uint4 ah[Q16], al[Q16], nh[Q16], nl[Q16];
        auto areq = [&](int tap, uint4 (&h)[Q16], uint4 (&lo_)[Q16]) {
            const int yy = ry + tap / 3 - 1, xx = rx + tap % 3 - 1;
            const bool ok = rvalid && yy >= 0 && yy < N && xx >= 0 && xx < N;
            const unsigned char *arow = Xb + (size_t)(ok ? yy * N + xx : cells) * ROWB + 16 * lh;
#pragma unroll
            for (int q = 0; q < Q16; ++q) {
                h[q] = *reinterpret_cast<const uint4 *>(arow + 32 * q);
                lo_[q] = *reinterpret_cast<const uint4 *>(arow + 2 * C + 32 * q);
            }
        };
        areq(0, ah, al);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap < 8) {
                const uint4 *wt = w16 + (size_t)(tap + 1) * Q16 * NT * 128;
#pragma unroll
                for (int q = 0; q < Q16; ++q) { wn[q][0] = wt[(size_t)q * NT * 128]; wn[q][1] = wt[(size_t)q * NT * 128 + 64]; }
                areq(tap + 1, nh, nl);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < Q16; ++q) {
                const f16x8 xh = *reinterpret_cast<const f16x8 *>(&ah[q]), xl = *reinterpret_cast<const f16x8 *>(&al[q]);
                const f16x8 wh = *reinterpret_cast<const f16x8 *>(&wc[q][0]), wlo = *reinterpret_cast<const f16x8 *>(&wc[q][1]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wh, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wlo, acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, wh, acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (tap < 8) {
#pragma unroll
                for (int q = 0; q < Q16; ++q) { wc[q][0] = wn[q][0]; wc[q][1] = wn[q][1]; ah[q] = nh[q]; al[q] = nl[q]; }
            }
        }
    } else {
    const float4 *wl = reinterpret_cast<const float4 *>(A.w) + (size_t)nt * 64 + lane;
    float4 bcur[Q], bnxt[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) bcur[q] = wl[(size_t)q * NT * 64];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        if (tap < 8) {
            const float4 *wt = wl + (size_t)(tap + 1) * Q * NT * 64;
#pragma unroll
            for (int q = 0; q < Q; ++q) bnxt[q] = wt[(size_t)q * NT * 64];
        }
        const int yy = ry + tap / 3 - 1, xx = rx + tap % 3 - 1;
        const bool ok = rvalid && yy >= 0 && yy < N && xx >= 0 && xx < N;
        const float *arow = X + (size_t)(ok ? yy * N + xx : cells) * LDW + 4 * lh;
        float4 af[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) af[q] = *reinterpret_cast<const float4 *>(arow + 8 * q);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].x, bcur[q].x, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].y, bcur[q].y, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].z, bcur[q].z, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].w, bcur[q].w, acc2, 0, 0, 0);
        }
        if (tap < 8) {
#pragma unroll
            for (int q = 0; q < Q; ++q) bcur[q] = bnxt[q];
        }
    }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] += acc2[i];
    TS_MARK(2)

    // ---- epilogue: C/D layout col = lane & 31 (output channel), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) ----
    const int co = nt * 32 + li;
    const bool cvalid = co < C;
    float s1 = 0.f, s2 = 0.f;
    const size_t base = base0;
    if (FORWARD) {
        float *out = A.out + base;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
            if (row < cells && cvalid) {
                const float v = F16 ? acc[i] * cS[1] : acc[i];
                out[(size_t)row * C + co] = v;
                s1 += v;
                s2 += v * v;
            }
        }
    } else {
        // the accumulators go through LDS (the input tile is done with) so that the skip gradient, the ReLU mask's
        // activation and the BatchNorm input are read -- and g written -- in coalesced 16-byte pieces
        epi_loads();
        __syncthreads();                                 // every wave has finished reading X
        float *Y = X;                                    // [cells][LDO]: this block's 32 output channels
        const float unscale = F16 ? cS[1] : 1.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
            if (row < cells) Y[(size_t)row * LDO + li] = F16 ? acc[i] * unscale : acc[i];
        }
        __syncthreads();
        float *out = A.out + base + nt * 32;
        // a thread's items all have the same four channels (256 is a multiple of O4): four pairs of running sums
        float a4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f}, vmax = 0.f;
        const int c0 = (tid % O4) * 4;
#pragma unroll
        for (int k = 0; k < ITERO; ++k) {
            const int i = tid + 256 * k;
            if (i >= totalo) break;
            const int pos = i / O4;
            const float4 y = *reinterpret_cast<const float4 *>(Y + (size_t)pos * LDO + c0);
            float4 v;
            v.x = ea[k].x > 0.f ? y.x + es[k].x * skip_on : 0.f;
            v.y = ea[k].y > 0.f ? y.y + es[k].y * skip_on : 0.f;
            v.z = ea[k].z > 0.f ? y.z + es[k].z * skip_on : 0.f;
            v.w = ea[k].w > 0.f ? y.w + es[k].w * skip_on : 0.f;
            *reinterpret_cast<float4 *>(out + (size_t)pos * C + c0) = v;
            vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            a4[0] += v.x; a4[1] += v.y; a4[2] += v.z; a4[3] += v.w;
            q4[0] += v.x * (er[k].x - pM[nt * 32 + c0]) * pI[nt * 32 + c0];
            q4[1] += v.y * (er[k].y - pM[nt * 32 + c0 + 1]) * pI[nt * 32 + c0 + 1];
            q4[2] += v.z * (er[k].z - pM[nt * 32 + c0 + 2]) * pI[nt * 32 + c0 + 2];
            q4[3] += v.w * (er[k].w - pM[nt * 32 + c0 + 3]) * pI[nt * 32 + c0 + 3];
        }
        // per channel: the 256 / O4 threads that share it -> this board's partial pair (k_trn_conv of layer l - 1 and
        // its filter-gradient kernel add the boards up)
        __syncthreads();
        float *rs = Y;                                   // [256][8]
#pragma unroll
        for (int j = 0; j < 4; ++j) { rs[tid * 8 + j] = a4[j]; rs[tid * 8 + 4 + j] = q4[j]; }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        if (lane == 0) red[wave] = vmax;
        __syncthreads();
        // (one atomic per block on ONE word: 256 of them cost ~2 us at the end of a 13 us launch -- but spread over 32 slot
        // words, as the wide step does, the consumers' slot reads cost more: 0.512 vs 0.503 ms a step, round 6)
        if (tid == 64)           // (a NaN's bits are above every finite float's: the consumers then take scale 1)
            atomicMax(&P.gmax[l - 1], __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
        if (tid < CHo) {
            const int grp4 = tid / 4, j = tid % 4;
            double a = 0, q = 0;
            for (int th = grp4; th < 256; th += O4) { a += rs[th * 8 + j]; q += rs[th * 8 + 4 + j]; }
            P.pgsum[((size_t)(l - 1) * P.B + b) * C + nt * 32 + tid] = make_float2((float)a, (float)q);
        }
        TS_MARK(3)
        TS_END
        return;
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
        P.pstat[((size_t)l * P.B + b) * C + nt * 32 + tid] = make_float2((float)a, (float)q);
    }
    TS_MARK(3)
    TS_END
}

// =================================================================================================================
// weight gradient of layer l: dW[co][ci][tap] = sum over boards and positions of draw_l[pos][co] act_{l-1}[pos + tap][ci]
//   A GEMM with a tiny output (C x 9 C) and a long reduction (B x cells positions).  Splitting the reduction over
//   P_k blocks costs P_k partial copies of the output; splitting the output over P_o blocks costs P_o reads of both
//   operands.  Here P_o = the (co tile, ci tile) pairs (4 for C = 64) and P_k = G board groups (64 at batch 128):
//   grid (pairs, G) = 256 workgroups.  A block stages, per board, the 32-channel halves of draw (BatchNorm backward
//   applied on the way in) and of the input ONCE and runs all 9 taps from them -- a tap is a row offset into the
//   staged input, the off-board taps read a zero row; its 4 waves split the positions and are summed through LDS.
//   Partials are reduced by k_trn_update in a fixed order: the step is reproducible.
// =================================================================================================================
struct WgradPtrs { const float *g, *raw, *act, *bnw; };        // g_l, raw_l, act_{l-1}, BatchNorm l's gamma (see ConvPtrs)

template <int C>
__device__ __forceinline__ void trn_wgrad_body(const WgradPtrs &W_, const TrnDev &P, const int l, const int G, const int pair, const int grp, float *lds) {
    constexpr int NT = (C + 31) / 32, CH = C < 32 ? C : 32, H4 = CH / 4, ITER = (121 * H4 + 255) / 256;
    const int N = P.N, cells = P.cells, KP = (cells + 1) & ~1, NP = N + 2;
    float *D = lds;                                     // [KP][32] draw, this block's co half (rows >= cells zero)
    float *A = D + (size_t)KP * 32;                      // [(N + 2)^2][32] input, this block's ci half, on a board with a
                                                         //   zero border: a tap is a constant row offset, no bounds test
    float *red = A + (size_t)NP * NP * 32;               // [3 taps][4 waves][1024] reduction rounds
    float *cA = red + 12 * 1024, *cM = cA + 32, *cI = cM + 32, *cK = cI + 32;     // cK [2][32]
    int *prow = reinterpret_cast<int *>(cK + 64);        // [KP] padded-board row of position k (0 for the padding k)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tm = pair / NT, tn = pair % NT;
    const int li = lane & 31, lh = lane >> 5;
    constexpr int TS_SLOT = 2;
    (void)TS_SLOT;
    TS_DECL
    // a board's three tensors (this block's channel halves) as 16-byte pieces, requested a board ahead: the first
    // board's before the LDS is initialised, the next one's before the current one's k-loop
    const int total = cells * H4;
    float4 vg[ITER], vr[ITER], va[ITER];
    auto request = [&](int b) {
        const size_t base = (size_t)b * cells * C;
        const float *gs = W_.g + base + tm * 32, *rs = W_.raw + base + tm * 32, *as = W_.act + base + tn * 32;
#pragma unroll
        for (int k = 0; k < ITER; ++k) {       // (clamped, unconditional: the staging skips the items beyond `total`)
            const int i = min(tid + 256 * k, total - 1), pos = i / H4, c = (i - pos * H4) * 4;
            vg[k] = *reinterpret_cast<const float4 *>(gs + (size_t)pos * C + c);
            vr[k] = *reinterpret_cast<const float4 *>(rs + (size_t)pos * C + c);
            va[k] = *reinterpret_cast<const float4 *>(as + (size_t)pos * C + c);
        }
    };
    if (grp < P.B) request(grp);
    double t0, t1;       // (sum g_l, sum g_l xhat_l) of this block's channels (k_trn_conv<BWD> of layer l runs beside this kernel)
    sum_partials<CH, 256>(P.pgsum + (size_t)l * P.B * C, C, tm * 32, P.presum ? 0 : P.B, reinterpret_cast<double2 *>(red), tid, t0, t1);
    if (tid < CH) {
        const int c = tm * 32 + tid;
        if (P.presum) presummed(P.sums, l, C, c, 2, t0, t1);
        float mean, inv;
        bn_coeffs(P, l, c, mean, inv);
        cM[tid] = mean;
        cI[tid] = inv;
        cA[tid] = W_.bnw[c] * inv;
        cK[tid] = (float)(t0 * (double)P.invN);
        cK[32 + tid] = (float)(t1 * (double)P.invN);
    }
    {   // the operand tiles start zero: the padding k-row of D and the border of the padded board stay that way
        float4 *z = reinterpret_cast<float4 *>(lds);
        for (int i = tid; i < (KP * 32 + NP * NP * 32) / 4; i += 256) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int i = tid; i < KP; i += 256) prow[i] = i < cells ? (i / N + 1) * NP + (i % N) + 1 : 0;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    __syncthreads();
    TS_MARK(0)
    const int S = KP / 2;
    for (int b = grp; b < P.B; b += G) {
#pragma unroll
        for (int k = 0; k < ITER; ++k) {
            const int i = tid + 256 * k;
            if (i >= total) break;
            const int pos = i / H4, c = (i - pos * H4) * 4;
            const float4 gv = vg[k], rv = vr[k];
            float4 v;
            v.x = cA[c] * (gv.x - cK[c] - (rv.x - cM[c]) * cI[c] * cK[32 + c]);
            v.y = cA[c + 1] * (gv.y - cK[c + 1] - (rv.y - cM[c + 1]) * cI[c + 1] * cK[32 + c + 1]);
            v.z = cA[c + 2] * (gv.z - cK[c + 2] - (rv.z - cM[c + 2]) * cI[c + 2] * cK[32 + c + 2]);
            v.w = cA[c + 3] * (gv.w - cK[c + 3] - (rv.w - cM[c + 3]) * cI[c + 3] * cK[32 + c + 3]);
            *reinterpret_cast<float4 *>(D + (size_t)pos * 32 + c) = v;
            *reinterpret_cast<float4 *>(A + (size_t)prow[pos] * 32 + c) = va[k];
        }
        __syncthreads();
        if (b + G < P.B) request(b + G);                 // travels under this board's k-loop
        TS_MARK(1)
        // this wave's k-steps s = wave, wave + 4, ...; the operands of step s + 4 are requested before step s's MFMAs
        // (a 2x2 board has two k-steps: waves 2 and 3 have none -- their first operands are read at a clamped step
        // and never used)
        int s = wave;
        const int s0 = min(s, S - 1);
        float a_c = D[(size_t)(2 * s0 + lh) * 32 + li], b_c[9];
        {
            const float *bp = A + (size_t)prow[2 * s0 + lh] * 32 + li;
#pragma unroll
            for (int t = 0; t < 9; ++t) b_c[t] = bp[((t / 3 - 1) * NP + (t % 3 - 1)) * 32];
        }
        for (; s < S; s += 4) {
            const int sn = s + 4 < S ? s + 4 : s;
            const float a_n = D[(size_t)(2 * sn + lh) * 32 + li];
            float b_n[9];
            const float *bp = A + (size_t)prow[2 * sn + lh] * 32 + li;
#pragma unroll
            for (int t = 0; t < 9; ++t) b_n[t] = bp[((t / 3 - 1) * NP + (t % 3 - 1)) * 32];
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c, b_c[t], acc[t], 0, 0, 0);
            a_c = a_n;
#pragma unroll
            for (int t = 0; t < 9; ++t) b_c[t] = b_n[t];
        }
        __syncthreads();
        TS_MARK(2)
    }
    // The four waves' shares of the nine tiles -> one, three taps per round through LDS (plain 16-byte writes: LDS float
    // atomics run at a small fraction of that rate).  red[(u * 4 + wave) * 4 + (i >> 2)][lane][i & 3]; the partial copy
    // is laid out [tap][co][ci] so that the rows leave in 128-byte pieces (k_trn_update reads it transposed).
    float *part = P.wpart + ((size_t)(l - 1) * G + grp) * ((size_t)C * C * 9);
#pragma unroll
    for (int t0 = 0; t0 < 9; t0 += 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4 *>(red + ((size_t)((u * 4 + wave) * 4 + q) * 64 + lane) * 4) =
                    make_float4(acc[t0 + u][4 * q], acc[t0 + u][4 * q + 1], acc[t0 + u][4 * q + 2], acc[t0 + u][4 * q + 3]);
        __syncthreads();
        for (int e = tid; e < 3 * 1024; e += 256) {
            const int u = e >> 10, col = (e >> 5) & 31, cil = e & 31;          // output (co = col, ci = cil) of tile u
            const int q = col >> 3, lh_ = (col >> 2) & 1, r = col & 3;
            const size_t o = ((size_t)q * 64 + cil + 32 * lh_) * 4 + r;
            const float *w0 = red + (size_t)(u * 4) * 1024;
            const int co = tm * 32 + col, ci = tn * 32 + cil;
            if (co < C && ci < C)
                part[((size_t)(t0 + u) * C + co) * C + ci] = (w0[o] + w0[1024 + o]) + (w0[2048 + o] + w0[3072 + o]);
        }
        __syncthreads();
    }
    TS_MARK(3)
    TS_END
}


// =================================================================================================================
// the same filter gradient on the split-f16 arithmetic.  The reduction index is the position, so both operands are
// wanted K-major while they arrive (and are staged) position-major: `ds_read_b64_tr_b16` reads a 4-row x 16-column
// block of 16-bit elements and hands every lane its COLUMN -- two of them are a lane's 8 consecutive k of
// v_mfma_f32_32x32x16_f16.  k enumerates a board 16 cells wide (k = 16 y + x; the cells x >= N are zero rows of
// draw), so a k-step is a board row, the input image -- the same 16-wide board with a zero border, cell (y, x) in row
// 16 (y + 1) + x + 1 -- is read at row k + 17 + 16 dy + dx for tap (dy, dx): a constant offset, 8-byte aligned
// whatever the tap, and with 64-byte rows (32 channels) the 4 rows a 32-lane half reads are one contiguous 256 bytes:
// conflict-free.  draw is scaled by the layer's power of two (see ROLE_BWD16), the partial sums are scaled back.
// =================================================================================================================
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f16x8 tr_frag(const unsigned char *p) {
    typedef s16x4 __attribute__((address_space(3))) *lds_ptr;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 256));       // four rows on
    s16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3]; r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return *reinterpret_cast<const f16x8 *>(&r);
}

template <int C, int MAXCELLS>
__device__ __forceinline__ void trn_wgrad16_body(const WgradPtrs &W_, const TrnDev &P, const int l, const int G, const int pair, const int grp, float *lds) {
    constexpr int NT = (C + 31) / 32, CH = C < 32 ? C : 32, H4 = CH / 4, ITER = (MAXCELLS * H4 + 255) / 256;
    const int N = P.N, cells = P.cells, KR = N * 16, BR = (N + 3) * 16;
    unsigned char *Dh = reinterpret_cast<unsigned char *>(lds), *Dl = Dh + (size_t)KR * 64;     // [KR][32] f16 draw hi / lo
    unsigned char *Bh = Dl + (size_t)KR * 64, *Bl = Bh + (size_t)BR * 64;                       // [BR][32] f16 input hi / lo
    float *red = reinterpret_cast<float *>(Bl + (size_t)BR * 64);     // [256] double2: sum_partials
    float *cA = red + 1024, *cM = cA + 32, *cI = cM + 32, *cK = cI + 32, *cS = cK + 64;    // cK [2][32], cS [2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tm = pair / NT, tn = pair % NT;
    constexpr int TS_SLOT = 2;
    (void)TS_SLOT;
    TS_DECL
    const int total = cells * H4;
    float4 vg[ITER], vr[ITER], va[ITER];
    auto request = [&](int b) {
        const size_t base = (size_t)b * cells * C;
        const float *gs = W_.g + base + tm * 32, *rs = W_.raw + base + tm * 32, *as = W_.act + base + tn * 32;
#pragma unroll
        for (int k = 0; k < ITER; ++k) {       // (clamped, unconditional: the staging skips the items beyond `total`)
            const int i = min(tid + 256 * k, total - 1), pos = i / H4, c = (i - pos * H4) * 4;
            vg[k] = *reinterpret_cast<const float4 *>(gs + (size_t)pos * C + c);
            vr[k] = *reinterpret_cast<const float4 *>(rs + (size_t)pos * C + c);
            va[k] = *reinterpret_cast<const float4 *>(as + (size_t)pos * C + c);
        }
    };
    // (sum g_l, sum g_l xhat_l) of this block's channels, from the per-board partials (k_trn_conv<BWD> of layer l runs
    // beside this kernel): requested first, added up while the first board's tensors travel
    // (and before those, the few words the coefficient phase needs -- see trn_conv_body)
    const int cc = tm * 32 + min(tid, CH - 1);
    const float e_w = W_.bnw[cc], e_gmax = __uint_as_float(P.gmax[l]);
    const double2 e_sl = *reinterpret_cast<const double2 *>(P.sums + ((size_t)l * C + cc) * 4);
    float2 pv[16];
    sum_partials_request<CH, 256, 16>(P.pgsum + (size_t)l * P.B * C, C, tm * 32, P.presum ? 0 : P.B, tid, pv);
    request(grp);
    double t0, t1;
    sum_partials_finish<CH, 256, 16>(P.pgsum + (size_t)l * P.B * C, C, tm * 32, P.presum ? 0 : P.B, reinterpret_cast<double2 *>(red), tid, pv, t0, t1);
    if (wave == 0) {
        float bd = 0.f;
        if (tid < CH) {
            if (P.presum) presummed(P.sums, l, C, tm * 32 + tid, 2, t0, t1);
            float mean, inv;
            bn_from_sums(e_sl.x, e_sl.y, P.invN, mean, inv);
            cM[tid] = mean;
            cI[tid] = inv;
            const float a = e_w * inv;
            const float k0 = (float)(t0 * (double)P.invN), k1 = (float)(t1 * (double)P.invN);
            cA[tid] = a;
            cK[tid] = k0;
            cK[32 + tid] = k1;
            bd = draw_bound(a, k0, k1, e_gmax, sqrtf((float)P.B * cells));
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) bd = fmaxf(bd, __shfl_xor(bd, o));
        if (tid == 0) {
            const float sc = pow2_scale(bd);
            cS[0] = sc;
            cS[1] = 1.f / sc;
        }
    }
    {   // the operand images start zero: the rows x >= N of draw and the border of the input stay that way
        float4 *z = reinterpret_cast<float4 *>(lds);
        for (int i = tid; i < (KR + BR) * 128 / 16; i += 256) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // The nine taps are shared out by COLUMN: wave w < 3 owns dx = w - 1 and its three dy, every owner walks all the
    // board rows; the fourth wave only helps with the staging.  A tap's tile is then complete in one wave's accumulators
    // and leaves for HBM straight from them (C/D layout: a register's lanes 0..31 are 32 consecutive ci of one co:
    // 128-byte rows) -- no reduction over the waves -- and, the point of the column: the input fragment of (row s, dy)
    // is the fragment of (row s + dy, 0), so a k-step needs ONE new input fragment (row s + 2, requested two rows
    // ahead) and rotates three.  That is 8 transposed reads per k-step (4 draw + 4 input: hi, lo x two halves of k)
    // instead of 16: a wave can have 15 LDS operations outstanding (lgkmcnt is four bits), and with 32 in a k-step the
    // loop ran at two LDS latencies per step -- 565 cycles against 288 of MFMA.
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const bool owner = wv < 3;
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    __syncthreads();
    TS_MARK(0)
    const float sc = cS[0];
    // this lane's corner of the transposed reads: row 8 (lane >> 5) + ((lane & 15) >> 2) of the k-step, channels
    // 16 ((lane >> 4) & 1) + 4 (lane & 3) ..
    const int frag_off = (8 * (lane >> 5) + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    // input fragment of board row r (r = -1 .. N: the zero border included) in this wave's column
    const int col_off = (17 + (owner ? wv - 1 : 0)) * 64 + frag_off;
    for (int b = grp; b < P.B; b += G) {
#pragma unroll
        for (int k = 0; k < ITER; ++k) {
            const int i = tid + 256 * k;
            if (i >= total) break;
            const int pos = i / H4, c = (i - pos * H4) * 4;
            const int y = pos / N, krow = y * 16 + (pos - y * N);
            const float4 gv = vg[k], rv = vr[k];
            float f[4], a4[4] = {va[k].x, va[k].y, va[k].z, va[k].w};
            f[0] = sc * (cA[c] * (gv.x - cK[c] - (rv.x - cM[c]) * cI[c] * cK[32 + c]));
            f[1] = sc * (cA[c + 1] * (gv.y - cK[c + 1] - (rv.y - cM[c + 1]) * cI[c + 1] * cK[32 + c + 1]));
            f[2] = sc * (cA[c + 2] * (gv.z - cK[c + 2] - (rv.z - cM[c + 2]) * cI[c + 2] * cK[32 + c + 2]));
            f[3] = sc * (cA[c + 3] * (gv.w - cK[c + 3] - (rv.w - cM[c + 3]) * cI[c + 3] * cK[32 + c + 3]));
            _Float16 dh[4], dl[4], ah[4], al[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dh[j] = (_Float16)f[j]; dl[j] = (_Float16)(f[j] - (float)dh[j]);
                ah[j] = (_Float16)a4[j]; al[j] = (_Float16)(a4[j] - (float)ah[j]);
            }
            *reinterpret_cast<uint2 *>(Dh + (size_t)krow * 64 + 2 * c) = *reinterpret_cast<const uint2 *>(dh);
            *reinterpret_cast<uint2 *>(Dl + (size_t)krow * 64 + 2 * c) = *reinterpret_cast<const uint2 *>(dl);
            *reinterpret_cast<uint2 *>(Bh + (size_t)(krow + 17) * 64 + 2 * c) = *reinterpret_cast<const uint2 *>(ah);
            *reinterpret_cast<uint2 *>(Bl + (size_t)(krow + 17) * 64 + 2 * c) = *reinterpret_cast<const uint2 *>(al);
        }
        __syncthreads();
        if (b + G < P.B) request(b + G);                 // travels under this board's k-loop
        TS_MARK(1)
        if (owner) {
            // board row s = one k-step (every lane takes part in a transposed read: the branch and the loop bounds are
            // uniform).  in0 / in1 / in2 = the input fragments of rows s - 1, s, s + 1
            f16x8 dhi = tr_frag(Dh + frag_off), dlo = tr_frag(Dl + frag_off);
            f16x8 h0 = tr_frag(Bh + col_off - 16 * 64), l0 = tr_frag(Bl + col_off - 16 * 64);
            f16x8 h1 = tr_frag(Bh + col_off), l1 = tr_frag(Bl + col_off);
            f16x8 h2 = tr_frag(Bh + col_off + 16 * 64), l2 = tr_frag(Bl + col_off + 16 * 64);
            for (int s = 0; s < N; ++s) {
                const int sn = s + 1 < N ? s + 1 : s;
                const f16x8 ndhi = tr_frag(Dh + frag_off + (size_t)sn * 16 * 64), ndlo = tr_frag(Dl + frag_off + (size_t)sn * 16 * 64);
                // row s + 2 (for the last k-step a row inside the image that is never used)
                const size_t r3 = (size_t)(s + 2 <= N ? s + 2 : N) * 16 * 64;
                const f16x8 h3 = tr_frag(Bh + col_off + r3), l3 = tr_frag(Bl + col_off + r3);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dhi, h0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dhi, h1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dhi, h2, acc[2], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dhi, l0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dhi, l1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dhi, l2, acc[2], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dlo, h0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dlo, h1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dlo, h2, acc[2], 0, 0, 0);
                dhi = ndhi; dlo = ndlo;
                h0 = h1; l0 = l1; h1 = h2; l1 = l2; h2 = h3; l2 = l3;
            }
        }
        __syncthreads();
        TS_MARK(2)
    }
    // the scale leaves here: tap (dy = u - 1, dx = wave - 1) is tap index 3 u + wave
    const float unscale = cS[1];
    float *part = P.wpart + ((size_t)(l - 1) * G + grp) * ((size_t)C * C * 9);
    const int li = lane & 31, lh = lane >> 5, ci = tn * 32 + li;
    if (owner) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int t = 3 * u + wv;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = tm * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
                if (co < C && ci < C) part[((size_t)t * C + co) * C + ci] = acc[u][i] * unscale;
            }
        }
    }
    TS_MARK(3)
    TS_END
}

template <int C, int ROLE>
__global__ __launch_bounds__(256) void k_trn_conv(ConvPtrs Q, int l, TrnDev P) {
    extern __shared__ __align__(16) float lds[];
    trn_conv_body<C, ROLE>(Q, P, l, blockIdx.x, blockIdx.y, lds);
}

template <int C>
__global__ __launch_bounds__(256) void k_trn_wgrad(WgradPtrs A, int l, int G, TrnDev P) {
    extern __shared__ __align__(16) float lds[];
    trn_wgrad_body<C>(A, P, l, G, blockIdx.x, blockIdx.y, lds);
}

// (MAXCELLS: the largest board a thread's staging registers are sized for -- 121 cells, 169 for the wide towers)
template <int C, int MAXCELLS = (C > 64 ? 169 : 121)>
__global__ __launch_bounds__(256) void k_trn_wgrad16(WgradPtrs A, int l, int G, TrnDev P) {
    extern __shared__ __align__(16) float lds[];
    trn_wgrad16_body<C, MAXCELLS>(A, P, l, G, blockIdx.x, blockIdx.y, lds);
}

// (Measured and dropped: one launch per backward layer with both consumers of g_l -- k_trn_conv<BWD>'s workgroups and
// k_trn_wgrad's -- in one grid, two resident per CU at 256 registers: 1.00-1.01 ms per step against 0.83 for the two
// kernels on two streams, whichever way the roles were mapped to workgroup indices.  As separate kernels the data chain
// runs ahead into layer l - 1 while layer l's filter gradient is still in flight; a joint launch ends with its slower
// half and nothing of the next layer can start under it.)

// =================================================================================================================
// heads, forward part 1: act_L = relu(BN_L(raw_L) + act_{L-2}); the two 1x1 convolutions; their batch sums
// =================================================================================================================
template <int C>
__global__ __launch_bounds__(TRN_MID_THREADS) void k_trn_heads_conv(TrnDev P) {
    constexpr int C4 = C / 4, LDX = C + 1, NTH = TRN_MID_THREADS, ITER = (121 * C4 + NTH - 1) / NTH;
    extern __shared__ __align__(16) float lds[];
    float *X = lds;                         // [cells][LDX]
    float *cA = X + (size_t)P.cells * LDX, *cB = cA + C;
    float *W = cB + C;                      // [6][C]
    float *red = W + 6 * C;                 // [6][2]
    const int b = blockIdx.x, tid = threadIdx.x, cells = P.cells, L = P.L;
    double t0, t1;
    sum_partials<C, NTH>(P.pstat + (size_t)L * P.B * C, C, 0, P.presum ? 0 : P.B,
                         reinterpret_cast<double2 *>(lds + ((P.cells * LDX + 8 * C + 16 + 3) & ~3)), tid, t0, t1);
    if (tid < C) {
        float mean, inv;
        if (P.presum) presummed(P.sums, L, C, tid, 0, t0, t1);
        if (b == 0 && !P.presum) {
            P.sums[((size_t)L * C + tid) * 4] = t0;
            P.sums[((size_t)L * C + tid) * 4 + 1] = t1;
        }
        bn_from_sums(t0, t1, P.invN, mean, inv);
        const float a = P.bnwL[tid] * inv;
        cA[tid] = a;
        cB[tid] = P.bnbL[tid] - mean * a;
    }
    for (int i = tid; i < 6 * C; i += NTH) W[i] = i < 2 * C ? P.vconv[i] : P.pconv[i - 2 * C];
    if (tid < 12) red[tid] = 0.f;
    const size_t base = (size_t)b * cells * C;
    const float4 *src = reinterpret_cast<const float4 *>(P.rawL + base);
    const bool has_res = L >= 2;
    const float4 *res = has_res ? reinterpret_cast<const float4 *>(P.actLm2 + base) : nullptr;
    float4 *dst = reinterpret_cast<float4 *>(P.actL + base);
    const int total = cells * C4;
    float4 va[ITER], vr[ITER];
#pragma unroll
    for (int k = 0; k < ITER; ++k) {
        const int i = tid + NTH * k;
        va[k] = i < total ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        vr[k] = (has_res && i < total) ? res[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITER; ++k) {
        const int i = tid + NTH * k;
        if (i >= total) break;
        const int pos = i / C4, c = (i - pos * C4) * 4;
        float4 v = va[k];
        v.x = fmaxf(v.x * cA[c] + cB[c] + vr[k].x, 0.f);
        v.y = fmaxf(v.y * cA[c + 1] + cB[c + 1] + vr[k].y, 0.f);
        v.z = fmaxf(v.z * cA[c + 2] + cB[c + 2] + vr[k].z, 0.f);
        v.w = fmaxf(v.w * cA[c + 3] + cB[c + 3] + vr[k].w, 0.f);
        dst[i] = v;
        float *x = X + (size_t)pos * LDX + c;
        x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
    }
    __syncthreads();
    float *hraw = P.hraw + (size_t)b * 6 * cells;
    for (int i = tid; i < 6 * cells; i += NTH) {
        const int o = i / cells, pos = i - o * cells;
        const float *x = X + (size_t)pos * LDX, *w = W + o * C;
        float s = 0.f;
#pragma unroll 8
        for (int c = 0; c < C; ++c) s += x[c] * w[c];
        hraw[i] = s;
        atomicAdd(&red[o * 2], s);          // (uniform-address LDS atomics: the compiler reduces them across the wave first;
        atomicAdd(&red[o * 2 + 1], s * s);  //  a hand-written per-plane wave reduction measured 17.1 vs 12.8 us)
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
__global__ __launch_bounds__(TRN_MID_THREADS) void k_trn_heads_fc(TrnDev P) {
    constexpr int NTH = TRN_MID_THREADS, NW = NTH / 64;       // (84 registers a thread: see TRN_MID_THREADS)
    __shared__ float ha[6 * 128];           // activated head planes [o][pos] (value 0..1, policy 2..5), flat = the FC inputs
    __shared__ float xh[6 * 128];           // their xhat
    __shared__ float h2[64], dh2s[64], logit[128], dlog[128], gflat[6 * 128], gpart[1024];
    __shared__ float cS[6], cT[6], cMn[6], cIv[6];
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
    for (int i = tid; i < 6 * cells; i += NTH) {
        const int o = i / cells;
        const float r = hraw[i], v = fmaxf(r * cS[o] + cT[o], 0.f);
        ha[i] = v;
        xh[i] = (r - cMn[o]) * cIv[o];
        hact[i] = v;
    }
    __syncthreads();
    // value_fc2 (2 cells -> 64) + ReLU and move_fc (4 cells -> cells): a wave per group of four outputs, lanes along
    // the input (four independent loads per lane and step, the wave reductions after the loop)
    const int KV = 2 * cells, KPp = 4 * cells;
    const int n_fc2 = 16, n_mf = (cells + 3) / 4;               // groups of four outputs
    // (every load unconditional at a clamped index, the surplus multiplied away: behind a per-lane condition the
    // compiler waits for each load in turn -- ~100 L2 round trips per wave, 25 us of this kernel's 40)
    for (int grp = wave; grp < n_fc2 + n_mf; grp += NW) {
        const bool v = grp < n_fc2;
        const int o0 = v ? grp * 4 : (grp - n_fc2) * 4, K = v ? KV : KPp, rows = v ? 64 : cells;
        const float *w = v ? P.fc2w : P.mfw, *x = v ? ha : ha + 2 * cells;
        const float *wr[4], *bias = v ? P.fc2b : P.mfb;
        float bs[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            wr[u] = w + (size_t)min(o0 + u, rows - 1) * K;
            bs[u] = bias[min(o0 + u, rows - 1)];
        }
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        if (v) {
            float wv[4][4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {          // K <= 242
                const int ic = min(lane + 64 * it, K - 1);
#pragma unroll
                for (int u = 0; u < 4; ++u) wv[it][u] = wr[u][ic];
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int i = lane + 64 * it;
                const float xv = i < K ? x[min(i, K - 1)] : 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u) s[u] += wv[it][u] * xv;
            }
        } else {
            float wv[8][4];
#pragma unroll
            for (int it = 0; it < 8; ++it) {          // K <= 484
                const int ic = min(lane + 64 * it, K - 1);
#pragma unroll
                for (int u = 0; u < 4; ++u) wv[it][u] = wr[u][ic];
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int i = lane + 64 * it;
                const float xv = i < K ? x[min(i, K - 1)] : 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u) s[u] += wv[it][u] * xv;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float r = wave_sum(s[u]);
            if (lane == 0 && o0 + u < rows) {
                if (v) h2[o0 + u] = fmaxf(r + bs[u], 0.f);
                else logit[o0 + u] = r + bs[u];
            }
        }
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
    // back through the FC layers to the head planes: thread (i, part) sums a slice of the outputs, the parts are
    // combined through LDS.  value: 242 inputs x 4 parts of 16 outputs; policy: 484 inputs x 2 parts of ~61 tiles.
    for (int part = tid / 256; part < 4; part += NTH / 256) {    // value plane: 4 parts
        const int i = tid % 256;
        if (i < KV) {
            float sv = 0.f;
#pragma unroll
            for (int o = part * 16; o < part * 16 + 16; ++o) sv += P.fc2w[(size_t)o * KV + i] * dh2s[o];
            gpart[part * 256 + i] = sv;
        }
    }
    __syncthreads();
    if (tid < KV) gflat[tid] = ha[tid] > 0.f ? (gpart[tid] + gpart[256 + tid]) + (gpart[512 + tid] + gpart[768 + tid]) : 0.f;
    __syncthreads();
    for (int part = tid / 512; part < 2; part += (NTH + 511) / 512) {
        const int i = tid % 512, half = (cells + 1) / 2;
        if (i < KPp) {
            float sp = 0.f;
            const int t1 = min(cells, (part + 1) * half);
#pragma unroll 8
            for (int t = part * half; t < t1; ++t) sp += P.mfw[(size_t)t * KPp + i] * dlog[t];
            gpart[part * 512 + i] = sp;
        }
    }
    __syncthreads();
    if (tid < KPp) gflat[2 * cells + tid] = ha[2 * cells + tid] > 0.f ? gpart[tid] + gpart[512 + tid] : 0.f;
    __syncthreads();
    float *g6 = P.g6 + (size_t)b * 6 * cells;
    for (int i = tid; i < 6 * cells; i += NTH) g6[i] = gflat[i];
    // the two reductions the head BatchNorms' backward needs
    if (tid < 6 * 32) {
        const int o = tid >> 5, j = tid & 31;
        float a = 0.f, q = 0.f;
        for (int pos = j; pos < cells; pos += 32) { const float gv = gflat[o * cells + pos]; a += gv; q += gv * xh[o * cells + pos]; }
#pragma unroll
        for (int sft = 16; sft >= 1; sft >>= 1) { a += __shfl_xor(a, sft); q += __shfl_xor(q, sft); }
        if (j == 0) {
            atomicAdd(&P.hsums[o * 4 + 2], (double)a);
            atomicAdd(&P.hsums[o * 4 + 3], (double)q);
        }
    }
}

// gradients of the FC layers: the batch is the reduction (fixed order).  Block roles by index:
//   [0, nM)        move_fc.weight  [cells][4 cells]: 4 rows x 256 columns per block
//   [nM, nM + nV)  value_fc2.weight [64][2 cells]:   4 rows x 256 columns per block
//   the rest       the biases and value_fc3, four outputs per block
struct HeadGradOffs { size_t fc2w, fc2b, fc3w, fc3b, mfw, mfb; };
__global__ __launch_bounds__(256) void k_trn_heads_wgrad(TrnDev P, HeadGradOffs O) {
    __shared__ float dl[4][256];             // the block's four gradient rows over the batch (B <= 256 per pass)
    const int cells = P.cells, B = P.B, KV = 2 * cells, KPp = 4 * cells, tid = threadIdx.x;
    const int ichunks = (KPp + 255) / 256, tgroups = (cells + 3) / 4, nM = ichunks * tgroups;
    const int vchunks = (KV + 255) / 256, nV = 16 * vchunks;      // (one chunk up to 11x11; two at 13x13: 338 columns)
    const int blk = blockIdx.x;
    if (blk < nM + nV) {
        const bool mf = blk < nM;
        const int r0 = mf ? (blk / ichunks) * 4 : ((blk - nM) / vchunks) * 4;
        const int i = mf ? (blk % ichunks) * 256 + tid : ((blk - nM) % vchunks) * 256 + tid;
        const int rows = mf ? cells : 64, cols = mf ? KPp : KV;
        const float *src = mf ? P.dlogit : P.dh2;
        const int sstride = mf ? P.dl_stride : 64;
        const float *x = P.hact + (mf ? 2 * cells : 0);
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int b0 = 0; b0 < B; b0 += 256) {
            const int nb = min(256, B - b0);
            __syncthreads();
            for (int e = tid; e < 4 * nb; e += 256) {
                const int u = e / nb, bb = e - u * nb;
                dl[u][bb] = r0 + u < rows ? src[(size_t)(b0 + bb) * sstride + r0 + u] : 0.f;
            }
            __syncthreads();
            if (i < cols) {
#pragma unroll 8
                for (int bb = 0; bb < nb; ++bb) {
                    const float xv = x[(size_t)(b0 + bb) * 6 * cells + i];
                    s[0] += dl[0][bb] * xv; s[1] += dl[1][bb] * xv; s[2] += dl[2][bb] * xv; s[3] += dl[3][bb] * xv;
                }
            }
        }
        if (i < cols) {
            float *gdst = P.grad + (mf ? O.mfw : O.fc2w);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (r0 + u < rows) gdst[(size_t)(r0 + u) * cols + i] = s[u];
        }
    } else {
        // biases and value_fc3: a wave per output, its lanes along the batch (blocks nM + 16 ..: four outputs each)
        const int e = (blk - nM - nV) * 4 + (tid >> 6), lane = tid & 63;
        if (e < cells + 64 + 64 + 1) {
            float s = 0.f;
            for (int b = lane; b < B; b += 64) {
                if (e < cells) s += P.dlogit[(size_t)b * P.dl_stride + e];
                else if (e < cells + 64) s += P.dh2[(size_t)b * 64 + e - cells];
                else if (e < cells + 128) s += P.dv3[b] * P.h2[(size_t)b * 64 + e - cells - 64];
                else s += P.dv3[b];
            }
            s = wave_sum(s);
            if (lane == 0) {
                if (e < cells) P.grad[O.mfb + e] = s;
                else if (e < cells + 64) P.grad[O.fc2b + e - cells] = s;
                else if (e < cells + 128) P.grad[O.fc3w + e - cells - 64] = s;
                else P.grad[O.fc3b] = s;
            }
        }
    }
}

// =================================================================================================================
// heads, backward: BatchNorm backward of the two 1x1 convolutions, their weight gradients (per-board partials), the
// gradient into the tower's output with its ReLU mask and BN_L's two reductions
// =================================================================================================================
// (NTH: 1024 threads; 512 at C = 256, where the board image and 1024 threads' running sums do not fit the LDS together)
template <int C, int NTH = (C > 128 ? 512 : TRN_SMALL_THREADS)>
__global__ __launch_bounds__(NTH) void k_trn_heads_bwd(TrnDev P) {
    constexpr int C4 = C / 4, LDX = C + 1, ITER = (121 * C4 + NTH - 1) / NTH;
    extern __shared__ __align__(16) float lds[];
    float *X = lds;                          // [cells][LDX] the tower's output (for the ReLU mask and the conv gradients)
    float *rs = X + (size_t)P.cells * LDX;   // [NTH][8] per-thread channel sums
    __shared__ float dh[6 * 128];
    __shared__ float W[6 * C];
    __shared__ float cA[6], cMn[6], cIv[6], cK1[6], cK2[6];
    __shared__ float pM[C], pI[C];
    const int b = blockIdx.x, tid = threadIdx.x, cells = P.cells, L = P.L;
    const size_t base = (size_t)b * cells * C;
    const float4 *act4 = reinterpret_cast<const float4 *>(P.actL + base), *raw4 = reinterpret_cast<const float4 *>(P.rawL + base);
    const int total = cells * C4;
    float4 va[ITER], vr[ITER];
#pragma unroll
    for (int k = 0; k < ITER; ++k) {           // requested first: they arrive while the coefficients are computed
        const int i = tid + NTH * k;
        va[k] = i < total ? act4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        vr[k] = i < total ? raw4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid < 6) {
        const double m = P.hsums[tid * 4] * (double)P.invN, v = P.hsums[tid * 4 + 1] * (double)P.invN - m * m;
        const float inv = (float)(1.0 / sqrt((v > 0 ? v : 0) + TRN_EPS));
        cMn[tid] = (float)m;
        cIv[tid] = inv;
        cA[tid] = (tid < 2 ? P.hbn_w[0][tid] : P.hbn_w[1][tid - 2]) * inv;
        cK1[tid] = (float)(P.hsums[tid * 4 + 2] * (double)P.invN);
        cK2[tid] = (float)(P.hsums[tid * 4 + 3] * (double)P.invN);
    }
    if (tid >= 64 && tid < 64 + C) bn_coeffs(P, L, tid - 64, pM[tid - 64], pI[tid - 64]);
    for (int i = tid; i < 6 * C; i += NTH) W[i] = i < 2 * C ? P.vconv[i] : P.pconv[i - 2 * C];
    __syncthreads();
    const float *g6 = P.g6 + (size_t)b * 6 * cells, *hraw = P.hraw + (size_t)b * 6 * cells;
    for (int i = tid; i < 6 * cells; i += NTH) {
        const int o = i / cells;
        dh[o * 128 + (i - o * cells)] = cA[o] * (g6[i] - cK1[o] - (hraw[i] - cMn[o]) * cIv[o] * cK2[o]);
    }
    __syncthreads();
    float4 *gL = reinterpret_cast<float4 *>(P.gL + base);
    float a4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f}, vmax = 0.f;
    __shared__ float wmax[NTH / 64];
    const int c0 = (tid % C4) * 4;              // NTH is a multiple of C4: a thread's items share their four channels
#pragma unroll
    for (int k = 0; k < ITER; ++k) {
        const int i = tid + NTH * k;
        if (i >= total) break;
        const int pos = i / C4;
        float d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < 6; ++o) {
            const float h = dh[o * 128 + pos];
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] += h * W[o * C + c0 + j];
        }
        const float a[4] = {va[k].x, va[k].y, va[k].z, va[k].w}, r[4] = {vr[k].x, vr[k].y, vr[k].z, vr[k].w};
        float gv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            gv[j] = a[j] > 0.f ? d[j] : 0.f;
            vmax = fmaxf(vmax, fabsf(gv[j]));
            a4[j] += gv[j];
            q4[j] += gv[j] * (r[j] - pM[c0 + j]) * pI[c0 + j];
            X[(size_t)pos * LDX + c0 + j] = a[j];
        }
        gL[i] = make_float4(gv[0], gv[1], gv[2], gv[3]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { rs[tid * 8 + j] = a4[j]; rs[tid * 8 + 4 + j] = q4[j]; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    if ((tid & 63) == 0) wmax[tid >> 6] = vmax;
    __syncthreads();
    if (tid == NTH - 1) {       // max |g_L|: the range the split-f16 consumers scale by
        float m = 0.f;
        for (int w = 0; w < NTH / 64; ++w) m = fmaxf(m, wmax[w]);
        atomicMax(&P.gmax[L], __float_as_uint(m));
    }
    if (tid < C) {
        const int grp4 = tid / 4, j = tid % 4;
        double a = 0, q = 0;
        for (int th = grp4; th < NTH; th += C4) { a += rs[th * 8 + j]; q += rs[th * 8 + 4 + j]; }
        P.pgsum[((size_t)L * P.B + b) * C + tid] = make_float2((float)a, (float)q);
    }
    // weight gradients of the 1x1 convolutions: this board's share (the tower's output is in LDS now)
    for (int i = tid; i < 6 * C; i += NTH) {
        const int o = i / C, cc = i - o * C;
        float s0 = 0.f, s1 = 0.f;
        int pos = 0;
        for (; pos + 2 <= cells; pos += 2) {
            s0 += dh[o * 128 + pos] * X[(size_t)pos * LDX + cc];
            s1 += dh[o * 128 + pos + 1] * X[(size_t)(pos + 1) * LDX + cc];
        }
        if (pos < cells) s0 += dh[o * 128 + pos] * X[(size_t)pos * LDX + cc];
        atomicAdd(&P.hconv_acc[(size_t)(b % TRN_REP) * 6 * C + i], (double)(s0 + s1));
    }
}

// =================================================================================================================
// stem backward: BN_0 backward, then dL/dT[tap][cell value][cout] (this board's share); k_trn_update turns the
// table's gradient into conv1.weight's and the embedding's
// =================================================================================================================
template <int C>
__global__ __launch_bounds__(TRN_MID_THREADS) void k_trn_stem_bwd(TrnDev P) {
    constexpr int C4 = C / 4, NTH = TRN_MID_THREADS, ITER = (121 * C4 + NTH - 1) / NTH;
    extern __shared__ __align__(16) float lds[];
    float *Dr = lds;                         // [cells][C]
    float *cA = Dr + (size_t)P.cells * C, *cM = cA + C, *cI = cM + C, *cK = cI + C;
    __shared__ unsigned char cellv[128];
    __shared__ unsigned char nbv[9 * 128];
    const int b = blockIdx.x, tid = threadIdx.x, N = P.N, cells = P.cells;
    const size_t base = (size_t)b * cells * C;
    const float4 *g4 = reinterpret_cast<const float4 *>(P.g[0] + base), *r4 = reinterpret_cast<const float4 *>(P.raw[0] + base);
    const int total = cells * C4;
    float4 vg[ITER], vr[ITER];
#pragma unroll
    for (int k = 0; k < ITER; ++k) {
        const int i = tid + NTH * k;
        vg[k] = i < total ? g4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        vr[k] = i < total ? r4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    double t0, t1;
    sum_partials<C, NTH>(P.pgsum, C, 0, P.presum ? 0 : P.B, reinterpret_cast<double2 *>(cK + 2 * C), tid, t0, t1);
    if (tid < C) {
        float mean, inv;
        if (P.presum) presummed(P.sums, 0, C, tid, 2, t0, t1);
        if (b == 0 && !P.presum) {
            P.sums[(size_t)tid * 4 + 2] = t0;
            P.sums[(size_t)tid * 4 + 3] = t1;
        }
        bn_coeffs(P, 0, tid, mean, inv);
        cM[tid] = mean;
        cI[tid] = inv;
        cA[tid] = P.bn_w[0][tid] * inv;
        cK[tid] = (float)(t0 * (double)P.invN);
        cK[C + tid] = (float)(t1 * (double)P.invN);
    }
    for (int i = tid; i < cells; i += NTH) cellv[i] = (unsigned char)P.board[(size_t)b * cells + i];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITER; ++k) {
        const int i = tid + NTH * k;
        if (i >= total) break;
        const int c = (i % C4) * 4;
        float4 v;
        v.x = cA[c] * (vg[k].x - cK[c] - (vr[k].x - cM[c]) * cI[c] * cK[C + c]);
        v.y = cA[c + 1] * (vg[k].y - cK[c + 1] - (vr[k].y - cM[c + 1]) * cI[c + 1] * cK[C + c + 1]);
        v.z = cA[c + 2] * (vg[k].z - cK[c + 2] - (vr[k].z - cM[c + 2]) * cI[c + 2] * cK[C + c + 2]);
        v.w = cA[c + 3] * (vg[k].w - cK[c + 3] - (vr[k].w - cM[c + 3]) * cI[c + 3] * cK[C + c + 3]);
        reinterpret_cast<float4 *>(Dr)[i] = v;
    }
    // dT[tap][v][co] += draw[pos][co] over the positions whose tap-neighbour holds v: a thread per (tap, co), the
    // three cell values in three accumulators; nbv[tap][pos] = that neighbour's value (3 = off the board)
    for (int i = tid; i < 9 * cells; i += NTH) {
        const int tap = i / cells, pos = i - tap * cells;
        const int y = pos / N, x = pos - y * N, yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        nbv[tap * 128 + pos] = (yy >= 0 && yy < N && xx >= 0 && xx < N) ? cellv[yy * N + xx] : (unsigned char)3;
    }
    __syncthreads();
    for (int i = tid; i < 9 * C; i += NTH) {
        const int tap = i / C, co = i - tap * C;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int pos = 0; pos < cells; ++pos) {
            const float v = Dr[(size_t)pos * C + co];
            const int cv = nbv[tap * 128 + pos];
            a0 += cv == 0 ? v : 0.f;
            a1 += cv == 1 ? v : 0.f;
            a2 += cv == 2 ? v : 0.f;
        }
        double *dT = P.stem_dT + (size_t)(b % TRN_REP) * 27 * C;
        atomicAdd(&dT[(tap * 3 + 0) * C + co], (double)a0);
        atomicAdd(&dT[(tap * 3 + 1) * C + co], (double)a1);
        atomicAdd(&dT[(tap * 3 + 2) * C + co], (double)a2);
    }
}

// =================================================================================================================
// SGD update of every tensor, in place (torch.optim.SGD: d = g + wd p; buf = mu buf + d; p -= lr buf).  A block
// handles 256 consecutive elements of one segment; the gradient of an element comes from wherever the backward pass left
// it -- there is no kernel between the backward pass and the update:
//   SEG_CONV    tower filter: the G partial copies of k_trn_wgrad, summed here in a fixed order
//   SEG_PLAIN   already in the flat gradient buffer (the FC layers: k_trn_heads_wgrad)
//   SEG_BNW / SEG_BNB  BatchNorm gamma / beta: the layer's sums {x, x^2, g, g xhat}; the gamma thread also moves the
//               running statistics (momentum 0.1, unbiased variance) and num_batches_tracked
//   SEG_HCONV   the heads' 1x1 convolutions: the boards' shares (TRN_REP copies of f64 atomics)
//   SEG_W1 / SEG_EMB  conv1.weight / encoder.weight from the stem table's gradient dT and each OTHER's value at the
//               start of the step (embw1_old: this kernel is moving both); the SEG_EMB block also writes the loss
// =================================================================================================================
enum { SEG_PLAIN = 0, SEG_CONV, SEG_BNW, SEG_BNB, SEG_HCONV, SEG_W1, SEG_EMB };
struct Segment {
    float *p, *mom;
    size_t n, goff;
    int layer;          // SEG_CONV: tower conv layer 1..L; SEG_BNW / SEG_BNB: BatchNorm 0..L, L + 1 = value_bn1, L + 2 = move_bn1
    int kind, aux;      // SEG_HCONV: offset of the tensor in the [6][C] head-conv gradient
    float *run_mean, *run_var;
    long long *tracked;
};

template <int C>
__global__ __launch_bounds__(256) void k_trn_update(TrnDev P, const Segment *segs, const int2 *blocks, int G) {
    __shared__ float dT[27 * C];
    const int2 bk = blocks[blockIdx.x];
    const Segment S = segs[bk.x];
    const float lr = P.hp[0], mu = P.hp[1], wd = P.hp[2];
    const int tid = threadIdx.x, L = P.L;
    if (S.kind == SEG_EMB) {
        // (one block: 12 entries) dT = the copies' sum, then encoder.weight [v][i4] with 16 threads per entry
        for (int i = tid; i < 27 * C; i += 256) {
            double v[TRN_REP], sd = 0;
#pragma unroll
            for (int r = 0; r < TRN_REP; ++r) v[r] = P.stem_dT[(size_t)r * 27 * C + i];
#pragma unroll
            for (int r = 0; r < TRN_REP; ++r) sd += v[r];
            dT[i] = (float)sd;
        }
        __syncthreads();
        if (tid == 255) {
            const float lv = (float)(P.lossacc[0] / (double)P.B), lm = (float)(P.lossacc[1] / (double)P.B);
            P.loss3[0] = lv + lm;
            P.loss3[1] = lv;
            P.loss3[2] = lm;
        }
        if (tid >= 12 * 16) return;
        const int e = tid / 16, j = tid % 16, v = e / 4, i4 = e - v * 4;
        const float *w1 = P.embw1_old + 12;
        float s = 0.f;
        for (int k = j; k < 9 * C; k += 16) {
            const int tap = k / C, co = k - tap * C;
            s += w1[(co * 4 + i4) * 9 + tap] * dT[(tap * 3 + v) * C + co];
        }
#pragma unroll
        for (int sft = 8; sft >= 1; sft >>= 1) s += __shfl_xor(s, sft);
        if (j != 0) return;
        P.grad[S.goff + e] = s;
        const float p = S.p[e], d = s + wd * p, buf = mu * S.mom[e] + d;
        S.mom[e] = buf;
        S.p[e] = p - lr * buf;
        return;
    }
    const size_t ei = (size_t)bk.y * 256 + tid;
    if (ei >= S.n) return;
    size_t e = ei;
    float gr;
    if (S.kind == SEG_CONV) {
        // the G partial copies of k_trn_wgrad, laid out [tap][co][ci], summed in a fixed order: threads walk THAT
        // order (coalesced reads) and touch the filter / momentum / gradient at (co C + ci) 9 + tap.  All of a
        // thread's reads are requested before the first is used (unconditional at a clamped copy index: in a counted
        // loop the compiler keeps four in flight and the kernel is 16 round trips long)
        const size_t cc9 = S.n / 9, tap = ei / cc9, cc = ei - tap * cc9;
        e = cc * 9 + tap;
        const float *part = P.wpart + (size_t)(S.layer - 1) * G * S.n + ei;
        float v[TRN_WG_GROUPS];
#pragma unroll
        for (int g = 0; g < TRN_WG_GROUPS; ++g) v[g] = part[(size_t)min(g, G - 1) * S.n];
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < TRN_WG_GROUPS; ++g) s4[g & 3] += g < G ? v[g] : 0.f;
        gr = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    } else if (S.kind == SEG_BNW || S.kind == SEG_BNB) {
        const int c = (int)ei;
        const double *sm = S.layer <= L ? P.sums + ((size_t)S.layer * C + c) * 4 : P.hsums + (size_t)(S.layer == L + 1 ? c : 2 + c) * 4;
        const double s0 = sm[0], s1 = sm[1], s2 = sm[2], s3 = sm[3];
        gr = (float)(S.kind == SEG_BNW ? s3 : s2);
        if (S.kind == SEG_BNW) {
            const double Nn = 1.0 / (double)P.invN, mean = s0 / Nn, var = s1 / Nn - mean * mean;
            float *rm = S.run_mean + c, *rv = S.run_var + c;
            *rm = (float)(0.9 * (double)*rm + 0.1 * mean);
            *rv = (float)(0.9 * (double)*rv + 0.1 * (var > 0 ? var : 0) * Nn / (Nn - 1.0));
            if (c == 0) *S.tracked += 1;
        }
    } else if (S.kind == SEG_HCONV) {
        double v[TRN_REP], sd = 0;
#pragma unroll
        for (int r = 0; r < TRN_REP; ++r) v[r] = P.hconv_acc[(size_t)r * 6 * C + S.aux + ei];
#pragma unroll
        for (int r = 0; r < TRN_REP; ++r) sd += v[r];
        gr = (float)sd;
    } else if (S.kind == SEG_W1) {
        // conv1.weight [co][i4][tap] = sum over the three cell values of embedding[v][i4] dT[tap][v][co]
        const int co = (int)ei / 36, r = (int)ei - co * 36, i4 = r / 9, tap = r - i4 * 9;
        double v[3][TRN_REP];
#pragma unroll
        for (int cv = 0; cv < 3; ++cv)
#pragma unroll
            for (int rr = 0; rr < TRN_REP; ++rr) v[cv][rr] = P.stem_dT[(size_t)rr * 27 * C + (tap * 3 + cv) * C + co];
        gr = 0.f;
#pragma unroll
        for (int cv = 0; cv < 3; ++cv) {
            double sd = 0;
#pragma unroll
            for (int rr = 0; rr < TRN_REP; ++rr) sd += v[cv][rr];
            gr += P.embw1_old[cv * 4 + i4] * (float)sd;
        }
    } else {
        gr = P.grad[S.goff + e];
    }
    if (S.kind != SEG_PLAIN) P.grad[S.goff + e] = gr;
    const float p = S.p[e], d = gr + wd * p, buf = mu * S.mom[e] + d, np = p - lr * buf;
    S.mom[e] = buf;
    S.p[e] = np;
}

// The tower filters' update for the wide towers, one launch per layer behind that layer's filter-gradient kernel on the
// side stream (so this HBM-bound pass runs under the MFMA-bound data chain instead of after it).  A block owns 256
// (co, ci) pairs x the nine taps: the G partial copies are read in THEIR order ([tap][co][ci]: coalesced, all of a
// thread's loads of a tap requested together), summed in k_trn_update's fixed order, turned through LDS, and the filter /
// momentum / gradient are walked in THEIR order ((co C + ci) 9 + tap: 9 KB contiguous per block).  (k_trn_update's
// element-per-thread form touches the three tensors at a 36-byte stride: every line nine times, from nine blocks.)
__global__ __launch_bounds__(256) void k_tw_update_conv(TrnDev P, Segment S, int G) {
    __shared__ float gsum[9 * 256];
    const int tid = threadIdx.x;
    const float lr = P.hp[0], mu = P.hp[1], wd = P.hp[2];
    const size_t cc9 = S.n / 9;
    const float *part = P.wpart + (size_t)(S.layer - 1) * G * S.n + (size_t)blockIdx.x * 256 + tid;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
        for (int g0 = 0; g0 < G; g0 += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)tap * cc9 + (size_t)min(g0 + u, G - 1) * S.n];
#pragma unroll
            for (int u = 0; u < 8; ++u) s4[u & 3] += g0 + u < G ? v[u] : 0.f;
        }
        gsum[tid * 9 + tap] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    }
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * 256 * 9;
    float p[9], m[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { p[k] = S.p[base + k * 256 + tid]; m[k] = S.mom[base + k * 256 + tid]; }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const size_t e = base + k * 256 + tid;
        const float gr = gsum[k * 256 + tid];
        P.grad[S.goff + e] = gr;
        const float d = gr + wd * p[k], buf = mu * m[k] + d;
        S.mom[e] = buf;
        S.p[e] = p[k] - lr * buf;
    }
}

// First kernel of a step, grid (C C 9 / 256, L + 1):
//   y < L   max |filter y + 1| (per-block maxima: the scale of the split-f16 fragments k_trn_stem_fwd writes) and the
//           fp32 MFMA-order copies the exact-fp32 roles convolve with: whatever wrote the weights last -- this trainer's
//           own update, an eager optimizer step on a ragged batch, load_state_dict, a weight broadcast -- the step
//           convolves with what the tensors hold NOW.
//             forward pack        [tap][q][ntile][lane = j + 32 h][t] = W[co = 32 ntile + j][ci = 8 q + 4 h + t][tap]
//             backward-data pack  the same order for the transposed, flipped filter W'[n = ci][k = co][8 - tap]
//   y == L  the per-step accumulators zeroed, the stem table (embedding folded through conv1), and the step's
//           hyper-parameters fetched from the host's pinned ring (slot = steps run so far)
template <int C>
__global__ __launch_bounds__(256) void k_trn_prep(TrnDev P) {
    constexpr int NT = (C + 31) / 32, Q = C / 8;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if ((int)blockIdx.y == P.L) {
        for (size_t i = e; i < (size_t)P.zero_count; i += (size_t)gridDim.x * 256) P.zero_base[i] = 0.0;
        if (e < (size_t)27 * C) {
            const int k = (int)(e / C), co = (int)(e % C), tap = k / 3, v = k - tap * 3;
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) sum += P.emb[v * 4 + j] * P.w1[(co * 4 + j) * 9 + tap];
            P.stemT[e] = sum;
        }
        if (e <= (size_t)P.L) P.gmax[e] = 0u;
        if (P.gslots && e < (size_t)(P.L + 1) * TW_GSLOTS) P.gslots[e] = 0u;
        if (e < (size_t)12 + C * 36) P.embw1_old[e] = e < 12 ? P.emb[e] : P.w1[e - 12];
        if (e == 0) {
            const unsigned int step = *P.step_ctr;
            const float *slot = P.hp_ring + (size_t)(step % TRN_HP_SLOTS) * 4;
            P.hp[0] = slot[0];
            P.hp[1] = slot[1];
            P.hp[2] = slot[2];
            *P.step_ctr = step + 1;
        }
        return;
    }
    const int l = blockIdx.y + 1;
    const bool on = e < (size_t)C * C * 9;
    const float v = on ? P.convw[l][e] : 0.f;
    {   // max |filter|: the scale of the split-f16 packs (k_trn_stem_fwd writes them)
        __shared__ float wm[4];
        float m = fabsf(v);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) P.wpmax[(size_t)l * gridDim.x + blockIdx.x] = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    }
    if (!on) return;
    const int tap = (int)(e % 9), ci = (int)(e / 9 % C), co = (int)(e / 9 / C);
    float *wf = P.Wf[l], *wb = P.Wb[l];
    if (wf == nullptr) return;            // wide towers: no exact-fp32 roles, no fp32 packs
    wf[((((size_t)tap * Q + (ci >> 3)) * NT + (co >> 5)) * 64 + (co & 31) + 32 * ((ci >> 2) & 1)) * 4 + (ci & 3)] = v;
    wb[((((size_t)(8 - tap) * Q + (co >> 3)) * NT + (ci >> 5)) * 64 + (ci & 31) + 32 * ((co >> 2) & 1)) * 4 + (co & 3)] = v;
}

// =================================================================================================================
// wide towers: C = 128 / 256 (BASELINE configs[4]'s width) on boards up to 13x13 -- and C = 64 on 12x12 / 13x13
//
// A 6x64 step is a chain of launches whose matrix work is ~1.5 us each: everything above is built around latency.  At
// 256 channels a layer pass is 18 GFLOP: throughput-bound, and a (board, 32 channels) workgroup that stages the whole
// input would re-read and re-normalise it eight times.  So the wide step is built the other way round:
//   * the three 3x3 convolutions per layer -- forward, backward-data, filter gradient -- run on the split-f16 MFMA
//     kernels that exist for this shape: the self-play tower's k_conv_wide_f16x3_s16 (net_kernels.hip) in its TRAIN mode
//     (raw fp32 output, per-board channel sums in the epilogue, no bias / residual / ReLU) for the first two, on
//     split-f16 IMAGES [B][cells][C hi | C lo] of their operand, and k_trn_wgrad16<C> as it is for the third;
//   * everything between two convolutions is one elementwise pass that writes the next convolution's image:
//       k_tw_bnact   act_l = relu(BN_l(raw_l) [+ act_{l-2}]) on batch statistics -> act_l (fp32, for the backward
//                    pass and the filter gradient) and its image, scaled by the layer's power of two
//       k_tw_bnbwd   draw_l = BatchNorm backward of g_l -> its image, scaled from max |g_l| like ROLE_BWD16
//     and the ReLU backward -- g_{l-1} = (conv^T output [+ g_{l+1}]) * (act_{l-1} > 0), its per-board (sum g, sum g xhat),
//     max |g| -- is the backward-data convolution's epilogue (TRAIN = 2 in net_kernels.hip)
//     with the batch sums of a block's 64 channels taken from the per-board partial pairs in the block (fixed order);
//   * the stem, the heads and the update are the kernels above, instantiated at this width.
// =================================================================================================================
int azx_net_wide_train_conv(int N, int C, const unsigned short *w16, const unsigned short *in, float *out32, int n_boards,
                            const float *unscale, float2 *stat, hipStream_t st);      // net_kernels.hip
int azx_net_wide_train_conv_bwd(int N, int C, const unsigned short *w16, const unsigned short *in, float *g_out, int n_boards,
                                const float *unscale, float2 *pgsum, const unsigned char *mask, const float *raw, const float *skip,
                                const double *sums, float invN, unsigned int *gmax, hipStream_t st);

// per-layer scales (the narrow path makes them in k_trn_stem_fwd): filter scale from k_trn_prep's per-block maxima,
// activation scale from the BatchNorm bounds (all C channels), fsc[l] = (sa, 1 / (sa sw), sw, 1 / sw).  One block.
// (grid L + 1: block l reduces layer l's two maxima into bsc[l] -- free until the backward pass --, then a one-block
// launch with `finish` set turns them into fsc)
template <int C>
__global__ __launch_bounds__(256) void k_tw_scales(TrnDev P, int finish) {
    constexpr int NB = (C * C * 9 + 255) / 256;
    __shared__ float sW[TRN_MAXL + 1], sBn[TRN_MAXL + 1], red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cells = P.cells;
    if (!finish) {
        const int l = blockIdx.x;
        float mw = 0.f, mb = 0.f;
        if (l >= 1)
            for (int i = tid; i < NB; i += 256) mw = fmaxf(mw, P.wpmax[(size_t)l * NB + i]);
        const float sq = sqrtf((float)P.B * cells);
        for (int c = tid; c < C; c += 256) mb = fmaxf(mb, fabsf(P.bn_w[l][c]) * sq + fabsf(P.bn_b[l][c]));
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { mw = fmaxf(mw, __shfl_xor(mw, o)); mb = fmaxf(mb, __shfl_xor(mb, o)); }
        if (lane == 0) { red[0][wave] = mw; red[1][wave] = mb; }
        __syncthreads();
        if (tid == 0) {
            const float m = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
            P.bsc[l] = make_float2(pow2_scale(m), fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3])));
            if (l >= 1) P.wmax[l] = __float_as_uint(m);
        }
        return;
    }
    if (tid <= P.L) { const float2 x = P.bsc[tid]; sW[tid] = x.x; sBn[tid] = x.y; }
    __syncthreads();
    if (tid >= 1 && tid <= P.L) {
        const int l = tid;                  // layer l convolves act_{l-1}: bounded by its BatchNorm's bound + its skip chain's
        float bd = sBn[l - 1];
        if (((l - 1) & 1) == 0 && l - 1 >= 2)
            for (int j = l - 3; j >= 0; j -= 2) bd += sBn[j];
        const float sa = pow2_scale(bd), sw = sW[l];
        P.fsc[l] = make_float4(sa, 1.f / (sa * sw), sw, 1.f / sw);
    }
}

// The tower filters of layers l0 .. as hi / lo f16 fragments in the wide pack of k_conv_wide_f16x3_s16
// ([tap][chunk of 64 k][half][ntile of 16][hi, lo][lane = j + 16 h][8 consecutive k], tile row j <-> channel by the wide
// permutation, see net_pack.hip), scaled by the layer's power of two.  A thread owns (n, 8 consecutive k) and all nine
// taps: forward (n = co, k = ci) it reads 72 contiguous floats; backward-data (n = ci, k = co, taps flipped) eight runs
// of nine, consecutive lanes consecutive runs.  grid (C C / 8 / 256, L, 2).
template <int C>
__global__ __launch_bounds__(256) void k_tw_pack(TrnDev P, unsigned short *const *wf, unsigned short *const *wb) {
    constexpr int NCH = C / 64, NT16 = C / 16, KB = C / 8;
    const int l = blockIdx.y + 1, bwd = blockIdx.z;
    const int it = blockIdx.x * 256 + threadIdx.x;
    if (it >= C * KB) return;
    // forward: kb fastest (a row of w is contiguous in ci); backward: n fastest
    const int n = bwd ? it % C : it / KB, kb = bwd ? it / C : it % KB;
    const float sc = P.fsc[l].z;
    const float *w = P.convw[l];
    float v[8][9];
    if (!bwd) {
        const float *src = w + ((size_t)n * C + 8 * kb) * 9;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) v[t][tap] = src[t * 9 + tap] * sc;
    } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float *src = w + ((size_t)(8 * kb + t) * C + n) * 9;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) v[t][8 - tap] = src[tap] * sc;
        }
    }
    // n -> (ntile, row j) by the inverse of: n = 64 (nt / 4) + 32 ((nt % 4) >> 1) + 8 (j >> 2) + 4 (nt & 1) + (j & 3)
    const int g4 = n >> 6, a = (n >> 5) & 1, jq = (n >> 3) & 3, bq = (n >> 2) & 1, r = n & 3;
    const int nt = 4 * g4 + 2 * a + bq, j = 4 * jq + r;
    const int k0 = 8 * kb, ch = k0 >> 6, half = (k0 >> 5) & 1, h = (k0 >> 3) & 3, lane = j + 16 * h;
    uint4 *dst = reinterpret_cast<uint4 *>(bwd ? wb[l] : wf[l]);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        _Float16 hi[8], lo[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) { hi[t] = (_Float16)v[t][tap]; lo[t] = (_Float16)(v[t][tap] - (float)hi[t]); }
        uint4 *o = dst + ((((size_t)(tap * NCH + ch) * 2 + half) * NT16 + nt) * 2) * 64 + lane;
        o[0] = *reinterpret_cast<const uint4 *>(hi);
        o[64] = *reinterpret_cast<const uint4 *>(lo);
    }
}

// eight consecutive channels of one position as hi / lo halves into an image row [C hi | C lo]
__device__ __forceinline__ void image_store(unsigned short *img, size_t row, int C, int c, const float (&v)[8]) {
    _Float16 hi[8], lo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { hi[j] = (_Float16)v[j]; lo[j] = (_Float16)(v[j] - (float)hi[j]); }
    unsigned short *r = img + row * (size_t)(2 * C);
    *reinterpret_cast<uint4 *>(r + c) = *reinterpret_cast<const uint4 *>(hi);
    *reinterpret_cast<uint4 *>(r + C + c) = *reinterpret_cast<const uint4 *>(lo);
}

// elementwise passes: grid (C / 64, B), 256 threads; thread = (8 channels c8, positions p0, p0 + 32, ...)
// this block's 64 channels of a layer's per-board partial pairs [B][C], summed over the boards in a fixed order (thread
// = channel x one of four board phases; all of a thread's loads in flight together); threads < 64 return the totals
__device__ __forceinline__ void tw_slice_totals(const float2 *part, int B, int C, int c0, double2 *sh, int tid, double &a, double &q) {
    const int c = c0 + (tid & 63), ph = tid >> 6;
    double sa = 0, sq = 0;
    for (int b0 = ph; b0 < B; b0 += 64) {
        float2 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int b = b0 + 4 * u;
            const float2 x = part[(size_t)min(b, max(B - 1, 0)) * C + c];
            const float on = b < B ? 1.f : 0.f;
            v[u] = make_float2(x.x * on, x.y * on);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) { sa += (double)v[u].x; sq += (double)v[u].y; }
    }
    sh[tid] = make_double2(sa, sq);
    __syncthreads();
    a = 0;
    q = 0;
    if (tid < 64) {
        const double2 x0 = sh[tid], x1 = sh[64 + tid], x2 = sh[128 + tid], x3 = sh[192 + tid];
        a = (x0.x + x1.x) + (x2.x + x3.x);
        q = (x0.y + x1.y) + (x2.y + x3.y);
    }
}

struct TwAct { const float *raw, *skip, *bnw, *bnb; const float2 *pstat; double *sums; float *act; unsigned short *img; const float4 *fsc; int presum;
               unsigned char *mask; };      // [B][cells][C / 8]: bit j of a byte = (act of channel 8 k + j) > 0, all the backward pass reads of act_l
__global__ __launch_bounds__(256) void k_tw_bnact(TwAct A, int cells, int C, int B, float invN) {
    __shared__ float cA[64], cB[64];
    __shared__ double2 sh[256];
    const int tid = threadIdx.x, c0 = blockIdx.x * 64, b = blockIdx.y;
    double t0, t1;
    tw_slice_totals(A.pstat, A.presum ? 0 : B, C, c0, sh, tid, t0, t1);
    if (tid < 64) {
        const int c = c0 + tid;
        if (A.presum) { t0 = A.sums[(size_t)c * 4]; t1 = A.sums[(size_t)c * 4 + 1]; }      // k_trn_totals made them
        else if (b == 0) { A.sums[(size_t)c * 4] = t0; A.sums[(size_t)c * 4 + 1] = t1; }      // for the backward pass and the update
        float mean, inv;
        bn_from_sums(t0, t1, invN, mean, inv);
        const float a = A.bnw[c] * inv;
        cA[tid] = a;
        cB[tid] = A.bnb[c] - mean * a;
    }
    const float sa = A.img ? A.fsc->x : 1.f;
    __syncthreads();
    const int cl = (tid & 7) * 8, c = c0 + cl;
    for (int pos = tid >> 3; pos < cells; pos += 32) {
        const size_t o = ((size_t)b * cells + pos) * C + c;
        const float4 r0 = *reinterpret_cast<const float4 *>(A.raw + o), r1 = *reinterpret_cast<const float4 *>(A.raw + o + 4);
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
        if (A.skip) { s0 = *reinterpret_cast<const float4 *>(A.skip + o); s1 = *reinterpret_cast<const float4 *>(A.skip + o + 4); }
        const float x[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w}, k[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        float v[8], vs[8];
        unsigned int bits = 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] = fmaxf(x[j] * cA[cl + j] + cB[cl + j] + k[j], 0.f); vs[j] = v[j] * sa; bits |= (v[j] > 0.f ? 1u : 0u) << j; }
        *reinterpret_cast<float4 *>(A.act + o) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4 *>(A.act + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
        if (A.mask) A.mask[o >> 3] = (unsigned char)bits;
        if (A.img) image_store(A.img, (size_t)b * cells + pos, C, c, vs);
    }
}

struct TwBnBwd { const float *g, *raw, *bnw; const float2 *pgsum; double *sums; unsigned int *gmax; const float4 *fsc; unsigned short *img; float2 *bsc; int presum;
                 const unsigned int *gslots; };       // [TW_GSLOTS] max |g_l| as the backward convolution's blocks filed it (beside *gmax: the heads' for l = L)
__global__ __launch_bounds__(256) void k_tw_bnbwd(TwBnBwd A, int cells, int C, int B, float invN) {
    __shared__ float cA[64], cM[64], cI[64], cK0[64], cK1[64], red[4];
    __shared__ double2 sh[256];
    const int tid = threadIdx.x, c0 = blockIdx.x * 64, b = blockIdx.y;
    // max |g_l|: the word the heads' backward files for l = L, and the TW_GSLOTS words the backward convolution's blocks file
    // (every lane reads slot lane & 31: each wave has all of them); block (0, 0) leaves the result in the word for later readers
    float gmax = __uint_as_float(*A.gmax);
    {
        float gs = __uint_as_float(A.gslots[tid & (TW_GSLOTS - 1)]);
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) gs = fmaxf(gs, __shfl_xor(gs, o));
        gmax = fmaxf(gmax, gs);
        if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *A.gmax = __float_as_uint(gmax);
    }
    const float sqn = sqrtf((float)B * cells);
    // The image has ONE scale, so every block needs a bound over all channels.  With the channels' own (mean g, mean g
    // xhat) that would be every channel's partial sums in every block; |mean g| <= max |g| and |mean g xhat| <= max |g|
    // (mean |xhat| <= 1) give |draw| <= |gamma inv| max |g| (2 + sqrt n) from what the forward pass filed -- a bound up
    // to ~sqrt(n) looser, i.e. the pair resolves 2^-31 of the largest entry instead of 2^-38: far below fp32's rounding.
    float bd = 0.f;
    for (int c = tid; c < C; c += 256) {
        float mean, inv;
        bn_from_sums(A.sums[(size_t)c * 4], A.sums[(size_t)c * 4 + 1], invN, mean, inv);
        const float a = A.bnw[c] * inv;
        bd = fmaxf(bd, fabsf(a) * gmax * (2.f + sqn));
        if (c >= c0 && c < c0 + 64) { cA[c - c0] = a; cM[c - c0] = mean; cI[c - c0] = inv; }
    }
    double t0, t1;
    tw_slice_totals(A.pgsum, A.presum ? 0 : B, C, c0, sh, tid, t0, t1);     // (sum g_l, sum g_l xhat_l) of this block's channels
    if (tid < 64) {
        if (A.presum) { t0 = A.sums[(size_t)(c0 + tid) * 4 + 2]; t1 = A.sums[(size_t)(c0 + tid) * 4 + 3]; }
        else if (b == 0) { A.sums[(size_t)(c0 + tid) * 4 + 2] = t0; A.sums[(size_t)(c0 + tid) * 4 + 3] = t1; }   // k_trn_update's BatchNorm gradients
        cK0[tid] = (float)(t0 * (double)invN);
        cK1[tid] = (float)(t1 * (double)invN);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) bd = fmaxf(bd, __shfl_xor(bd, o));
    if ((tid & 63) == 0) red[tid >> 6] = bd;
    __syncthreads();
    const float sc = pow2_scale(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
    if (blockIdx.x == 0 && b == 0 && tid == 0) *A.bsc = make_float2(sc, A.fsc->w / sc);
    const int cl = (tid & 7) * 8, c = c0 + cl;
    for (int pos = tid >> 3; pos < cells; pos += 32) {
        const size_t o = ((size_t)b * cells + pos) * C + c;
        const float4 g0 = *reinterpret_cast<const float4 *>(A.g + o), g1 = *reinterpret_cast<const float4 *>(A.g + o + 4);
        const float4 r0 = *reinterpret_cast<const float4 *>(A.raw + o), r1 = *reinterpret_cast<const float4 *>(A.raw + o + 4);
        const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, rv[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v[j] = sc * (cA[cl + j] * (gv[j] - cK0[cl + j] - (rv[j] - cM[cl + j]) * cI[cl + j] * cK1[cl + j]));
        image_store(A.img, (size_t)b * cells + pos, C, c, v);
    }
}

// ---- the heads and the stem's backward for the wide towers: boards up to 13x13 (the kernels above keep a whole board
// x all channels in LDS and lay the head planes out 128 wide) ---------------------------------------------------------
#define TW_CP 192                  // head-plane / logit arrays: cells padded

// the two 1x1 head convolutions on act_L (k_tw_bnact made it) and their batch sums: a wave per position
__global__ __launch_bounds__(TRN_SMALL_THREADS) void k_tw_hconv(TrnDev P) {
    __shared__ float W[6 * 256], red[12];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cells = P.cells, C = P.C;
    for (int i = tid; i < 6 * C; i += TRN_SMALL_THREADS) W[i] = i < 2 * C ? P.vconv[i] : P.pconv[i - 2 * C];
    if (tid < 12) red[tid] = 0.f;
    __syncthreads();
    float a1[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, a2[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float *hraw = P.hraw + (size_t)b * 6 * cells;
    for (int pos = wave; pos < cells; pos += TRN_SMALL_THREADS / 64) {
        const float *x = P.actL + ((size_t)b * cells + pos) * C;
        float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int c = lane * 4; c < C; c += 256) {
            const float4 v = *reinterpret_cast<const float4 *>(x + c);
#pragma unroll
            for (int o = 0; o < 6; ++o) s[o] += v.x * W[o * C + c] + v.y * W[o * C + c + 1] + v.z * W[o * C + c + 2] + v.w * W[o * C + c + 3];
        }
#pragma unroll
        for (int o = 0; o < 6; ++o) {
            const float r = wave_sum(s[o]);
            if (lane == 0) { hraw[o * cells + pos] = r; a1[o] += r; a2[o] += r * r; }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int o = 0; o < 6; ++o) { atomicAdd(&red[o * 2], a1[o]); atomicAdd(&red[o * 2 + 1], a2[o]); }
    }
    __syncthreads();
    if (tid < 12) atomicAdd(&P.hsums[(tid >> 1) * 4 + (tid & 1)], (double)red[tid]);
}

// k_trn_heads_fc for any board up to TW_CP cells: BN + ReLU of the head planes, the FC layers, masked log-softmax, the
// loss and the gradient back to the head planes (network.py:77-102, :146-152); plain loops, one block per board
__global__ __launch_bounds__(TRN_SMALL_THREADS) void k_tw_heads_fc(TrnDev P) {
    constexpr int NTH = TRN_SMALL_THREADS, NW = NTH / 64, CP = TW_CP;
    __shared__ float ha[6 * CP], xh[6 * CP], h2[64], dh2s[64], logit[CP], dlog[CP], gflat[6 * CP];
    __shared__ float cS[6], cT[6], cMn[6], cIv[6];
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
    for (int i = tid; i < 6 * cells; i += NTH) {
        const int o = i / cells;
        const float r = hraw[i], v = fmaxf(r * cS[o] + cT[o], 0.f);
        ha[i] = v;
        xh[i] = (r - cMn[o]) * cIv[o];
        hact[i] = v;
    }
    __syncthreads();
    const int KV = 2 * cells, KPp = 4 * cells;
    // value_fc2 (2 cells -> 64) + ReLU and move_fc (4 cells -> cells): a wave per output, lanes along the input
    for (int o = wave; o < 64 + cells; o += NW) {
        const bool v = o < 64;
        const int oo = v ? o : o - 64, K = v ? KV : KPp;
        const float *w = (v ? P.fc2w : P.mfw) + (size_t)oo * K, *x = v ? ha : ha + 2 * cells;
        float s = 0.f;
        for (int i = lane; i < K; i += 64) s += w[i] * x[i];
        s = wave_sum(s);
        if (lane == 0) {
            if (v) h2[oo] = fmaxf(s + P.fc2b[oo], 0.f);
            else logit[oo] = s + P.mfb[oo];
        }
    }
    __syncthreads();
    if (wave == 0) {
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
        // gather the legal moves' logits, log-softmax over them (padding entries: -99 in the reference, exp(-99 - max)
        // of the sum: below fp32's resolution), policy loss and dL/dlogit by tile
        const int32_t *lm = P.legal + (size_t)b * cells;
        const float *mp = P.prob + (size_t)b * cells;
        float *lp = P.logprob + (size_t)b * cells;
        constexpr int R = CP / 64;
        int m[R];
        float x[R], pr[R];
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int j = lane + 64 * r;
            m[r] = j < cells ? lm[j] : 0;
            pr[r] = j < cells ? mp[j] : 0.f;
            x[r] = m[r] > 0 ? logit[m[r] - 1] : -INFINITY;
            mx = fmaxf(mx, x[r]);
        }
        mx = wave_max(mx);
        float es = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) es += m[r] > 0 ? expf(x[r] - mx) : 0.f;
        const float lse = mx + logf(wave_sum(es));
        float S = 0.f, nll = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            S += m[r] > 0 ? pr[r] : 0.f;
            nll += m[r] > 0 ? -pr[r] * (x[r] - lse) : 0.f;
        }
        S = wave_sum(S);
        nll = wave_sum(nll);
        for (int t = lane; t < CP; t += 64) dlog[t] = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int j = lane + 64 * r;
            if (j < cells) lp[j] = m[r] > 0 ? x[r] - lse : -99.f - lse;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (m[r] > 0) dlog[m[r] - 1] = (expf(x[r] - lse) * S - pr[r]) / (float)B;
        if (lane == 0) atomicAdd(&P.lossacc[1], (double)nll);
    }
    __syncthreads();
    for (int t = tid; t < P.dl_stride; t += NTH) P.dlogit[(size_t)b * P.dl_stride + t] = t < cells ? dlog[t] : 0.f;
    // back through the FC layers to the head planes: a thread per input
    for (int i = tid; i < KV + KPp; i += NTH) {
        float sg = 0.f;
        if (i < KV) {
            for (int o = 0; o < 64; ++o) sg += P.fc2w[(size_t)o * KV + i] * dh2s[o];
        } else {
            const int ip = i - KV;
            for (int t = 0; t < cells; ++t) sg += P.mfw[(size_t)t * KPp + ip] * dlog[t];
        }
        gflat[i] = ha[i] > 0.f ? sg : 0.f;
    }
    __syncthreads();
    float *g6 = P.g6 + (size_t)b * 6 * cells;
    for (int i = tid; i < 6 * cells; i += NTH) g6[i] = gflat[i];
    if (tid < 6 * 32) {        // the two reductions the head BatchNorms' backward needs
        const int o = tid >> 5, j = tid & 31;
        float a = 0.f, q = 0.f;
        for (int pos = j; pos < cells; pos += 32) { const float gv = gflat[o * cells + pos]; a += gv; q += gv * xh[o * cells + pos]; }
#pragma unroll
        for (int sft = 16; sft >= 1; sft >>= 1) { a += __shfl_xor(a, sft); q += __shfl_xor(q, sft); }
        if (j == 0) {
            atomicAdd(&P.hsums[o * 4 + 2], (double)a);
            atomicAdd(&P.hsums[o * 4 + 3], (double)q);
        }
    }
}

// k_trn_heads_bwd by (64 channels, board): BatchNorm backward of the head planes, the gradient into the tower's output
// with its ReLU mask, BN_L's two reductions as this board's partial pair, max |g_L|, and the 1x1 filters' gradients
__global__ __launch_bounds__(256) void k_tw_heads_bwd(TrnDev P) {
    __shared__ float dh[6][TW_CP], W6[6][64], pM[64], pI[64], rs[256][17], wm[4];
    __shared__ float cA[6], cMn[6], cIv[6], cK1[6], cK2[6];
    const int tid = threadIdx.x, c0 = blockIdx.x * 64, b = blockIdx.y, cells = P.cells, C = P.C, L = P.L;
    if (tid < 6) {
        const double m = P.hsums[tid * 4] * (double)P.invN, v = P.hsums[tid * 4 + 1] * (double)P.invN - m * m;
        const float inv = (float)(1.0 / sqrt((v > 0 ? v : 0) + TRN_EPS));
        cMn[tid] = (float)m;
        cIv[tid] = inv;
        cA[tid] = (tid < 2 ? P.hbn_w[0][tid] : P.hbn_w[1][tid - 2]) * inv;
        cK1[tid] = (float)(P.hsums[tid * 4 + 2] * (double)P.invN);
        cK2[tid] = (float)(P.hsums[tid * 4 + 3] * (double)P.invN);
    }
    if (tid >= 64 && tid < 128) bn_coeffs(P, L, c0 + tid - 64, pM[tid - 64], pI[tid - 64]);
    for (int i = tid; i < 6 * 64; i += 256) {
        const int o = i >> 6, cc = c0 + (i & 63);
        W6[o][i & 63] = o < 2 ? P.vconv[(size_t)o * C + cc] : P.pconv[(size_t)(o - 2) * C + cc];
    }
    __syncthreads();
    const float *g6 = P.g6 + (size_t)b * 6 * cells, *hraw = P.hraw + (size_t)b * 6 * cells;
    for (int i = tid; i < 6 * cells; i += 256) {
        const int o = i / cells;
        dh[o][i - o * cells] = cA[o] * (g6[i] - cK1[o] - (hraw[i] - cMn[o]) * cIv[o] * cK2[o]);
    }
    __syncthreads();
    const int cl = (tid & 7) * 8, c = c0 + cl;
    float a8[8], q8[8], wacc[6][8], vmax = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a8[j] = 0.f; q8[j] = 0.f; }
#pragma unroll
    for (int o = 0; o < 6; ++o)
#pragma unroll
        for (int j = 0; j < 8; ++j) wacc[o][j] = 0.f;
    for (int pos = tid >> 3; pos < cells; pos += 32) {
        const size_t off = ((size_t)b * cells + pos) * C + c;
        const float4 m0 = *reinterpret_cast<const float4 *>(P.actL + off), m1 = *reinterpret_cast<const float4 *>(P.actL + off + 4);
        const float4 r0 = *reinterpret_cast<const float4 *>(P.rawL + off), r1 = *reinterpret_cast<const float4 *>(P.rawL + off + 4);
        const float mv[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w}, rv[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        float d[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, v[8];
#pragma unroll
        for (int o = 0; o < 6; ++o) {
            const float h = dh[o][pos];
#pragma unroll
            for (int j = 0; j < 8; ++j) { d[j] += h * W6[o][cl + j]; wacc[o][j] += h * mv[j]; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = mv[j] > 0.f ? d[j] : 0.f;
            vmax = fmaxf(vmax, fabsf(v[j]));
            a8[j] += v[j];
            q8[j] += v[j] * (rv[j] - pM[cl + j]) * pI[cl + j];
        }
        *reinterpret_cast<float4 *>(P.gL + off) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4 *>(P.gL + off + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { rs[tid][j] = a8[j]; rs[tid][8 + j] = q8[j]; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    if ((tid & 63) == 0) wm[tid >> 6] = vmax;
    __syncthreads();
    if (tid == 64) atomicMax(&P.gmax[L], __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
    const int grp = (tid & 63) >> 3, jj = tid & 7;
    if (tid < 64) {
        double a = 0, q = 0;
        for (int p = 0; p < 32; ++p) { a += rs[p * 8 + grp][jj]; q += rs[p * 8 + grp][8 + jj]; }
        P.pgsum[((size_t)L * P.B + b) * C + c0 + tid] = make_float2((float)a, (float)q);
    }
    // the 1x1 filters' gradients: this board's share of [6][C], two planes per round through rs
#pragma unroll
    for (int o0 = 0; o0 < 6; o0 += 2) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) { rs[tid][j] = wacc[o0][j]; rs[tid][8 + j] = wacc[o0 + 1][j]; }
        __syncthreads();
        if (tid < 128) {
            const int u = tid >> 6;
            float sacc = 0.f;
            for (int p = 0; p < 32; ++p) sacc += rs[p * 8 + grp][8 * u + jj];
            atomicAdd(&P.hconv_acc[(size_t)(b % TRN_REP) * 6 * C + (size_t)(o0 + u) * C + c0 + (tid & 63)], (double)sacc);
        }
    }
}

// k_trn_stem_bwd by (64 channels, board): BN_0 backward, then this board's share of dL/dT[tap][cell value][cout]
__global__ __launch_bounds__(256) void k_tw_stem_bwd(TrnDev P) {
    __shared__ float Dr[176][65];
    __shared__ float cA[64], cM[64], cI[64], cK0[64], cK1[64];
    __shared__ double2 sh[256];
    __shared__ unsigned char cellv[176], nbv[9][176];
    const int tid = threadIdx.x, c0 = blockIdx.x * 64, b = blockIdx.y, cells = P.cells, C = P.C, N = P.N;
    double t0, t1;
    tw_slice_totals(P.pgsum, P.presum ? 0 : P.B, C, c0, sh, tid, t0, t1);
    if (tid < 64) {
        const int c = c0 + tid;
        if (P.presum) presummed(P.sums, 0, C, c, 2, t0, t1);
        else if (b == 0) { P.sums[(size_t)c * 4 + 2] = t0; P.sums[(size_t)c * 4 + 3] = t1; }
        float mean, inv;
        bn_coeffs(P, 0, c, mean, inv);
        cM[tid] = mean;
        cI[tid] = inv;
        cA[tid] = P.bn_w[0][c] * inv;
        cK0[tid] = (float)(t0 * (double)P.invN);
        cK1[tid] = (float)(t1 * (double)P.invN);
    }
    for (int i = tid; i < cells; i += 256) cellv[i] = (unsigned char)P.board[(size_t)b * cells + i];
    __syncthreads();
    const int cl = (tid & 7) * 8;
    for (int pos = tid >> 3; pos < cells; pos += 32) {
        const size_t off = ((size_t)b * cells + pos) * C + c0 + cl;
        const float4 g0 = *reinterpret_cast<const float4 *>(P.g[0] + off), g1 = *reinterpret_cast<const float4 *>(P.g[0] + off + 4);
        const float4 r0 = *reinterpret_cast<const float4 *>(P.raw[0] + off), r1 = *reinterpret_cast<const float4 *>(P.raw[0] + off + 4);
        const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, rv[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j)
            Dr[pos][cl + j] = cA[cl + j] * (gv[j] - cK0[cl + j] - (rv[j] - cM[cl + j]) * cI[cl + j] * cK1[cl + j]);
    }
    for (int i = tid; i < 9 * cells; i += 256) {
        const int tap = i / cells, pos = i - tap * cells;
        const int y = pos / N, x = pos - y * N, yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        nbv[tap][pos] = (yy >= 0 && yy < N && xx >= 0 && xx < N) ? cellv[yy * N + xx] : (unsigned char)3;
    }
    __syncthreads();
    for (int i = tid; i < 9 * 64; i += 256) {
        const int tap = i >> 6, co = i & 63;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int pos = 0; pos < cells; ++pos) {
            const float v = Dr[pos][co];
            const int cv = nbv[tap][pos];
            a0 += cv == 0 ? v : 0.f;
            a1 += cv == 1 ? v : 0.f;
            a2 += cv == 2 ? v : 0.f;
        }
        double *dT = P.stem_dT + (size_t)(b % TRN_REP) * 27 * C;
        atomicAdd(&dT[(tap * 3 + 0) * C + c0 + co], (double)a0);
        atomicAdd(&dT[(tap * 3 + 1) * C + c0 + co], (double)a1);
        atomicAdd(&dT[(tap * 3 + 2) * C + c0 + co], (double)a2);
    }
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
    int n_blocks = 0, n_conv_blocks = 0;
    HeadGradOffs hoffs;
    std::vector<size_t> conv_goff;
    std::vector<Segment> conv_seg;                       // wide towers: the tower filters' segments, updated per layer (k_tw_update_conv)
    float *hp_dev = nullptr, *hp_ring = nullptr;       // hp_ring: pinned host memory the prep kernel reads
    unsigned long long host_steps = 0;
    bool broken = false;                                 // a step failed while being queued (see azx_trn_step)
    hipEvent_t ring_ev[2] = {nullptr, nullptr};         // marks of steps TRN_HP_SLOTS / 2 apart: the host never laps the device
    int32_t *in_board = nullptr, *in_legal = nullptr;
    float *in_prob = nullptr, *in_reward = nullptr;
    size_t zero_bytes = 0;       // sums + hsums + lossacc, contiguous
    size_t grad_floats = 0;
    std::map<std::string, std::pair<void *, size_t>> dbg;      // name -> (device ptr, bytes)
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipStream_t cap = nullptr, side = nullptr;
    std::vector<hipEvent_t> events;
    bool use_graph = false;
    bool fwd16 = true;           // AZX_TRAIN_FWD=fp32: the forward convolutions on the fp32 MFMA as well
    bool bwd16 = true;           // AZX_TRAIN_BWD=fp32: the backward-data convolutions ...
    bool wgrad16 = true;         // AZX_TRAIN_WGRAD=fp32: ... and the filter gradients
    bool fork = true;            // AZX_TRAIN_FORK=0: the weight-gradient passes in line with the data chain (profiling)
    // wide towers (C = 128 / 256)
    bool wide = false;
    bool wgrad2 = false;         // 256 channels: the two-ci-tile, six-owner-wave filter gradient (k_tw_wgrad2); AZX_TRAIN_WGRAD2=0 / 1 overrides
    std::vector<unsigned short *> Ww16f, Ww16b, A16;       // per layer: the wide filter packs, the activations' images
    unsigned short **Ww16f_dev = nullptr, **Ww16b_dev = nullptr;
    std::vector<unsigned short *> D16;                      // per layer: the BatchNorm-backward images (k_tw_wgrad reads them later)
    std::vector<unsigned char *> relu_mask;                 // per layer: act_l > 0, one bit per element (the backward convolution's epilogue)
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
    if (chans != 16 && chans != 32 && chans != 64 && chans != 128 && chans != 256)
        return tfail(AZX_EINVAL, "train: base_chans must be 16, 32, 64, 128 or 256");
    // 64 channels on 12x12 / 13x13 take the wide step too (its kernels tile a board as 176 position rows and need C to be
    // a multiple of 64): the reference's width on BASELINE configs[4]'s board
    const bool wide = chans >= 128 || (chans == 64 && N > 11);
    if (N < 2 || N > (wide ? 13 : 11))
        return tfail(AZX_EINVAL, "train: the native step covers boards up to 11x11 with 16 / 32 channels (121 cells = four 32-row "
                                 "MFMA tiles) and up to 13x13 with 64 / 128 / 256");
    if (wide && N < 3) return tfail(AZX_EINVAL, "train: the wide towers (128 / 256 channels) need a board of 3x3 or more");
    if (blocks < 1 || batch < 1 || 2 * blocks > TRN_MAXL) return tfail(AZX_EINVAL, "train: num_blocks must be 1..19 and the batch positive");
    AzxTrain *t = new AzxTrain();
    memset(&t->d, 0, sizeof t->d);
    t->device = device;
    t->fwd16 = !(getenv("AZX_TRAIN_FWD") && !strcmp(getenv("AZX_TRAIN_FWD"), "fp32"));
    t->bwd16 = !(getenv("AZX_TRAIN_BWD") && !strcmp(getenv("AZX_TRAIN_BWD"), "fp32"));
    t->wgrad16 = !(getenv("AZX_TRAIN_WGRAD") && !strcmp(getenv("AZX_TRAIN_WGRAD"), "fp32"));
    TrnDev &d = t->d;
    d.N = N; d.cells = N * N; d.C = chans; d.L = 2 * blocks; d.B = batch;
    d.invN = (float)(1.0 / ((double)batch * d.cells));
    d.dl_stride = d.cells > 128 ? 192 : 128;
    d.presum = d.B > TRN_PRESUM_BATCH;
    t->G = std::min(batch, TRN_WG_GROUPS);
    t->wide = wide;
    if (wide && chans == 64) t->wgrad16 = true;      // (k_trn_wgrad16<64> stages boards of up to 121 cells: no fp32 filter gradient on 12x12 / 13x13)
    // 256 channels only: at 128 the two are equal on 13x13 (4.84 vs 4.85 ms a step) and the new one loses on small boards
    // (10x128 on 9x9: 2.30 vs 2.12 ms) -- AZX_TRAIN_WGRAD2=1 forces it there, =0 switches it off
    t->wgrad2 = wide && chans >= 128 && (getenv("AZX_TRAIN_WGRAD2") ? atoi(getenv("AZX_TRAIN_WGRAD2")) != 0 : chans == 256);
    // wide: (C / 32)^2 tile pairs already fill the chip with few board groups, and a group costs a partial copy of C C 9
    if (wide) t->G = std::min(batch, chans == 256 ? 8 : 32);      // (C = 256: 8 groups measured 10.7 ms per step against 11.1 with 16)
    if (getenv("AZX_TRAIN_WG_GROUPS")) t->G = std::max(1, std::min(t->G, atoi(getenv("AZX_TRAIN_WG_GROUPS"))));
    const int L = d.L, C = chans, cells = d.cells, B = batch;
    const size_t A = (size_t)B * cells * C;
    t->raw.resize(L + 1); t->act.resize(L + 1); t->g.resize(L + 1); t->Wf.assign(L + 1, nullptr); t->Wb.assign(L + 1, nullptr);
    bool ok = true;
    for (int l = 0; l <= L; ++l) {
        ok = ok && (t->raw[l] = talloc<float>(t, A)) && (t->act[l] = talloc<float>(t, A)) && (t->g[l] = talloc<float>(t, A));
        if (l >= 1 && !wide) {
            const size_t wn = (size_t)9 * (C / 8) * ((C + 31) / 32) * 64 * 4;
            ok = ok && (t->Wf[l] = talloc<float>(t, wn)) && (t->Wb[l] = talloc<float>(t, wn));
            if (t->fwd16) ok = ok && (d.Wf16[l] = talloc<unsigned short>(t, (size_t)9 * (C / 16) * ((C + 31) / 32) * 2 * 64 * 8));
            if (t->bwd16) ok = ok && (d.Wb16[l] = talloc<unsigned short>(t, (size_t)9 * (C / 16) * ((C + 31) / 32) * 2 * 64 * 8));
        }
    }
    if (wide) {
        // the wide packs ([tap][chunk][half][ntile][hi, lo][lane][8]: 2 x 9 C C halves per layer and direction), the
        // activations' images (as many bytes as the fp32 tensors) and the two per-layer scratch tensors
        t->Ww16f.assign(L + 1, nullptr); t->Ww16b.assign(L + 1, nullptr); t->A16.assign(L + 1, nullptr);
        for (int l = 1; l <= L && ok; ++l)
            ok = (t->Ww16f[l] = talloc<unsigned short>(t, (size_t)2 * 9 * C * C)) && (t->Ww16b[l] = talloc<unsigned short>(t, (size_t)2 * 9 * C * C));
        for (int l = 0; l < L && ok; ++l) ok = (t->A16[l] = talloc<unsigned short>(t, 2 * A)) != nullptr;
        t->D16.assign(L + 1, nullptr);
        for (int l = 1; l <= L && ok; ++l) ok = (t->D16[l] = talloc<unsigned short>(t, 2 * A)) != nullptr;
        t->relu_mask.assign(L + 1, nullptr);
        for (int l = 0; l < L && ok; ++l) ok = (t->relu_mask[l] = talloc<unsigned char>(t, A / 8)) != nullptr;
        ok = ok &&
             (t->Ww16f_dev = upload_table(t, t->Ww16f)) && (t->Ww16b_dev = upload_table(t, t->Ww16b)) &&
             (d.bsc = talloc<float2>(t, TRN_MAXL + 2));
    }
    // sums | hsums | lossacc contiguous: one memset per step
    const size_t nsum = (size_t)(L + 1) * C * 4 + 6 * 4 + 2 + (size_t)TRN_REP * (27 * C + 6 * C);
    double *z = ok ? talloc<double>(t, nsum) : nullptr;
    ok = ok && z;
    if (ok) {
        d.sums = z; d.hsums = z + (size_t)(L + 1) * C * 4; d.lossacc = d.hsums + 24;
        d.stem_dT = d.lossacc + 2; d.hconv_acc = d.stem_dT + (size_t)TRN_REP * 27 * C;
        t->zero_bytes = nsum * sizeof(double);
        d.zero_base = z;
        d.zero_count = (int)nsum;
    }
    ok = ok && (d.hraw = talloc<float>(t, (size_t)B * 6 * cells)) && (d.hact = talloc<float>(t, (size_t)B * 6 * cells)) &&
         (d.g6 = talloc<float>(t, (size_t)B * 6 * cells)) && (d.h2 = talloc<float>(t, (size_t)B * 64)) &&
         (d.dh2 = talloc<float>(t, (size_t)B * 64)) && (d.dv3 = talloc<float>(t, B)) &&
         (d.dlogit = talloc<float>(t, (size_t)B * (d.cells > 128 ? 192 : 128))) && (d.value = talloc<float>(t, B)) &&
         (d.logprob = talloc<float>(t, (size_t)B * cells)) && (d.loss3 = talloc<float>(t, 4)) &&
         (d.wpart = talloc<float>(t, (size_t)L * t->G * C * C * 9)) &&
         (t->in_board = talloc<int32_t>(t, (size_t)B * cells)) && (t->in_legal = talloc<int32_t>(t, (size_t)B * cells)) &&
         (t->in_prob = talloc<float>(t, (size_t)B * cells)) && (t->in_reward = talloc<float>(t, B)) &&
         (t->hp_dev = talloc<float>(t, 4)) && (d.stemT = talloc<float>(t, (size_t)27 * C)) && (d.step_ctr = talloc<unsigned int>(t, 4)) &&
         (d.gmax = talloc<unsigned int>(t, 3 * (TRN_MAXL + 2))) &&
         (!wide || (d.gslots = talloc<unsigned int>(t, (size_t)(TRN_MAXL + 2) * TW_GSLOTS)));
    if (ok) ok = hipHostMalloc((void **)&t->hp_ring, (size_t)TRN_HP_SLOTS * 4 * sizeof(float)) == hipSuccess &&
                 hipEventCreateWithFlags(&t->ring_ev[0], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&t->ring_ev[1], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        azx_trn_destroy(t);
        return tfail(AZX_ENOMEM, "train: hipMalloc failed");
    }
    d.wpmax = talloc<float>(t, (size_t)(TRN_MAXL + 2) * ((C * C * 9 + 255) / 256));
    d.pstat = talloc<float2>(t, (size_t)(L + 1) * B * C);
    d.pgsum = talloc<float2>(t, (size_t)(L + 1) * B * C);
    if (!d.wpmax || !d.pstat || !d.pgsum) {
        azx_trn_destroy(t);
        return tfail(AZX_ENOMEM, "train: hipMalloc failed");
    }
    d.wmax = d.gmax + (TRN_MAXL + 2);
    d.fsc = talloc<float4>(t, TRN_MAXL + 2);
    d.embw1_old = talloc<float>(t, (size_t)12 + C * 36);
    if (!d.fsc || !d.embw1_old) {
        azx_trn_destroy(t);
        return tfail(AZX_ENOMEM, "train: hipMalloc failed");
    }
    d.board = t->in_board; d.legal = t->in_legal; d.prob = t->in_prob; d.reward = t->in_reward; d.hp = t->hp_dev; d.hp_ring = t->hp_ring;
    for (int l = 0; l <= L; ++l) {
        d.raw[l] = t->raw[l]; d.act[l] = t->act[l]; d.g[l] = t->g[l]; d.Wf[l] = t->Wf[l]; d.Wb[l] = t->Wb[l];
    }
    d.rawL = t->raw[L]; d.actL = t->act[L]; d.actLm2 = L >= 2 ? t->act[L - 2] : nullptr; d.gL = t->g[L];
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
    t->dbg["dlogit"] = {d.dlogit, (size_t)B * d.dl_stride * 4};
    // AZX_TRAIN_GRAPH=1: the step as one captured HIP graph.  Off by default: with the weight-gradient passes on their
    // own stream, plain launches run the step in 0.92 ms where the graph executor's placement of the two branches
    // takes 1.01 ms (the host needs ~0.3 ms to queue a step: it stays ahead either way).
    t->use_graph = getenv("AZX_TRAIN_GRAPH") && !strcmp(getenv("AZX_TRAIN_GRAPH"), "1");
    t->fork = !(getenv("AZX_TRAIN_FORK") && !strcmp(getenv("AZX_TRAIN_FORK"), "0"));
    if (hipDeviceSynchronize() != hipSuccess) {
        azx_trn_destroy(t);
        return tfail(AZX_EHIP, "train: device sync after the allocations failed");
    }
    *out = t;
    return AZX_OK;
}

void azx_trn_destroy(AzxTrain *t) {
    if (!t) return;
#ifdef AZX_TRN_STAMP
    {
        std::vector<unsigned long long> raw((size_t)3 * TS_WAVES * 8);
        if (hipMemcpyFromSymbol(raw.data(), HIP_SYMBOL(g_trn_stamp), raw.size() * 8) == hipSuccess)
            for (int r = 0; r < 3; ++r) {
                double h[8] = {0};
                for (int w = 0; w < TS_WAVES; ++w)
                    for (int k = 0; k < 8; ++k) h[k] += (double)raw[((size_t)r * TS_WAVES + w) * 8 + k];
                if (h[6] == 0) continue;
                static const char *nm[6] = {"coefficients / init + barrier", "staging + barrier", "k-loop (+ barrier in wgrad)", "epilogue / reduce + write",
                                            "(conv) the partial sums' reduction", "(conv) start -> partial sums arrived"};
                double tot = 0;
                for (int k = 0; k < 6; ++k) tot += h[k];
                fprintf(stderr, "k_trn_%s, last launch, %.0f waves: %.0f cycles/wave, %.2f us/wave wall, shader clock %.0f MHz\n",
                        r == 2 ? "wgrad" : r ? "conv<BWD>" : "conv<FWD>", h[6], tot / h[6], h[7] / h[6] / 100.0, 100.0 * tot / h[7]);
                for (int k = 0; k < 6; ++k) fprintf(stderr, "  %-40s %6.1f%%  %9.0f cycles/wave\n", nm[k], 100.0 * h[k] / tot, h[k] / h[6]);
            }
    }
#endif
    if (t->exec) (void)hipGraphExecDestroy(t->exec);
    if (t->graph) (void)hipGraphDestroy(t->graph);
    for (hipEvent_t e : t->events) (void)hipEventDestroy(e);
    if (t->cap) (void)hipStreamDestroy(t->cap);
    if (t->side) (void)hipStreamDestroy(t->side);
    for (void *p : t->allocs) (void)hipFree(p);
    if (t->hp_ring) (void)hipHostFree(t->hp_ring);
    for (hipEvent_t e : t->ring_ev) if (e) (void)hipEventDestroy(e);
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
        for (int l = 0; l <= L; ++l) { d.bn_w[l] = (const float *)bnw[l]->ptr; d.bn_b[l] = (const float *)bnb[l]->ptr; }
        d.bnwL = d.bn_w[L];
        d.bnbL = d.bn_b[L];
    }
    t->hoffs = {fc2w->goff, fc2b->goff, fc3w->goff, fc3b->goff, mfw->goff, mfb->goff};
    // update segments and the block table
    {
        // the tower filters' blocks first: their update (which is also the second stage of the filter-gradient
        // reduction, the step's largest read) is launched on its own behind the last filter-gradient kernel
        std::vector<Segment> segs;
        std::vector<int2> blocks, rest;
        for (Bound *b : params) {
            Segment s;
            memset(&s, 0, sizeof s);
            s.p = (float *)b->ptr; s.mom = b->mom; s.n = b->n; s.goff = b->goff; s.kind = SEG_PLAIN;
            for (int l = 1; l <= L; ++l) if (conv[l] == b) { s.layer = l; s.kind = SEG_CONV; }
            for (int i = 0; i < L + 3; ++i) {
                if (bnw[i] == b) {
                    s.layer = i; s.kind = SEG_BNW;
                    s.run_mean = (float *)rmean[i]->ptr; s.run_var = (float *)rvar[i]->ptr; s.tracked = (long long *)trk[i]->ptr;
                }
                if (bnb[i] == b) { s.layer = i; s.kind = SEG_BNB; }
            }
            if (b == vconv) { s.kind = SEG_HCONV; s.aux = 0; }
            if (b == pconv) { s.kind = SEG_HCONV; s.aux = 2 * C; }
            if (b == w1) s.kind = SEG_W1;
            if (b == emb) s.kind = SEG_EMB;
            if (s.kind == SEG_CONV) { t->conv_goff.resize(L + 1); t->conv_goff[s.layer] = b->goff; }
            const int si = (int)segs.size();
            segs.push_back(s);
            if (s.kind == SEG_CONV && t->wide) {          // no blocks in the table: one launch per layer, see k_tw_update_conv
                t->conv_seg.resize(L + 1);
                t->conv_seg[s.layer] = s;
                continue;
            }
            for (size_t e = 0; e < b->n; e += 256) (s.kind == SEG_CONV ? blocks : rest).push_back(make_int2(si, (int)(e / 256)));
        }
        t->n_conv_blocks = (int)blocks.size();
        blocks.insert(blocks.end(), rest.begin(), rest.end());
        t->segs = upload_table(t, segs);
        t->blocks = upload_table(t, blocks);
        t->n_blocks = (int)blocks.size();
        if (!t->segs || !t->blocks) return tfail(AZX_ENOMEM, "train: uploading the update tables failed");
    }
    for (int l = 1; l <= L; ++l) d.convw[l] = (const float *)conv[l]->ptr;
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
    constexpr int SMALL = TRN_SMALL_THREADS;
    hipLaunchKernelGGL(k_trn_prep<C>, dim3((C * C * 9 + 255) / 256, L + 1), dim3(256), 0, st, d);
    hipLaunchKernelGGL(k_trn_stem_fwd<C>, dim3(B), dim3(SMALL), 0, st, d);
    // (large batches: a layer's per-board pairs are summed once behind their producer, see k_trn_totals)
    auto totals = [&](const float2 *part, int l, int k) {
        if (d.presum) hipLaunchKernelGGL(k_trn_totals, dim3((C + 15) / 16), dim3(256), 0, st, part + (size_t)l * B * C, B, C, d.sums + (size_t)l * C * 4 + k);
    };
    totals(d.pstat, 0, 0);
    const size_t conv_lds = ((size_t)std::max(std::max((cells + 1) * (C + 4), cells * 36), 2048) + 8 * C + 264 + 1024) * sizeof(float);
    for (int l = 1; l <= L; ++l)
    {
        ConvPtrs q;
        const bool has_res = ((l - 1) & 1) == 0 && l - 1 >= 2;
        q.in0 = t->raw[l - 1]; q.in1 = has_res ? t->act[l - 3] : t->raw[l - 1];
        q.bnw = d.bn_w[l - 1]; q.bnb = d.bn_b[l - 1];
        q.w = t->fwd16 ? (const void *)d.Wf16[l] : (const void *)t->Wf[l];
        q.act_out = t->act[l - 1]; q.out = t->raw[l];
        q.e_act = q.e_raw = q.e_skip = nullptr;
        if (t->fwd16) hipLaunchKernelGGL((k_trn_conv<C, ROLE_FWD16>), dim3(NT, B), dim3(256), conv_lds, st, q, l, d);
        else hipLaunchKernelGGL((k_trn_conv<C, ROLE_FWD>), dim3(NT, B), dim3(256), conv_lds, st, q, l, d);
        totals(d.pstat, l, 0);
    }
    const size_t hc_lds = ((size_t)((cells * (C + 1) + 8 * C + 16 + 3) & ~3)) * sizeof(float) + (size_t)TRN_MID_THREADS * 16;
    hipLaunchKernelGGL(k_trn_heads_conv<C>, dim3(B), dim3(TRN_MID_THREADS), hc_lds, st, d);
    hipLaunchKernelGGL(k_trn_heads_fc, dim3(B), dim3(TRN_MID_THREADS), 0, st, d);
    size_t ev = 0;
    auto next_event = [&]() -> hipEvent_t {
        if (ev == t->events.size()) {
            hipEvent_t e;
            // (no system-scope fence: these events order two streams of ONE device; with the default flags every
            // record drains the caches to system scope between two dependent kernels of the data chain)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) return nullptr;
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
    const int hw_blocks = ((4 * cells + 255) / 256) * ((cells + 3) / 4) + 16 * ((2 * cells + 255) / 256) + (cells + 129 + 3) / 4;
    hipLaunchKernelGGL(k_trn_heads_wgrad, dim3(hw_blocks), dim3(256), 0, ws, d, t->hoffs);
    const size_t hb_lds = ((size_t)cells * (C + 1) + (size_t)SMALL * 8) * sizeof(float);
    hipLaunchKernelGGL(k_trn_heads_bwd<C>, dim3(B), dim3(SMALL), hb_lds, st, d);
    const int KP = (cells + 1) & ~1;
    const size_t wg_lds = ((size_t)KP * 32 + (size_t)(d.N + 2) * (d.N + 2) * 32 + 12 * 1024 + 5 * 32 + KP) * sizeof(float);

    const size_t wg16_lds = (size_t)(d.N * 16 + (d.N + 3) * 16) * 128 + (1024 + 5 * 32 + 2) * sizeof(float);
    for (int l = L; l >= 1; --l) {
        totals(d.pgsum, l, 2);
        // g_l and BN_l's sums are complete here: the weight gradient of layer l runs beside the data chain
        if (!fork_side()) return tfail(AZX_EHIP, "train: forking the weight-gradient stream failed");
        // (measured with these launches removed: 0.500 ms per step with the fork events, 0.488 without them -- an event
        // costs the data chain ~1 us -- against 0.549 as shipped and 0.681 with the filter gradients in line: two thirds
        // of their 205 us hide under the data chain)
        const WgradPtrs wq = {t->g[l], t->raw[l], t->act[l - 1], d.bn_w[l]};
        if (t->wgrad16) hipLaunchKernelGGL(k_trn_wgrad16<C>, dim3(NT * NT, G), dim3(256), wg16_lds, ws, wq, l, G, d);
        else hipLaunchKernelGGL(k_trn_wgrad<C>, dim3(NT * NT, G), dim3(256), wg_lds, ws, wq, l, G, d);
        ConvPtrs q;
        const bool has_skip = ((l - 1) & 1) == 0 && l + 1 <= L;
        q.in0 = t->g[l]; q.in1 = t->raw[l];
        q.bnw = d.bn_w[l]; q.bnb = d.bn_b[l];
        q.w = t->bwd16 ? (const void *)d.Wb16[l] : (const void *)t->Wb[l];
        q.act_out = nullptr; q.out = t->g[l - 1];
        q.e_act = t->act[l - 1]; q.e_raw = t->raw[l - 1]; q.e_skip = has_skip ? t->g[l + 1] : t->act[l - 1];
        if (t->bwd16) hipLaunchKernelGGL((k_trn_conv<C, ROLE_BWD16>), dim3(NT, B), dim3(256), conv_lds, st, q, l, d);
        else hipLaunchKernelGGL((k_trn_conv<C, ROLE_BWD>), dim3(NT, B), dim3(256), conv_lds, st, q, l, d);
    }
    const size_t sb_lds = ((size_t)cells * C + 5 * C) * sizeof(float) + (size_t)TRN_MID_THREADS * 16;
    totals(d.pgsum, 0, 2);
    hipLaunchKernelGGL(k_trn_stem_bwd<C>, dim3(B), dim3(TRN_MID_THREADS), sb_lds, st, d);
    auto join_side = [&]() -> bool {
        if (!fork) return true;
        hipEvent_t e = next_event();
        return e && hipEventRecord(e, side) == hipSuccess && hipStreamWaitEvent(st, e, 0) == hipSuccess;
    };
    // every gradient is complete once the filter-gradient stream has drained: the tower filters' update (113 MB of
    // partial copies to reduce) follows the last filter-gradient kernel on that stream, under the stem's backward, the
    // finalize and the other tensors' update on this one
    if (!join_side()) return tfail(AZX_EHIP, "train: joining the weight-gradient stream failed");
    if (t->n_conv_blocks > 0)
        hipLaunchKernelGGL(k_trn_update<C>, dim3(t->n_conv_blocks), dim3(256), 0, ws, d, (const Segment *)t->segs, (const int2 *)t->blocks, G);
    if (t->n_blocks > t->n_conv_blocks)
        hipLaunchKernelGGL(k_trn_update<C>, dim3(t->n_blocks - t->n_conv_blocks), dim3(256), 0, st, d, (const Segment *)t->segs,
                           (const int2 *)t->blocks + t->n_conv_blocks, G);
    if (!join_side()) return tfail(AZX_EHIP, "train: joining the weight-gradient stream failed");
    if (hipGetLastError() != hipSuccess) return tfail(AZX_EHIP, "train: a kernel of the step failed to launch");
    return AZX_OK;
}

// The filter gradient of a wide layer from the two IMAGES that exist anyway -- draw_l's (k_tw_bnbwd, kept per layer) and
// act_{l-1}'s (k_tw_bnact): trn_wgrad16_body's tiling, operand layout and k-loop (a (32 co, 32 ci) tile pair x a group
// of boards per block, K = positions through ds_read_b64_tr_b16, the nine taps owned by column), with the staging
// reduced to 16-byte copies -- no BatchNorm arithmetic, no per-board sums, two tensors instead of three.
struct TwWgrad { const unsigned short *dimg, *aimg; const float2 *bsc; const float4 *fsc; float *part; };
template <int C>
__global__ __launch_bounds__(256) void k_tw_wgrad(TwWgrad A, int N, int B, int G) {
    constexpr int NT = C / 32, ITER = (169 * 16 + 255) / 256;
    extern __shared__ __align__(16) float lds[];
    const int cells = N * N, KR = N * 16, BR = (N + 3) * 16;
    unsigned char *Dh = reinterpret_cast<unsigned char *>(lds), *Dl = Dh + (size_t)KR * 64;     // [KR][32] f16 draw hi / lo
    unsigned char *Bh = Dl + (size_t)KR * 64, *Bl = Bh + (size_t)BR * 64;                       // [BR][32] f16 input hi / lo
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware block order: consecutive workgroups go to the eight XCDs round-robin, each with its own 4 MB L2.  All
    // (C / 32)^2 tile pairs of a board group read the same boards' two images (344 KB a board at 256 channels), so a
    // group is given to ONE XCD (workgroup id mod 8 = group mod 8) and its pairs to consecutive slots there: the XCD's 64
    // resident blocks walk one group's boards together and an image comes from HBM once, not once per tile column.
    int pair = blockIdx.x, grp = blockIdx.y;
    if ((G & 7) == 0) {
        const int id = blockIdx.y * gridDim.x + blockIdx.x, xcd = id & 7, slot = id >> 3, np = gridDim.x;
        grp = 8 * (slot / np) + xcd;
        pair = slot % np;
    }
    const int tm = pair / NT, tn = pair % NT;
    const int total = cells * 16;
    // (the native vector type: an array of HIP's uint4 STRUCT stays in scratch memory, see DESIGN 8.4's compiler traps)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v[ITER];
    auto request = [&](int b) {
#pragma unroll
        for (int k = 0; k < ITER; ++k) {       // (clamped, unconditional: the staging skips the items beyond `total`)
            const int i = min(tid + 256 * k, total - 1), pos = i >> 4, q = i & 15;
            const unsigned short *src = ((q >> 3) ? A.aimg : A.dimg) + ((size_t)b * cells + pos) * (2 * C) + ((q >> 2) & 1) * C +
                                        ((q >> 3) ? tn : tm) * 32 + (q & 3) * 8;
            v[k] = *reinterpret_cast<const u32x4 *>(src);
        }
    };
    request(grp);
    const float unscale = 1.f / (A.bsc->x * A.fsc->x);
    {   // the operand images start zero: the rows x >= N of draw and the border of the input stay that way
        float4 *z = reinterpret_cast<float4 *>(lds);
        for (int i = tid; i < (KR + BR) * 128 / 16; i += 256) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const bool owner = wv < 3;
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    __syncthreads();
    const int frag_off = (8 * (lane >> 5) + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const int col_off = (17 + (owner ? wv - 1 : 0)) * 64 + frag_off;
    for (int b = grp; b < B; b += G) {
#pragma unroll
        for (int k = 0; k < ITER; ++k) {
            const int i = tid + 256 * k;
            if (i >= total) break;
            const int pos = i >> 4, q = i & 15, y = pos / N, krow = y * 16 + (pos - y * N);
            unsigned char *dst = (q >> 3) ? (((q >> 2) & 1) ? Bl : Bh) + (size_t)(krow + 17) * 64
                                          : (((q >> 2) & 1) ? Dl : Dh) + (size_t)krow * 64;
            *reinterpret_cast<u32x4 *>(dst + (q & 3) * 16) = v[k];
        }
        __syncthreads();
        if (b + G < B) request(b + G);                 // travels under this board's k-loop
        if (owner) {
            f16x8 dhi = tr_frag(Dh + frag_off), dlo = tr_frag(Dl + frag_off);
            f16x8 h0 = tr_frag(Bh + col_off - 16 * 64), l0 = tr_frag(Bl + col_off - 16 * 64);
            f16x8 h1 = tr_frag(Bh + col_off), l1 = tr_frag(Bl + col_off);
            f16x8 h2 = tr_frag(Bh + col_off + 16 * 64), l2 = tr_frag(Bl + col_off + 16 * 64);
            for (int s = 0; s < N; ++s) {
                const int sn = s + 1 < N ? s + 1 : s;
                const f16x8 ndhi = tr_frag(Dh + frag_off + (size_t)sn * 16 * 64), ndlo = tr_frag(Dl + frag_off + (size_t)sn * 16 * 64);
                const size_t r3 = (size_t)(s + 2 <= N ? s + 2 : N) * 16 * 64;
                const f16x8 h3 = tr_frag(Bh + col_off + r3), l3 = tr_frag(Bl + col_off + r3);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dhi, h0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dhi, h1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dhi, h2, acc[2], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dhi, l0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dhi, l1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dhi, l2, acc[2], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dlo, h0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dlo, h1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dlo, h2, acc[2], 0, 0, 0);
                dhi = ndhi; dlo = ndlo;
                h0 = h1; l0 = l1; h1 = h2; l1 = l2; h2 = h3; l2 = l3;
            }
        }
        __syncthreads();
    }
    float *part = A.part + (size_t)grp * ((size_t)C * C * 9);
    const int li = lane & 31, lh = lane >> 5, ci = tn * 32 + li;
    if (owner) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int t = 3 * u + wv;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = tm * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
                part[((size_t)t * C + co) * C + ci] = acc[u][i] * unscale;
            }
        }
    }
}

// k_tw_wgrad with TWO ci tiles per block (VERDICT r5 #4a): a block is (32 co, 64 ci) x a group of boards, 512 threads; draw's
// slice is staged ONCE for both ci tiles (92 KB of LDS: one block per CU).  Waves 0..5 own a (ci tile w / 3, tap column w % 3)
// unit with three accumulator tiles; waves w and w + 4 share a SIMD, so six whole units would load the SIMDs 26 / 26 / 13 / 13
// row-steps a board: waves 6 and 7 take the lower rows of waves 0 and 1's units (20 / 20 / 19 / 19) and hand their sums over
// through LDS at the end.  Same operand layout, k-loop and partial copies as k_tw_wgrad; (C / 32) x (C / 64) tile pairs per
// board group.  The default at 256 channels (AZX_TRAIN_WGRAD2=0 / 1 overrides; k_tw_wgrad stays the kernel of 64 and 128).
// Measured (19x256 on 13x13, batch 128; profiles/r6_train_wide_ab.txt 8, 11, 12): the kernel alone 86.9 vs 90.6 us with whole
// units, 84 with the rows balanced -- and the STEP on its two streams 9.6 vs 10.4 ms.  Not because it shares a SIMD with the
// backward convolution (2 x 176 + 216 registers exceed 512; a build in which they fit, 2 x 168 + 168, measures the same
// step): one 8-wave block per CU leaves room for the data chain's elementwise and update blocks and the two chains take
// turns per CU at block granularity.  The balanced rows gain 7 % alone and nothing in the step (the idle SIMD halves were
// already used by the other stream's blocks).  Capped at 168 registers it spills 12 bytes a lane: 9.89 vs 9.82 ms.
template <int C>
__global__ __launch_bounds__(512) void k_tw_wgrad2(TwWgrad A, int N, int B, int G) {
    constexpr int NT2 = C / 64, PPP = 24, ITER = (169 * PPP + 511) / 512;      // pieces per position: 8 draw + 2 x 8 input
    extern __shared__ __align__(16) float lds[];
    const int cells = N * N, KR = N * 16, BR = (N + 3) * 16;
    unsigned char *Dh = reinterpret_cast<unsigned char *>(lds), *Dl = Dh + (size_t)KR * 64;      // [KR][32] f16 draw hi / lo
    unsigned char *Bb = Dl + (size_t)KR * 64;                                                    // [ci tile][hi, lo][BR][32] f16 input
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int pair = blockIdx.x, grp = blockIdx.y;
    if ((G & 7) == 0) {       // a board group on ONE XCD (see k_tw_wgrad)
        const int id = blockIdx.y * gridDim.x + blockIdx.x, xcd = id & 7, slot = id >> 3, np = gridDim.x;
        grp = 8 * (slot / np) + xcd;
        pair = slot % np;
    }
    const int tm = pair / NT2, tn2 = pair % NT2;
    const int total = cells * PPP;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v[ITER];
    auto request = [&](int b) {
#pragma unroll
        for (int k = 0; k < ITER; ++k) {
            const int i = min(tid + 512 * k, total - 1), pos = i / PPP, q = i - pos * PPP;
            const bool inp = q >= 8;
            const int qq = inp ? q - 8 : q, j = qq >> 3, hl = (qq >> 2) & 1;          // (draw: j = 0)
            const unsigned short *src = (inp ? A.aimg : A.dimg) + ((size_t)b * cells + pos) * (2 * C) + hl * C +
                                        (inp ? tn2 * 64 + j * 32 : tm * 32) + (qq & 3) * 8;
            v[k] = *reinterpret_cast<const u32x4 *>(src);
        }
    };
    request(grp);
    const float unscale = 1.f / (A.bsc->x * A.fsc->x);
    {
        float4 *z = reinterpret_cast<float4 *>(lds);
        for (int i = tid; i < (2 * KR + 4 * BR) * 64 / 16; i += 512) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    // waves 0..5 own (ci tile, tap column) = (w / 3, w % 3); waves 6 and 7 take the LOWER rows of waves 0 and 1's units, so
    // that every SIMD (waves w, w + 4) has 19-20 row-steps of matrix work per board instead of 26 / 26 / 13 / 13
    const int unit = wv < 6 ? wv : wv - 6;
    const int oj = unit / 3, oc = unit % 3;       // this wave's ci tile and tap column
    const int s_lo = wv >= 6 ? (N + 1) / 2 : 0, s_hi = wv < 2 ? (N + 1) / 2 : N;
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    __syncthreads();
    const int frag_off = (8 * (lane >> 5) + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const unsigned char *Bh = Bb + (size_t)(2 * oj) * BR * 64, *Bl = Bh + (size_t)BR * 64;
    const int col_off = (17 + (oc - 1)) * 64 + frag_off;
    for (int b = grp; b < B; b += G) {
#pragma unroll
        for (int k = 0; k < ITER; ++k) {
            const int i = tid + 512 * k;
            if (i >= total) break;
            const int pos = i / PPP, q = i - pos * PPP, y = pos / N, krow = y * 16 + (pos - y * N);
            const bool inp = q >= 8;
            const int qq = inp ? q - 8 : q, j = qq >> 3, hl = (qq >> 2) & 1;
            unsigned char *dst = inp ? Bb + (size_t)(2 * j + hl) * BR * 64 + (size_t)(krow + 17) * 64
                                     : (hl ? Dl : Dh) + (size_t)krow * 64;
            *reinterpret_cast<u32x4 *>(dst + (qq & 3) * 16) = v[k];
        }
        __syncthreads();
        if (b + G < B) request(b + G);                 // travels under this board's k-loop
        {
            const size_t r0 = (size_t)s_lo * 16 * 64;
            f16x8 dhi = tr_frag(Dh + frag_off + r0), dlo = tr_frag(Dl + frag_off + r0);
            f16x8 h0 = tr_frag(Bh + col_off - 16 * 64 + r0), l0 = tr_frag(Bl + col_off - 16 * 64 + r0);
            f16x8 h1 = tr_frag(Bh + col_off + r0), l1 = tr_frag(Bl + col_off + r0);
            f16x8 h2 = tr_frag(Bh + col_off + 16 * 64 + r0), l2 = tr_frag(Bl + col_off + 16 * 64 + r0);
            for (int s = s_lo; s < s_hi; ++s) {
                const int sn = s + 1 < N ? s + 1 : s;
                const f16x8 ndhi = tr_frag(Dh + frag_off + (size_t)sn * 16 * 64), ndlo = tr_frag(Dl + frag_off + (size_t)sn * 16 * 64);
                const size_t r3 = (size_t)(s + 2 <= N ? s + 2 : N) * 16 * 64;
                const f16x8 h3 = tr_frag(Bh + col_off + r3), l3 = tr_frag(Bl + col_off + r3);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dhi, h0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dhi, h1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dhi, h2, acc[2], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dhi, l0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dhi, l1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dhi, l2, acc[2], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dlo, h0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dlo, h1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dlo, h2, acc[2], 0, 0, 0);
                dhi = ndhi; dlo = ndlo;
                h0 = h1; l0 = l1; h1 = h2; l1 = l2; h2 = h3; l2 = l3;
            }
        }
        __syncthreads();
    }
    float *part = A.part + (size_t)grp * ((size_t)C * C * 9);
    const int li = lane & 31, lh = lane >> 5, ci = tn2 * 64 + oj * 32 + li;
    // waves 6 and 7 hand their partial sums to waves 0 and 1 through LDS (the operand images are done with)
    float *hand = lds + (size_t)(wv & 1) * (48 * 64) + lane;
    if (wv >= 6) {
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) hand[(u * 16 + i) * 64] = acc[u][i];
    }
    __syncthreads();
    if (wv < 2) {
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[u][i] += hand[(u * 16 + i) * 64];
    }
    if (wv < 6) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int t = 3 * u + oc;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = tm * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
                part[((size_t)t * C + co) * C + ci] = acc[u][i] * unscale;
            }
        }
    }
}

// (Measured and dropped, round 6 -- the hypothesis: one board ahead, a request has ONE k-loop (117 MFMAs = 1.5 us) to come
// back in while all 64 tile pairs of a board group ask for the same board at once, so the first to ask pays an HBM round
// trip the k-loop does not cover.  (1) A second 44-register set, boards requested TWO ahead: the compiler takes 256
// registers and spills 136 bytes a lane, 120 vs 91 us.  (2) One dword of each piece of the board two ahead touched so
// that the real request hits L2, 11 registers: 156 vs 92 us (the touches sit in the same in-order vmcnt queue as the
// pieces the staging waits for).  profiles/r6_train_wide_ab.txt)
// (Measured and dropped, round 5: a 64 x 64 tile per block -- four waves = (co half, ci half), each with all nine taps in
// 144 accumulator registers, every SIMD with matrix work, the two 64-channel image slices staged once for four tile
// pairs, one block per CU: 80 vs 92 us per launch, but the blocks that fill the chip once need 16 board groups instead
// of 8 -- twice the partial copies for the update to reduce: 10.60 vs 10.45 ms per step at 19x256, 4.93 vs 4.25 at
// 19x128.  With one block per CU the LDS store phase and its barriers are exposed.)

// the wide step (C = 128 / 256): see "wide towers" above
template <int C>
static int enqueue_step_wide(AzxTrain *t, hipStream_t st, hipStream_t side, bool fork) {
    const TrnDev &d = t->d;
    const int L = d.L, B = d.B, cells = d.cells, G = t->G, N = d.N;
    constexpr int NT = C / 32, SMALL = TRN_SMALL_THREADS;
    const dim3 eg(C / 64, B), eb(256);
    hipLaunchKernelGGL(k_trn_prep<C>, dim3((C * C * 9 + 255) / 256, L + 1), dim3(256), 0, st, d);
    hipLaunchKernelGGL(k_tw_scales<C>, dim3(L + 1), dim3(256), 0, st, d, 0);
    hipLaunchKernelGGL(k_tw_scales<C>, dim3(1), dim3(256), 0, st, d, 1);
    hipLaunchKernelGGL(k_tw_pack<C>, dim3((C * (C / 8) + 255) / 256, L, 2), dim3(256), 0, st, d, t->Ww16f_dev, t->Ww16b_dev);
    hipLaunchKernelGGL(k_trn_stem_fwd<C>, dim3(B), dim3(SMALL), 0, st, d);
    auto totals = [&](const float2 *part, int l, int k) {         // large batches, see k_trn_totals
        if (d.presum) hipLaunchKernelGGL(k_trn_totals, dim3(C / 16), dim3(256), 0, st, part + (size_t)l * B * C, B, C, d.sums + (size_t)l * C * 4 + k);
    };
    for (int l = 0; l < L; ++l) {
        // act_l from raw_l (batch statistics from the per-board partials), then raw_{l+1} = conv(act_l)
        const bool has_res = (l & 1) == 0 && l >= 2;
        const TwAct a = {t->raw[l], has_res ? t->act[l - 2] : nullptr, d.bn_w[l], d.bn_b[l], d.pstat + (size_t)l * B * C,
                         d.sums + (size_t)l * C * 4, t->act[l], t->A16[l], d.fsc + (l + 1), d.presum, t->relu_mask[l]};
        totals(d.pstat, l, 0);
        hipLaunchKernelGGL(k_tw_bnact, eg, eb, 0, st, a, cells, C, B, d.invN);
        if (int rc = azx_net_wide_train_conv(N, C, t->Ww16f[l + 1], t->A16[l], t->raw[l + 1], B, &d.fsc[l + 1].y,
                                             d.pstat + (size_t)(l + 1) * B * C, st))
            return tfail(rc, "train: launching a wide forward convolution failed");
    }
    {   // act_L (and BN_L's totals), the head convolutions, the FC layers and the loss
        const TwAct a = {t->raw[L], L >= 2 ? t->act[L - 2] : nullptr, d.bn_w[L], d.bn_b[L], d.pstat + (size_t)L * B * C,
                         d.sums + (size_t)L * C * 4, t->act[L], nullptr, d.fsc, d.presum, nullptr};
        totals(d.pstat, L, 0);
        hipLaunchKernelGGL(k_tw_bnact, eg, eb, 0, st, a, cells, C, B, d.invN);
    }
    hipLaunchKernelGGL(k_tw_hconv, dim3(B), dim3(SMALL), 0, st, d);
    hipLaunchKernelGGL(k_tw_heads_fc, dim3(B), dim3(SMALL), 0, st, d);
    size_t ev = 0;
    auto next_event = [&]() -> hipEvent_t {
        if (ev == t->events.size()) {
            hipEvent_t e;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) return nullptr;
            t->events.push_back(e);
        }
        return t->events[ev++];
    };
    hipStream_t ws = fork ? side : st;
    auto fork_side = [&]() -> bool {
        if (!fork) return true;
        hipEvent_t e = next_event();
        return e && hipEventRecord(e, st) == hipSuccess && hipStreamWaitEvent(side, e, 0) == hipSuccess;
    };
    auto join_side = [&]() -> bool {
        if (!fork) return true;
        hipEvent_t e = next_event();
        return e && hipEventRecord(e, side) == hipSuccess && hipStreamWaitEvent(st, e, 0) == hipSuccess;
    };
    if (!fork_side()) return tfail(AZX_EHIP, "train: forking the weight-gradient stream failed");
    const int hw_blocks = ((4 * cells + 255) / 256) * ((cells + 3) / 4) + 16 * ((2 * cells + 255) / 256) + (cells + 129 + 3) / 4;
    hipLaunchKernelGGL(k_trn_heads_wgrad, dim3(hw_blocks), dim3(256), 0, ws, d, t->hoffs);
    hipLaunchKernelGGL(k_tw_heads_bwd, eg, eb, 0, st, d);
    const size_t wgi_lds = (size_t)(N * 16 + (N + 3) * 16) * 128;
    const size_t wg16_lds = (size_t)(N * 16 + (N + 3) * 16) * 128 + (1024 + 5 * 32 + 2) * sizeof(float);
    for (int l = L; l >= 1; --l) {
        // (sum g_l, sum g_l xhat_l) come from the partials of k_tw_heads_bwd (l = L) / the backward convolution's epilogue
        const TwBnBwd bb = {t->g[l], t->raw[l], d.bn_w[l], d.pgsum + (size_t)l * B * C, d.sums + (size_t)l * C * 4, d.gmax + l,
                            d.fsc + l, t->D16[l], d.bsc + l, d.presum, d.gslots + (size_t)l * TW_GSLOTS};
        totals(d.pgsum, l, 2);
        hipLaunchKernelGGL(k_tw_bnbwd, eg, eb, 0, st, bb, cells, C, B, d.invN);
        // draw_l's image is complete: the filter gradient of layer l runs beside the rest of the data chain
        if (!fork_side()) return tfail(AZX_EHIP, "train: forking the weight-gradient stream failed");
        if (t->wgrad16) {
            const TwWgrad wq = {t->D16[l], t->A16[l - 1], d.bsc + l, d.fsc + l, d.wpart + (size_t)(l - 1) * G * ((size_t)C * C * 9)};
            if (t->wgrad2) hipLaunchKernelGGL(k_tw_wgrad2<C>, dim3(NT * (NT / 2), G), dim3(512), (size_t)(2 * N * 16 + 4 * (N + 3) * 16) * 64, ws, wq, N, B, G);
            else hipLaunchKernelGGL(k_tw_wgrad<C>, dim3(NT * NT, G), dim3(256), wgi_lds, ws, wq, N, B, G);
        } else {      // AZX_TRAIN_WGRAD=fp32 here: the filter gradient from the fp32 tensors (k_trn_wgrad16, BatchNorm backward on the way in)
            const WgradPtrs wq = {t->g[l], t->raw[l], t->act[l - 1], d.bn_w[l]};
            hipLaunchKernelGGL(k_trn_wgrad16<C>, dim3(NT * NT, G), dim3(256), wg16_lds, ws, wq, l, G, d);
        }
        // (the update right behind its filter gradient on that stream: all 38 updates after the last gradient, over both
        // streams, measured 10.15 vs 9.94 ms a step)
        hipLaunchKernelGGL(k_tw_update_conv, dim3(C * C / 256), dim3(256), 0, ws, d, t->conv_seg[l], G);
        const bool has_skip = ((l - 1) & 1) == 0 && l + 1 <= L;
        // conv^T with the ReLU mask, the skip gradient, g_{l-1}'s per-board (sum g, sum g xhat) and max |g| in its epilogue
        // (as a separate elementwise launch behind a plain conv^T: 72 + 23 us per layer against 92 fused, 10.74 vs 10.68 ms
        // per step -- the fused form saves a tensor's round trip, not time)
        if (int rc = azx_net_wide_train_conv_bwd(N, C, t->Ww16b[l], t->D16[l], t->g[l - 1], B, &d.bsc[l].y,
                                                 d.pgsum + (size_t)(l - 1) * B * C, t->relu_mask[l - 1], t->raw[l - 1],
                                                 has_skip ? t->g[l + 1] : nullptr, d.sums + (size_t)(l - 1) * C * 4, d.invN,
                                                 d.gslots + (size_t)(l - 1) * TW_GSLOTS, st))
            return tfail(rc, "train: launching a wide backward convolution failed");
    }
    totals(d.pgsum, 0, 2);
    hipLaunchKernelGGL(k_tw_stem_bwd, eg, eb, 0, st, d);
    if (!join_side()) return tfail(AZX_EHIP, "train: joining the weight-gradient stream failed");
    if (t->n_conv_blocks > 0)
        hipLaunchKernelGGL(k_trn_update<C>, dim3(t->n_conv_blocks), dim3(256), 0, ws, d, (const Segment *)t->segs, (const int2 *)t->blocks, G);
    if (t->n_blocks > t->n_conv_blocks)
        hipLaunchKernelGGL(k_trn_update<C>, dim3(t->n_blocks - t->n_conv_blocks), dim3(256), 0, st, d, (const Segment *)t->segs,
                           (const int2 *)t->blocks + t->n_conv_blocks, G);
    if (!join_side()) return tfail(AZX_EHIP, "train: joining the weight-gradient stream failed");
    if (hipGetLastError() != hipSuccess) return tfail(AZX_EHIP, "train: a kernel of the step failed to launch");
    return AZX_OK;
}

// (what each kernel asks for, not a blanket cap: static + dynamic LDS together must stay within the CU's 160 KB)
template <int C>
static int raise_limits_wide(int cells, int N) {
    (void)cells;
    const size_t need[3] = {(size_t)(N * 16 + (N + 3) * 16) * 128 + (1024 + 5 * 32 + 2) * sizeof(float),
                            (size_t)(N * 16 + (N + 3) * 16) * 128, (size_t)(2 * N * 16 + 4 * (N + 3) * 16) * 64};
    const void *f[3] = {(const void *)k_trn_wgrad16<C>, (const void *)k_tw_wgrad<C>, (const void *)k_tw_wgrad2<C>};
    for (int i = 0; i < 3; ++i)
        if (need[i] > 48 * 1024 && hipFuncSetAttribute(f[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)need[i]) != hipSuccess)
            return tfail(AZX_EHIP, "train: raising a kernel's dynamic LDS limit failed (kernel " + std::to_string(i) + ", " +
                                   std::to_string(need[i]) + " bytes)");
    return AZX_OK;
}

static int raise_limits(int C, int cells = 0, int N = 0) {
    if (C == 128) return raise_limits_wide<128>(cells, N);
    if (C == 256) return raise_limits_wide<256>(cells, N);
    if (C == 64 && N > 11) return raise_limits_wide<64>(cells, N);
    const int cap = 128 * 1024;     // the largest user (k_trn_wgrad<64>) takes 94 KB; some kernels add static LDS on top
    const void *f64[] = {(const void *)k_trn_conv<64, ROLE_FWD>, (const void *)k_trn_conv<64, ROLE_FWD16>, (const void *)k_trn_conv<64, ROLE_BWD>, (const void *)k_trn_conv<64, ROLE_BWD16>, (const void *)k_trn_wgrad<64>, (const void *)k_trn_wgrad16<64>,
                         (const void *)k_trn_heads_conv<64>, (const void *)k_trn_stem_bwd<64>, (const void *)k_trn_heads_bwd<64>};
    const void *f32[] = {(const void *)k_trn_conv<32, ROLE_FWD>, (const void *)k_trn_conv<32, ROLE_FWD16>, (const void *)k_trn_conv<32, ROLE_BWD>, (const void *)k_trn_conv<32, ROLE_BWD16>, (const void *)k_trn_wgrad<32>, (const void *)k_trn_wgrad16<32>,
                         (const void *)k_trn_heads_conv<32>, (const void *)k_trn_stem_bwd<32>, (const void *)k_trn_heads_bwd<32>};
    const void *f16[] = {(const void *)k_trn_conv<16, ROLE_FWD>, (const void *)k_trn_conv<16, ROLE_FWD16>, (const void *)k_trn_conv<16, ROLE_BWD>, (const void *)k_trn_conv<16, ROLE_BWD16>, (const void *)k_trn_wgrad<16>, (const void *)k_trn_wgrad16<16>,
                         (const void *)k_trn_heads_conv<16>, (const void *)k_trn_stem_bwd<16>, (const void *)k_trn_heads_bwd<16>};
    const void **f = C == 64 ? f64 : C == 32 ? f32 : f16;
    for (int i = 0; i < 9; ++i)
        if (hipFuncSetAttribute(f[i], hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess)
            return tfail(AZX_EHIP, "train: raising a kernel's dynamic LDS limit failed");
    return AZX_OK;
}

static int enqueue_any(AzxTrain *t, hipStream_t st, hipStream_t side, bool fork) {
    switch (t->d.C) {
        case 256: return enqueue_step_wide<256>(t, st, side, fork);
        case 128: return enqueue_step_wide<128>(t, st, side, fork);
        case 64: return t->wide ? enqueue_step_wide<64>(t, st, side, fork) : enqueue_step<64>(t, st, side, fork);
        case 32: return enqueue_step<32>(t, st, side, fork);
        default: return enqueue_step<16>(t, st, side, fork);
    }
}

// after a step has been queued: count it, and every half ring leave a mark the host can wait on before it re-uses
// the slots (see azx_trn_step)
static int step_done(AzxTrain *t, hipStream_t st) {
    const unsigned long long sidx = t->host_steps % TRN_HP_SLOTS;
    t->host_steps += 1;
    if ((sidx + 1) % (TRN_HP_SLOTS / 2) == 0) {
        // steps [.., sidx] are queued: when this mark fires, the half of the ring ending at sidx is free again
        const int which = (int)(((sidx + 1) / (TRN_HP_SLOTS / 2)) & 1);
        if (hipEventRecord(t->ring_ev[which], st) != hipSuccess) return tfail(AZX_EHIP, "train: marking the hyper-parameter ring failed");
    }
    return AZX_OK;
}

int azx_trn_step(AzxTrain *t, float lr, float momentum, float weight_decay, hipStream_t st) {
    if (!t->is_bound) return tfail(AZX_ESTATE, "train: azx_train_bind has not been called");
    // a step that failed half-way may have run its first kernel, which advances the device's step count, without the
    // host's having moved: every later step would read another step's hyper-parameter slot.  Refuse instead.
    if (t->broken) return tfail(AZX_ESTATE, "train: an earlier step failed while it was being queued; create a new trainer");
    if (!t->cap) {
        if (int rc = raise_limits(t->d.C, t->d.cells, t->d.N)) return rc;
        // the side stream (filter gradients) takes the priority of the stream the first step is queued on: a trainer that
        // runs beside self-play on a high-priority stream (azalea_amd/play_ahead.py) gets both of its streams ahead
        int prio = 0;
        if (hipStreamGetPriority(st, &prio) != hipSuccess) prio = 0;
        if (hipStreamCreateWithFlags(&t->cap, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithPriority(&t->side, hipStreamNonBlocking, prio) != hipSuccess)
            return tfail(AZX_EHIP, "train: creating the capture streams failed");
    }
    // The hyper-parameters travel through a ring in pinned host memory that the step's first kernel reads (slot = steps
    // run so far, counted on both sides): one captured graph serves every learning rate and no extra node carries
    // them.  The host may queue at most TRN_HP_SLOTS steps ahead: every half ring it waits for the mark set half a
    // ring ago.
    {
        const unsigned long long sidx = t->host_steps % TRN_HP_SLOTS;
        // about to re-use the half ring starting at sidx: its previous occupants are the steps that ended with the mark
        // recorded half a ring BEFORE the latest one, i.e. the other event (the latest mark is the step just queued)
        if (t->host_steps >= TRN_HP_SLOTS && sidx % (TRN_HP_SLOTS / 2) == 0)
            if (hipEventSynchronize(t->ring_ev[((sidx / (TRN_HP_SLOTS / 2)) & 1) ^ 1]) != hipSuccess)
                return tfail(AZX_EHIP, "train: waiting for the hyper-parameter ring failed");
        float *slot = t->hp_ring + sidx * 4;
        slot[0] = lr; slot[1] = momentum; slot[2] = weight_decay; slot[3] = 0.f;
    }
    if (!t->use_graph) {
        // plain launches: the weight-gradient passes fork onto the side stream after an event on `st`
        if (int rc = enqueue_any(t, st, t->side, t->fork)) { t->broken = true; return rc; }
        return step_done(t, st);
    }
    if (!t->exec) {
        if (hipStreamBeginCapture(t->cap, hipStreamCaptureModeRelaxed) != hipSuccess)
            return tfail(AZX_EHIP, "train: hipStreamBeginCapture failed");
        int rc = enqueue_any(t, t->cap, t->side, t->fork);
        hipGraph_t graph = nullptr;
        const hipError_t ce = hipStreamEndCapture(t->cap, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); t->broken = true; return rc; }
        if (ce != hipSuccess || !graph) return tfail(AZX_EHIP, std::string("train: hipStreamEndCapture failed: ") + hipGetErrorString(ce));
        t->graph = graph;
        if (hipGraphInstantiate(&t->exec, graph, nullptr, nullptr, 0) != hipSuccess)
            return tfail(AZX_EHIP, "train: hipGraphInstantiate failed");
    }
    if (hipGraphLaunch(t->exec, st) != hipSuccess) { t->broken = true; return tfail(AZX_EHIP, "train: hipGraphLaunch failed"); }
    return step_done(t, st);
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
