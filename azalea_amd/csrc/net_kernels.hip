// net_kernels.hip -- HexNetwork inference forward for gfx950 (MI355X), hand-written MFMA.
//
// Restates azalea/network.py:17-85 (Network/Resblock forward, eval-mode BatchNorm) and
// :120-152 (HexNetwork: embedding, policy FC, legal-move gather + log_softmax).
//
// k_tower_mfma: the whole residual tower for a tile of boards in ONE launch.  Activations of
// the tile stay in LDS ([pos][C] fp32, row stride C+4 floats so ds_read_b128 A-fragment reads
// are bank-conflict free); each 3x3 convolution is an implicit GEMM (M = board positions padded
// to 32-row tiles, N = C_out, K = 9 taps x C_in) on v_mfma_f32_32x32x2_f32 (exact fp32, the
// 1e-4 logit tolerance rules out bf16); folded-BN bias, residual add and ReLU run in the MFMA
// epilogue straight back into LDS.  Weights are pre-packed on the host into the B-fragment
// order so every lane fetches its 4 k-steps with one coalesced 16-byte load from L2.
// k_heads: value/policy heads + masked softmax, one block per board.
// k_*_generic: plain VALU fallback for channel counts the MFMA tiling does not cover.
#include "net_priv.h"

// Diagnostic builds only (-DAZX_NET_ABLATE=bits): time the f16x3 tower without its A-fragment LDS
// reads (1), MFMAs (2) or weight loads (4).  Outputs are garbage; the shipped build uses 0.
#ifndef AZX_NET_ABLATE
#define AZX_NET_ABLATE 0
#endif
// -DAZX_SAT_TRACK=0: the split-f16 epilogues without the activation range tracking (A/B builds only)
#ifndef AZX_SAT_TRACK
#define AZX_SAT_TRACK 1
#endif

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <type_traits>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

static thread_local std::string g_net_err;
const char *azx_net_error() { return g_net_err.c_str(); }

int azx_net_fail(int code, const char *msg) {
    g_net_err = msg;
    return code;
}
static int nfail(int code, const char *msg) { return azx_net_fail(code, msg); }


// ============================================================================================
// fused residual tower on MFMA
//   C      channels (multiple of 32)
//   MT     32-row M tiles per board (ncells <= 32*MT)
//   BPB    boards per 256-thread block;  waves per board = 4/BPB = MS*NS
//   MS/NS  how a board's (MT x C/32) output tiles are split over its waves
// ============================================================================================
template <int C, int MT, int BPB, int MS, int NS>
__global__ __launch_bounds__(256) void k_tower_mfma(NetDev P, const uint8_t *__restrict__ ev_board,
                                                     const int32_t *__restrict__ n_eval_ptr,
                                                     int n_eval_host, float *__restrict__ act_out) {
    constexpr int NT = C / 32;
    constexpr int MTW = MT / MS, NTW = NT / NS;
    constexpr int LDW = C + 4;                      // LDS row stride in floats
    static_assert(MS * NS * BPB == 4, "4 waves per block");
    static_assert(MT % MS == 0 && NT % NS == 0, "tile split");
    extern __shared__ __align__(16) float lds[];
    const int n_eval = n_eval_ptr ? *n_eval_ptr : n_eval_host;
    const int e0 = blockIdx.x * BPB;
    if (e0 >= n_eval) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = P.N, ncells = P.ncells;
    const int rows = ncells + 1;                    // + one all-zero row (padding / tile tail)
    const int wb = wave / (MS * NS);                // board of this wave within the block
    const int part = wave % (MS * NS);
    const int ms = part / NS, ns = part % NS;
    float *X = lds + (size_t)wb * 2 * rows * LDW;   // block input / residual / block output
    float *Y = X + (size_t)rows * LDW;              // conv1 output

    // ---- stem: table lookups (embedding o conv3x3 o BN) + ReLU into X ------------------------
    for (int b = 0; b < BPB; ++b) {
        float *Xb = lds + (size_t)b * 2 * rows * LDW;
        const int e = e0 + b;
        const uint8_t *bd = ev_board + (size_t)min(e, n_eval - 1) * AZX_CELL_STRIDE;
        for (int idx = tid; idx < ncells * C; idx += 256) {
            const int pos = idx / C, co = idx - pos * C;
            const int y = pos / N, x = pos - y * N;
            float acc = P.stem_b[co];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
                if (yy >= 0 && yy < N && xx >= 0 && xx < N)
                    acc += P.stemT[(tap * 3 + bd[yy * N + xx]) * C + co];
            }
            Xb[pos * LDW + co] = fmaxf(acc, 0.0f);
        }
        for (int c = tid; c < LDW; c += 256) {      // zero rows
            Xb[ncells * LDW + c] = 0.0f;
            Xb[(size_t)rows * LDW + ncells * LDW + c] = 0.0f;
        }
    }
    __syncthreads();

    // per-lane geometry of the A fragment rows: lane (i = lane&31, h = lane>>5)
    const int li = lane & 31, lh = lane >> 5;
    int ry[MTW], rx[MTW];
    bool rvalid[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        const int r = (ms * MTW + m) * 32 + li;
        rvalid[m] = r < ncells;
        ry[m] = r / N;
        rx[m] = r - ry[m] * N;
    }

    for (int layer = 0; layer < P.layers; ++layer) {
        const float *src = (layer & 1) ? Y : X;
        float *dst = (layer & 1) ? X : Y;
        f32x16 acc[MTW][NTW];
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int n = 0; n < NTW; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

        const float4 *wl = reinterpret_cast<const float4 *>(P.Wp) +
                           (size_t)layer * 9 * (C / 8) * NT * 64;
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            int aoff[MTW];
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const int yy = ry[m] + dy, xx = rx[m] + dx;
                const bool ok = rvalid[m] && yy >= 0 && yy < N && xx >= 0 && xx < N;
                aoff[m] = (ok ? (yy * N + xx) : ncells) * LDW + 4 * lh;
            }
            const float4 *wt = wl + (size_t)tap * (C / 8) * NT * 64;
#pragma unroll 2
            for (int q = 0; q < C / 8; ++q) {
                float4 bfrag[NTW];
#pragma unroll
                for (int n = 0; n < NTW; ++n)
                    bfrag[n] = wt[((size_t)q * NT + (ns * NTW + n)) * 64 + lane];
                float4 afrag[MTW];
#pragma unroll
                for (int m = 0; m < MTW; ++m)
                    afrag[m] = *reinterpret_cast<const float4 *>(src + aoff[m] + 8 * q);
#pragma unroll
                for (int m = 0; m < MTW; ++m)
#pragma unroll
                    for (int n = 0; n < NTW; ++n) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[m].x, bfrag[n].x, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[m].y, bfrag[n].y, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[m].z, bfrag[n].z, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[m].w, bfrag[n].w, acc[m][n], 0, 0, 0);
                    }
            }
        }
        // ---- epilogue: + folded-BN bias (+ residual) -> ReLU -> LDS -----------------------
        // C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
        const bool residual = (layer & 1) != 0;       // conv2 of a Resblock: y += x (network.py:37)
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            const int co = (ns * NTW + n) * 32 + li;
            const float bb = P.bias[layer * C + co];
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (ms * MTW + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row < ncells) {
                        float v = acc[m][n][r] + bb;
                        if (residual) v += dst[row * LDW + co];
                        dst[row * LDW + co] = fmaxf(v, 0.0f);
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- tower output -> HBM [e][ncells][C] (the heads kernel consumes it) -------------------
    for (int b = 0; b < BPB; ++b) {
        const int e = e0 + b;
        if (e >= n_eval) break;
        const float *Xb = lds + (size_t)b * 2 * rows * LDW;   // layers is even: result is in X
        float4 *out = reinterpret_cast<float4 *>(act_out + (size_t)e * ncells * C);
        for (int idx = tid; idx < ncells * (C / 4); idx += 256) {
            const int pos = idx / (C / 4), c4 = idx - pos * (C / 4);
            out[idx] = *reinterpret_cast<const float4 *>(Xb + pos * LDW + 4 * c4);
        }
    }
}


// ============================================================================================
// fused residual tower on f16 MFMA with a 2-term split ("f16x3"): every fp32 operand x is carried
// as hi = f16(x), lo = f16(x - hi) (22 significant bits) and a product sum is accumulated in
// fp32 as  hi*hi + hi*lo + lo*hi  -- three v_mfma_f32_32x32x16_f16 at 16x the fp32-MFMA rate,
// i.e. ~5x the exact-fp32 kernel above at fp32-class accuracy (the dropped lo*lo term is 2^-22
// relative).  C = 64, N <= 11.
//   * two waves own one board (32 output channels each): the layer's whole output lives in their
//     accumulators (4 tiles of 32x32 each), so a layer overwrites its input in place and the
//     block input (residual) is simply kept in registers -- one LDS activation buffer per board;
//   * 4 boards per 512-thread block (2 waves per SIMD); LDS: 4 x 121 rows x 272 B (128 B hi | 128 B lo | 16 B pad:
//     ds_read_b128 A fragments conflict-free) + one shared zero row + 3 x 8 KB weight stages;
//   * weights stream global -> registers -> LDS in 8 KB stages (tap x 32 input channels), triple
//     buffered (published two stages ahead), one barrier per stage; fragments of the next k-step
//     (also across the stage boundary) are in flight while the current 12 MFMAs issue.
// ============================================================================================
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// DIAGNOSTIC, off by default (AZX_LO_BITS = 10 compiles to nothing).  Operand sparsity buys clock: the tower is
// power-limited (1 280 W), and zeroing low mantissa bits of the lo halves -- the operands of two of the
// three products, hi*lo and lo*hi, 2^-11 of the result -- lowers the multipliers' switching, so the power manager
// raises the clock: with lo rounded to 7 / 6 / 5 / 4 mantissa bits the same launch runs 1.2 / 2.0 / 3.0 / 4.1 %
// faster (all of lo zero: 11 %).  NOT shipped: on the reference's trained checkpoint (golden G8, peaked policies) the
// full-precision error is already 3.8e-5 of the 1e-4 tolerance and 7 bits gives 2.0e-4 (seeded nets G3 / G5r: 1e-6
// -> 5e-6), tools/net_err_lib.py.  AZX_LO_BITS = explicit mantissa bits kept in every lo half, activations
// (split2_f16) and weights (f16bits at pack time) alike, rounded to nearest.
__device__ __forceinline__ void split_f16(float v, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}
// The same split for a PAIR of values in four instructions instead of eight: one packed conversion for the two hi
// halves, v - hi as a mixed-precision FMA that reads its f16 operand straight out of the packed pair
// (v_fma_mix_f32: exact, the difference is representable), one packed conversion for the lo halves.  The
// compiler's own code for split_f16 converts every hi twice (once alone to subtract it, once packed to store it)
// and does not form the mixed FMA.  Bit-identical to split_f16 (a microbenchmark over 2^21 operands incl. zeros,
// denormal-range and f16-overflow-edge values: 0 mismatches).
__device__ __forceinline__ void split2_f16(float a, float b, uint32_t &hpk, uint32_t &lpk) {
    float la, lb;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hpk) : "v"(a), "v"(b));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(la) : "v"(hpk), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hpk), "v"(b));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lpk) : "v"(la), "v"(lb));
#if AZX_LO_BITS < 10
    lpk = (lpk + LO_RND * 0x10001u) & (LO_MASK * 0x10001u);     // round the lo halves to AZX_LO_BITS mantissa bits
#endif
}

// Diagnostic build only (-DAZX_NET_STAMP): per-region s_memtime sums of k_tower_f16x3_s16 (wave 0 of
// every block), printed by azx_net_destroy.  The shipped kernel executes no stamp.
#ifdef AZX_NET_STAMP
__device__ unsigned long long g_tower_stamp[10];   // 7 regions, block count, s_memrealtime ticks (100 MHz)
#define NT_DECL unsigned long long nt_last = __builtin_amdgcn_s_memtime(), nt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long nt_rt0 = __builtin_amdgcn_s_memrealtime();
__device__ unsigned long long g_tower_trace[3 * 32768];   // per block of the last launch: cu key, start, end (100 MHz)
#define NT_MARK(r) { const unsigned long long nt_now = __builtin_amdgcn_s_memtime(); nt_acc[r] += nt_now - nt_last; nt_last = nt_now; }
#define NT_FLUSH if (tid == 0) { for (int r_ = 0; r_ < 7; ++r_) atomicAdd(&g_tower_stamp[r_], nt_acc[r_]); atomicAdd(&g_tower_stamp[7], 1ull); atomicAdd(&g_tower_stamp[8], __builtin_amdgcn_s_memrealtime() - nt_rt0); \
    if (blockIdx.x < 32768) { const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xc = __builtin_amdgcn_s_getreg((31 << 11) | 20); \
        g_tower_trace[3 * blockIdx.x] = ((unsigned long long)(xc & 0xf) << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf); \
        g_tower_trace[3 * blockIdx.x + 1] = nt_rt0; g_tower_trace[3 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime(); } }
// the wide conv kernel's regions (every wave's lane 0): 0 prologue, 1 staging + barriers, 2 k-loop, 3 epilogue
__device__ unsigned long long g_wide_stamp[10];
#define WT_FLUSH if (lane == 0) { for (int r_ = 0; r_ < 7; ++r_) atomicAdd(&g_wide_stamp[r_], nt_acc[r_]); atomicAdd(&g_wide_stamp[7], 1ull); \
    atomicAdd(&g_wide_stamp[8], __builtin_amdgcn_s_memrealtime() - nt_rt0); }
#else
#define NT_DECL
#define NT_MARK(r)
#define NT_FLUSH
#define WT_FLUSH
#endif

#ifndef F16X3_BPB
#define F16X3_BPB 2   // boards per block: 2 -> 70 KB LDS, two blocks per CU overlap each other's prologue/epilogue
#endif
// ============================================================================================
// k_tower_f16x3_s16: the fused split-f16 tower on v_mfma_f32_16x16x32_f16.  Dense f16 MFMA on this
// chip is power-limited: a saturated 32x32x16 stream holds ~1.5-1.6 GHz (~1.4-1.5 PFLOP/s on random
// data), the 16x16x32 shape ~1.8 GHz (~1.75 PFLOP/s) -- tools/microbench/mfma_shapes.hip -- so the
// same products are issued as 16x16 tiles.  Block layout: 2 boards per
// 256-thread block, a wave owns 64 positions x 64 channels = 4 x 4 tiles of 16 x 16); the LDS image has
// the same 272-byte rows, with the chunks of a row and the rows of a tile ordered for the lane groups of
// ds_read_b128 (see lrow / lchunk below).  A k-step is one tap x 32 input channels (18 per layer): 8 weight fragments (4 channel
// tiles x hi/lo, double-buffered, from L2) + 8 activation fragments (4 position tiles x hi/lo,
// from LDS, single-buffered: tile m's registers are reloaded for the next k-step as soon as its
// 12 MFMAs have issued) feed 48 MFMAs, one load in each of the first MFMAs' shadows.
// ============================================================================================
#ifndef AZX_S16_FENCE
#define AZX_S16_FENCE 1   // scheduling fence after every third MFMA (measured best of none / 3rd / every)
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(F16X3_BPB * 128, 2) void k_tower_f16x3_s16(NetDev P, const uint8_t *__restrict__ ev_board,
                                                                        const int32_t *__restrict__ n_eval_ptr,
                                                                        int n_eval_host, float *__restrict__ act_out,
                                                                        float *__restrict__ hfeat) {
    constexpr int C = 64, ROWB = 272, MT = 4, NT = 4;     // 16-row position tiles / 16-channel tiles per wave
    extern __shared__ __align__(16) unsigned char smem[];
    const int n_eval = n_eval_ptr ? *n_eval_ptr : n_eval_host;
    const int e0 = blockIdx.x * F16X3_BPB;
    if (e0 >= n_eval) return;
    // the wave index is uniform over a wave: as a scalar, everything derived from it (channel / position base, the
    // 64-bit part of the weight-fragment addresses) is computed on the scalar unit instead of per lane
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    NT_DECL
    const int wb = wave >> 1, wh = wave & 1;             // board within the block, which half of its positions
    const int N = P.N, ncells = P.ncells;
    const int e = e0 + wb;
    const bool live = e < n_eval;
    const int board_b = 128 * ROWB;
    // the shared zeros come first, two rows of them: a padding tap reads at (its own address mod 256), i.e. on the
    // bank slot it would have used, so the padded lanes of a read do not collide with the others' slots
    const int zero_off = 0;
    const int x_off = 2 * ROWB + wb * board_b;
    unsigned char *X = smem + x_off;
    // lane (i = lane & 15: position inside a tile, h = lane >> 4: k-group of the operands / channel
    // quad of the result).  Transposed product: D[channel 4h + reg][position i].
    const int li = lane & 15, lh = lane >> 4;
    // LDS banking of ds_read_b128 (MI355X_MICROARCH, LDS): a wave's read is served in four groups of 16 lanes that are
    // NOT the four k-groups -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32 -- so a group mixes columns {0-3, 12-15}
    // of k-group g with columns {4-11} of k-group g + 1.  With rows 17 slots of 16 B apart (slot = row + chunk mod 16)
    // and a row's chunks in channel order, column 12 of g met column 11 of g + 1 on one slot in every group: 4 extra
    // cycles on each 4-cycle read (SQ_LDS_BANK_CONFLICT = 4.05 per LDS instruction, profiles/r3_resnet_pmc_counters).
    // Conflict-free for every tap shift: k-groups g and g ^ 1 sit 8 slots (128 B) apart in the row, and the columns
    // {4-11} are the rows {0-3, 8-11} of the tile -- a set that + 8 maps onto itself, like its complement.
    //   row of a tile column:  S16_ROW(li) = li ^ 4 for li < 8, li otherwise
    //   chunk (part p, k-half kh, k-group g) of a row at byte 128 (g & 1) + 16 (4 p + 2 kh + (g >> 1))
    const int lrow = li < 8 ? li ^ 4 : li;
    const int lchunk = 128 * (lh & 1) + 16 * (lh >> 1);

    unsigned long long tapok = 0ull;                     // bit tap*4 + m
    int rbase[MT], ry_[MT], rx_[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int r = 64 * wh + 16 * m + lrow;
        const int ry = r / N, rx = r - ry * N;
        ry_[m] = ry;
        rx_[m] = rx;
        rbase[m] = x_off + r * ROWB + lchunk;            // 8 channels (16 B) per k-group
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = ry + tap / 3 - 1, xx = rx + tap % 3 - 1;
            if (r < ncells && yy >= 0 && yy < N && xx >= 0 && xx < N) tapok |= 1ull << (tap * 4 + m);
        }
    }
    // The 36 (tap, tile) fragment offsets are invariant over the layers; hoisted out of the block loop they
    // would hold 36 VGPRs beside accumulators, residual and fragments (the kernel then spilled 25 registers
    // to scratch: 490 MB of HBM writes per launch).  The mask words and row bases pass through an opaque asm
    // at the top of every layer, so an offset is formed where it is used (bit-field extract, add, and: three
    // VALU in an MFMA's shadow; a padding tap reads the zero row at LDS offset 0) and lives for its two
    // k-steps only.
    uint32_t tapok_lo = (uint32_t)tapok, tapok_hi = (uint32_t)(tapok >> 32);
    auto act_offset = [&](int tap, int m) -> int {
        const int delta = ((tap / 3 - 1) * N + (tap % 3 - 1)) * ROWB;
        const uint32_t word = tap < 8 ? tapok_lo : tapok_hi;
        const int mask = (int)(word << (31 - ((tap * 4 + m) & 31))) >> 31;     // v_bfe_i32: 0 or -1
        return (mask | 0xF0) & (rbase[m] + delta);
    };

    f32x4 res[MT][NT];
    // the layer's folded-BN bias of this lane's 16 channels: requested BEFORE the barrier that ends the k-loop (the
    // weight fragments' registers are free by then), so its L2 round trip passes while the wave waits for its
    // partner instead of at the top of every epilogue
    auto load_bias = [&](const float *bias, float4 (&b4)[NT]) {
#pragma unroll
        for (int n = 0; n < NT; ++n) b4[n] = *reinterpret_cast<const float4 *>(bias + 16 * n + 4 * lh);
    };
    float satmax = 0.f;                              // largest activation this lane has split into hi + lo halves
    auto epilogue = [&](f32x4 (&acc)[MT][NT], const float4 (&bias4)[NT], auto kind_tag) {
        constexpr int kind = decltype(kind_tag)::value;  // 0 conv1, 1 conv2 (+ residual), 2 stem
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const float4 b4 = bias4[n];
            const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float v4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = acc[m][n][j] + bv[j];
                    if (kind == 1) v += res[m][n][j];
                    v = fmaxf(v, 0.0f);
                    if (kind != 0) res[m][n][j] = v;
                    v4[j] = v;
                }
                if (AZX_SAT_TRACK) satmax = fmaxf(fmaxf(satmax, v4[0]), fmaxf(v4[1], fmaxf(v4[2], v4[3])));
                uint2 h4, l4;
                split2_f16(v4[0], v4[1], h4.x, l4.x);
                split2_f16(v4[2], v4[3], h4.y, l4.y);
                // channels cb..cb+3 = half a chunk: k-half n >> 1, k-group 2 (n & 1) + (lh >> 1), bytes 8 (lh & 1)..
                unsigned char *pw = X + (64 * wh + 16 * m + lrow) * ROWB + 128 * (lh >> 1) + 16 * n + 8 * (lh & 1);
                *reinterpret_cast<uint2 *>(pw) = h4;
                *reinterpret_cast<uint2 *>(pw + 64) = l4;
            }
        }
    };

    // ---- stem: one 32-wide k-step with one-hot activations (see k_tower_f16x3_s16) ------------------
    {
        const int NH = N + 2;
        unsigned char *cells = X + 121 * ROWB;
        const uint8_t *bd = ev_board + (size_t)(live ? e : n_eval - 1) * AZX_CELL_STRIDE;
        if (wh == 0) {
            for (int i = lane; i < NH * NH; i += 64) {
                const int y = i / NH - 1, x = i - (y + 1) * NH - 1;
                cells[i] = (y >= 0 && y < N && x >= 0 && x < N) ? bd[y * N + x] : (uint8_t)3;
            }
        }
        for (int i = tid; i < 2 * ROWB / 4; i += F16X3_BPB * 128) reinterpret_cast<uint32_t *>(smem + zero_off)[i] = 0u;
        __syncthreads();
        f32x4 acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        const uint4 *ws = reinterpret_cast<const uint4 *>(P.Ws16);   // [ntile][hi,lo][lane]
        f16x8 wfh[NT], wfl[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const uint4 qh = ws[(n * 2) * 64 + lane], ql = ws[(n * 2 + 1) * 64 + lane];
            wfh[n] = *reinterpret_cast<const f16x8 *>(&qh);
            wfl[n] = *reinterpret_cast<const f16x8 *>(&ql);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            uint32_t onehot = 0u;                        // bit k = 3*tap + colour of that neighbour
            if (64 * wh + 16 * m + lrow < ncells) {
                const unsigned char *c0 = cells + ry_[m] * NH + rx_[m];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const uint32_t v = c0[(tap / 3) * NH + tap % 3];
                    onehot |= (v < 3u ? 1u : 0u) << (3 * tap + v);
                }
            }
            const uint32_t byte = (onehot >> (8 * lh)) & 0xffu;      // k = 8 h + j
            uint4 q;
            q.x = ((byte >> 0) & 1u) * 0x3C00u | ((byte >> 1) & 1u) * 0x3C000000u;
            q.y = ((byte >> 2) & 1u) * 0x3C00u | ((byte >> 3) & 1u) * 0x3C000000u;
            q.z = ((byte >> 4) & 1u) * 0x3C00u | ((byte >> 5) & 1u) * 0x3C000000u;
            q.w = ((byte >> 6) & 1u) * 0x3C00u | ((byte >> 7) & 1u) * 0x3C000000u;
            const f16x8 xf = *reinterpret_cast<const f16x8 *>(&q);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfh[n], xf, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfl[n], xf, acc[m][n], 0, 0, 0);
            }
        }
        float4 sb4[NT];
        load_bias(P.stem_b, sb4);
        __syncthreads();   // every wave has read the staged cells: the tail rows are free again
        epilogue(acc, sb4, std::integral_constant<int, 2>{});
    }
    __syncthreads();
    NT_MARK(0)

    const uint4 *wsrc = reinterpret_cast<const uint4 *>(P.Wh16);     // 512 uint4 per stage: [ntile 4][hi,lo][lane]
    int stage = 0;
    // The weight register set lives across layers: the fragments of a layer's FIRST k-step (channel tiles {0,1})
    // are requested during the LAST k-step of the layer before -- they do not depend on the activations -- so their
    // L2 round trip passes under the epilogue instead of in front of every layer's first MFMA.
    const int last_stage = P.layers * 18 - 1;
    f16x8 wh_[NT], wl_[NT];
    auto load_w = [&](int st, int nn, int part) {
        const uint4 qq = wsrc[(size_t)min(st, last_stage) * 512 + (nn * 2 + part) * 64 + lane];
        if (part) wl_[nn] = *reinterpret_cast<const f16x8 *>(&qq);
        else wh_[nn] = *reinterpret_cast<const f16x8 *>(&qq);
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) load_w(0, i >> 1, i & 1);
    auto conv_layer = [&](int layer, auto residual_tag) {
        constexpr bool residual = decltype(residual_tag)::value;
        asm volatile("" : "+v"(tapok_lo), "+v"(tapok_hi), "+v"(rbase[0]), "+v"(rbase[1]), "+v"(rbase[2]), "+v"(rbase[3]));
        f32x4 acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        // One register set for the weights and one for the activations (64 + 64 VGPRs would not fit
        // beside the accumulators and the residual): a k-step runs as two halves, channel tiles
        // {0,1} then {2,3}, over all four position tiles.  Weights {2,3} of this step are fetched
        // during the first half, weights {0,1} of the next step during the second; position tile
        // m's activations are refilled for the next step once its second-half MFMAs have issued
        // (the last tile's during the next step's first half).
        f16x8 xh[MT], xl[MT];
        auto load_x = [&](int tt, int mm, int part) {
            const unsigned char *pa = smem + act_offset(tt >> 1, mm) + (tt & 1) * 32 + part * 64;
            if (part) xl[mm] = *reinterpret_cast<const f16x8 *>(pa);
            else xh[mm] = *reinterpret_cast<const f16x8 *>(pa);
        };
#pragma unroll
        for (int m = 0; m < MT - 1; ++m) { load_x(0, m, 0); load_x(0, m, 1); }
        // k-step t = 0..17: tap t/2, channels 32 (t%2) .. +31: 2 x 24 MFMAs, 16 loads
#pragma unroll
        for (int t = 0; t < 18; ++t) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int q = 0; q < 24; ++q) {
                    const int m = q / 6, n = 2 * h + (q % 6) / 3, p = q % 3;
                    if (q == 1 || q == 4 || q == 7 || q == 10) {
                        const int idx = (q - 1) / 3;                       // 0..3: tile, part
                        if (h == 0) load_w(stage + t, 2 + (idx >> 1), idx & 1);
                        else load_w(stage + t + 1, idx >> 1, idx & 1);      // t = 17: the next layer's first k-step
                    } else if (h == 0 && (q == 13 || q == 16)) {
                        load_x(t, MT - 1, q == 16);                        // the lagging last tile
                    } else if (h == 1 && q >= 8 && (q % 6 == 2 || q % 6 == 5)) {
                        if (t + 1 < 18) load_x(t + 1, q / 6 - 1, q % 6 == 5);   // tile q/6-1 is done for this step
                    }
                    const f16x8 wv = p == 1 ? wl_[n] : wh_[n];
                    const f16x8 xv = p == 2 ? xl[m] : xh[m];
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, xv, acc[m][n], 0, 0, 0);
#if AZX_S16_FENCE == 2
                    __builtin_amdgcn_sched_barrier(0);
#elif AZX_S16_FENCE == 1
                    if (q % 3 == 2) __builtin_amdgcn_sched_barrier(0);
#elif AZX_S16_FENCE == 3
                    if (q % 6 == 5) __builtin_amdgcn_sched_barrier(0);
#elif AZX_S16_FENCE == 4
                    if (q % 2 == 1) __builtin_amdgcn_sched_barrier(0);
#endif
                }
            }
        }
        stage += 18;
        float4 lb4[NT];
        load_bias(P.bias + layer * C, lb4);
        NT_MARK(2)
        __syncthreads();   // both waves of the board finished reading it
        NT_MARK(3)
        epilogue(acc, lb4, std::integral_constant<int, residual ? 1 : 0>{});
        NT_MARK(4)
        __syncthreads();   // the partner wave wrote the other rows of this board
        NT_MARK(5)
    };
    for (int blk = 0; blk < P.blocks; ++blk) {
        conv_layer(2 * blk, std::false_type{});
        conv_layer(2 * blk + 1, std::true_type{});
    }

    // heads' 1x1 convs + folded BN + ReLU (network.py:77, :83) as one more (tiny) MFMA layer: the last epilogue left
    // the final activations in LDS as hi/lo f16 like every layer's; the six filters are one 16-row A tile (rows 6..15
    // zero), the centre tap's fragments the B operand: 2 k-steps x 3 products per position tile, D[filter 4 lh +
    // reg][position li].  (As fp32 FMAs over the registers with two cross-lane sums per output it was 3 % of the
    // kernel: 48 LDS-crossbar shuffles per wave.)
    if (hfeat != nullptr) {
        const uint4 *whd = reinterpret_cast<const uint4 *>(P.Whd16);
        f16x8 ah[2], al[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint4 qh = whd[(ks * 2) * 64 + lane], ql = whd[(ks * 2 + 1) * 64 + lane];
            ah[ks] = *reinterpret_cast<const f16x8 *>(&qh);
            al[ks] = *reinterpret_cast<const f16x8 *>(&ql);
        }
        const float4 hb = *reinterpret_cast<const float4 *>(P.hbias16 + 4 * lh);
        const float hbv[4] = {hb.x, hb.y, hb.z, hb.w};
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            f32x4 hacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const unsigned char *pa = smem + rbase[m] + ks * 32;
                const f16x8 xh = *reinterpret_cast<const f16x8 *>(pa), xl = *reinterpret_cast<const f16x8 *>(pa + 64);
                hacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], xh, hacc, 0, 0, 0);
                hacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[ks], xh, hacc, 0, 0, 0);
                hacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], xl, hacc, 0, 0, 0);
            }
            const int row = 64 * wh + 16 * m + lrow;
            if (live && row < ncells) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = 4 * lh + r;
                    if (o < 6) hfeat[((size_t)e * 6 + o) * ncells + row] = fmaxf(hacc[r] + hbv[r], 0.0f);
                }
            }
        }
    }
    if (live && act_out != nullptr) {
        float *out = act_out + (size_t)e * ncells * C;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int row = 64 * wh + 16 * m + lrow;
            if (row < ncells) {
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    *reinterpret_cast<float4 *>(out + row * C + 16 * n + 4 * lh) =
                        make_float4(res[m][n][0], res[m][n][1], res[m][n][2], res[m][n][3]);
            }
        }
    }
    if (satmax > 65504.0f) atomicOr(P.sat_flag, 1u);     // an activation left the f16 range: its hi half is +inf (NetDev::sat_flag)
    NT_MARK(6)
    NT_FLUSH
}

// ============================================================================================
// wide residual tower (C a multiple of 128, N <= 13: BASELINE configs[4], 13x13 / 19x256): one
// launch per conv layer, activations in HBM.  A board's activations (169 x 256 x (hi,lo) f16 =
// 173 KB) do not fit the 160 KB LDS, so the in-LDS fusion of k_tower_f16x3_s16 is not available; at
// 7.6 GFLOP per position the 13 MB of activation traffic per position is < 15 % of the MFMA time.
//   * HBM layout per board: [cell][C hi f16 | C lo f16] (the same hi/lo split as above);
//   * a 256-thread block computes ONE board x 128 output channels: wave (wm, wn) owns position
//     tiles 3wm..3wm+2 and channel tiles of co_base + 64 wn (3 x 2 accumulator tiles, 18 MFMAs per
//     16-channel k-step); grid = (boards, C / 128);
//   * the input is staged through LDS in 64-channel chunks with the row format of k_tower_f16x3
//     (128 B hi | 128 B lo | 16 B pad, conflict-free ds_read_b128 fragments, one zero row for the
//     padding taps); weights stream from L2 in fragment order; the k-loop is the one of
//     k_tower_f16x3_s16 (one prefetch load in each MFMA's shadow);
//   * epilogue: + folded-BN bias (+ residual, read from the block input's buffer) -> ReLU -> split
//     -> HBM; conv2 of a Resblock writes over the block input in place (a block reads exactly the
//     residual elements it overwrites); the last layer also writes the fp32 copy k_heads reads.
// ============================================================================================
#define WIDE_ROWB 272
#define WIDE_MW 3
#define WIDE_NW 2

// ---- the wide tower on v_mfma_f32_16x16x32_f16 (see k_tower_f16x3_s16 for why) ----------------
// Same block decomposition as k_conv_wide_f16x3 (one board x 128 output channels per 256-thread
// block, wave (wm, wn): 96 positions x 64 channels = 6 x 4 tiles of 16 x 16), same HBM layout and
// LDS staging; a k-step is one tap x 32 channels of the staged 64-channel chunk (18 per chunk):
// 72 MFMAs, 8 weight + 12 activation fragment loads, one register set each, two channel halves.
#ifndef WIDE_STAGE_GROUP
#define WIDE_STAGE_GROUP 6
#endif
// diagnostic builds (tools/ab_wide_ablate.sh; wrong results, timing only): 1 = stage chunk 0 only,
// 2 = no epilogue memory traffic, 4 = weight fragments loaded once per chunk, 8 = activation fragments
// loaded once per chunk
#ifndef AZX_WIDE_ABLATE
#define AZX_WIDE_ABLATE 0
#endif

#define WIDE16_MT 6
#define WIDE16_NT 4
// TRAIN = 1 (csrc/train_wide.hip, the training step's forward and backward-data convolutions of a wide tower): the same
// product on the same staging and k-loop, but the accumulators start from zero (no bias, no residual), and the epilogue
// writes the RAW fp32 output times `unscale` (the operands' power-of-two scales taken out) -- no ReLU, no f16 image --
// plus, when `stat` is given, this board's per-channel (sum, sum of squares) of what it wrote: train-mode BatchNorm's
// batch statistics as per-board partial pairs (summed over the boards in a fixed order by their consumers).
// TRAIN = 2: the backward-data convolution with the step's next elementwise pass in its epilogue (WideBwdFuse): the product
// is dL/dact_{l-1}; what is written is g_{l-1} = (product [+ skip]) where act_{l-1} > 0, else 0; `stat` receives this
// board's (sum g, sum g xhat_{l-1}) per channel (xhat from raw_{l-1} and BN_{l-1}'s batch sums) and *gmax max |g|.
// (`mask`: act_{l-1} > 0 as one bit per element, a byte per 8 consecutive channels -- 0.7 MB per layer at 19x256 / 13x13 /
// B = 128 where the fp32 activations were a 22 MB read)
struct WideBwdFuse { const unsigned char *mask; const float *raw, *skip; const double *sums; float invN; unsigned int *gmax; };
// NTW = 16-channel tiles per wave: 4 (a block = board x 128 channels: the inference decomposition) or, TRAIN only, 2 (a
// block = board x 64 channels).  A training batch of 128 boards is 256 of the former -- one workgroup per CU, one wave per
// SIMD, nobody to run while a block stages its next chunk -- and 512 of the latter: two independent workgroups per CU, the
// regime the kernel was tuned in, at the price of twice the activation-fragment reads per MFMA (well inside the LDS rate).
template <int TRAIN, int NTW = WIDE16_NT>
__device__ __forceinline__ void conv_wide_s16_body(const NetDev &P, int layer, const unsigned short *__restrict__ in,
                                                   unsigned short *out, const unsigned short *resid,
                                                   float *__restrict__ out32,
                                                   const int32_t *__restrict__ n_eval_ptr, int n_eval_host,
                                                   int e_base, int e_end, float unscale, float2 *__restrict__ stat,
                                                   const WideBwdFuse &F = WideBwdFuse{}) {
    constexpr int MT = WIDE16_MT, NT = NTW, ROWB = WIDE_ROWB;
    static_assert(NT == 4 || (NT == 2 && TRAIN != 0), "two tiles per wave: training instantiations only");
    float satmax = 0.f;
    extern __shared__ __align__(16) unsigned char smem[];
    const int n_eval = n_eval_ptr ? *n_eval_ptr : n_eval_host;
    const int C = P.C, N = P.N, ncells = P.ncells;
    const int NCH = C / 64, NT16 = C / 16;
    NT_DECL
    // XCD-aware block order: consecutive workgroups go to the eight XCDs round-robin, each with its own L2.
    // The C / 128 column blocks of a board are given indices 8 apart, so they run on the same XCD at about
    // the same time and the second one finds the board's input in that L2 instead of fetching it from HBM
    // again (with (board, column) as (x, y) the two were a whole launch wave apart).
    // (grid = (8 C / 128, boards / 8): the linear workgroup id is blockIdx.y * gridDim.x + blockIdx.x)
    // [e_base, e_end): the boards of this launch (the host splits a layer's boards over two streams)
    const int e = e_base + blockIdx.y * 8 + (blockIdx.x & 7);
    if (e >= n_eval || e >= e_end) return;
    const int co_base = (blockIdx.x >> 3) * (32 * NT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Position tiles: a board of up to 176 cells is eleven 16-row tiles, not twelve.  Wave wm = 0 holds
    // tiles 0..5 and wm = 1 tiles 5..10 (rows 80..175); the shared tile 5 is computed by wm = 0 for the
    // first two of the wave's four channel tiles and by wm = 1 for the other two, so each wave issues
    // 22 tile products per k-step (6 x 2 + 5 x 2) instead of 24, one twelfth of which was padding.
    const int wm = wave & 1, wn = wave >> 1;
    const int row0 = wm ? 80 : 0;
    const int li = lane & 15, lh = lane >> 4;
    const size_t rowg = (size_t)C * 4;
    const unsigned char *gin = reinterpret_cast<const unsigned char *>(in) + (size_t)e * ncells * rowg;
    // LDS image of a 64-channel chunk, conflict-free for ds_read_b128 as in k_tower_f16x3_s16: two zero rows first
    // (a padding tap reads at its own address mod 256), tile column li is tile row S16_ROW(li), and chunk (part p,
    // k-half kh, k-group g) of a row sits at byte 128 (g & 1) + 16 (4 p + 2 kh + (g >> 1)).
    constexpr int IMG0 = 2 * ROWB;
    const int lrow = li < 8 ? li ^ 4 : li;
    const int lchunk = 128 * (lh & 1) + 16 * (lh >> 1);

    unsigned long long tapok = 0ull;                     // bit tap*6 + m (54 bits)
    int rbase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int r = row0 + 16 * m + lrow;
        const int ry = r / N, rx = r - ry * N;
        rbase[m] = IMG0 + r * ROWB + lchunk;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = ry + tap / 3 - 1, xx = rx + tap % 3 - 1;
            if (r < ncells && yy >= 0 && yy < N && xx >= 0 && xx < N) tapok |= 1ull << (tap * 6 + m);
        }
    }
    // The 54 (tap, tile) fragment offsets are invariant over the chunks; hoisted out of the chunk loop they hold
    // 54 VGPRs and the kernel spills.  The mask words and row bases pass through an opaque asm at the top of
    // every chunk, so an offset is formed where it is used and lives for its two k-steps only.
    uint32_t tapok_lo = (uint32_t)tapok & 0x3fffffffu, tapok_hi = (uint32_t)(tapok >> 30);   // taps 0..4 | 5..8
    auto act_offset = [&](int tap, int m) -> int {
        const int delta = ((tap / 3 - 1) * N + (tap % 3 - 1)) * ROWB;
        const uint32_t word = tap < 5 ? tapok_lo : tapok_hi;
        const int bit = (tap < 5 ? tap : tap - 5) * 6 + m;
        const int mask = (int)(word << (31 - bit)) >> 31;                    // v_bfe_i32: 0 or -1
        return (mask | 0xF0) & (rbase[m] + delta);
    };
    if (tid < IMG0 / 4) reinterpret_cast<uint32_t *>(smem)[tid] = 0u;

    // Output channels of this wave: 64 wn .. 64 wn + 63 of the block's 128.  The weights are packed so that
    // accumulator register r of tile n in lane group lh is channel 32 (n >> 1) + 8 lh + 4 (n & 1) + r of
    // those (see azx_net_set_weights): a lane holds 8 consecutive channels per tile pair, i.e. one 16-byte
    // piece of the hi plane and one of the lo plane per row, and the four lane groups of a row together one
    // full 64-byte line -- residual loads and output stores are whole lines, half as many instructions.
    auto chan0 = [&](int np) -> int { return co_base + 16 * NT * wn + 32 * np + 8 * lh; };
    // The accumulators start from bias (+ residual), fetched here together with the first input chunk (one
    // memory round trip, nothing else to do yet) instead of in the epilogue, where every wave of the block
    // waited for it with the matrix pipe idle.
    f32x4 acc[MT][NT];
    if constexpr (TRAIN != 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
        const unsigned char *gres0 = resid ? reinterpret_cast<const unsigned char *>(resid) + (size_t)e * ncells * rowg : nullptr;
#if AZX_WIDE_ABLATE & 16
        gres0 = nullptr;
#endif
        float bv[2][8];
#pragma unroll
        for (int np = 0; np < 2; ++np) {
            const float4 b0 = *reinterpret_cast<const float4 *>(P.bias + (size_t)layer * C + chan0(np));
            const float4 b1 = *reinterpret_cast<const float4 *>(P.bias + (size_t)layer * C + chan0(np) + 4);
            bv[np][0] = b0.x; bv[np][1] = b0.y; bv[np][2] = b0.z; bv[np][3] = b0.w;
            bv[np][4] = b1.x; bv[np][5] = b1.y; bv[np][6] = b1.z; bv[np][7] = b1.w;
        }
        // three position tiles at a time: all their pieces requested together (48 registers in flight)
#pragma unroll
        for (int m0 = 0; m0 < MT; m0 += 3) {
            f16x8 rh[3][2], rl[3][2];
            if (gres0) {
#pragma unroll
                for (int mm = 0; mm < 3; ++mm) {
                    const int rc = min(row0 + 16 * (m0 + mm) + lrow, ncells - 1);
#pragma unroll
                    for (int np = 0; np < 2; ++np) {
                        rh[mm][np] = *reinterpret_cast<const f16x8 *>(gres0 + (size_t)rc * rowg + chan0(np) * 2);
                        rl[mm][np] = *reinterpret_cast<const f16x8 *>(gres0 + (size_t)rc * rowg + (size_t)C * 2 + chan0(np) * 2);
                    }
                }
            }
#pragma unroll
            for (int mm = 0; mm < 3; ++mm)
#pragma unroll
                for (int np = 0; np < 2; ++np)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float v = bv[np][j];
                        if (gres0) v += (float)rh[mm][np][j] + (float)rl[mm][np][j];
                        acc[m0 + mm][2 * np + (j >> 2)][j & 3] = v;
                    }
        }
    }

    const uint4 *wsrc = reinterpret_cast<const uint4 *>(P.Wh16);
    // this wave's first 16-channel tile: uniform over the wave, so as a scalar the 64-bit part of the weight-fragment
    // addresses is formed on the scalar unit (per lane it was a v_mad_i64_i32 + two 64-bit shifts/adds per load)
    const int nt0 = __builtin_amdgcn_readfirstlane(co_base / 16 + NT * wn);
    // weights of k-step q (global index over layer, tap, chunk, half): [q][ntile][part][lane]
    auto wptr = [&](int q, int n, int part) -> const uint4 * {
        return wsrc + ((size_t)q * (NT16 * 2) + (size_t)((nt0 + n) * 2 + part)) * 64 + lane;
    };

    // the chunk loop, instantiated per wm (which tile product is the other wave's is a compile-time
    // pattern: straight-line code either way; both paths pass the same barriers)
    // (Measured and dropped, round 6: a phase shift between the two co-resident blocks of a CU -- a training launch is ONE
    // round of blocks, two per CU, started together; every second arrival on a CU (an atomic per XCC_ID / HW_ID) slept half
    // a chunk's period so that one block would stage under the other's k-loop: 65.3 vs 62.5 us forward, 84.1 vs 82.7
    // backward, the filter gradient 95.3 vs 93.7 -- the blocks do not run in lock-step, the sleep is simply paid.)
    // (Measured and dropped, round 6: requesting chunk c + 1's pieces into registers before chunk c's k-loop -- a
    // training batch is ONE round of blocks, nobody else's k-loop covers a block's staging -- 63.1 vs 62.5 us forward,
    // 82.0 vs 82.2 backward: the staging round trips are not what the kernel waits for.  profiles/r6_train_wide_ab.txt)
    auto main_loop = [&](auto wm_tag) __attribute__((always_inline)) {
    constexpr int WM = decltype(wm_tag)::value;
    // (Measured and dropped, round 6: three weight register sets, k-step g's fragments requested during k-step g - 2 and
    // across chunk boundaries -- a two-tile k-step is 576 matrix-pipe cycles, about an L2 round trip: 67.5 vs 60.9 us
    // forward, 85.9 vs 80.0 backward; the weights are not late, the 50 extra registers cost.  profiles/r6_train_wide_ab.txt)
    for (int chunk = 0; chunk < NCH; ++chunk) {
#if AZX_WIDE_ABLATE & 1
        if (chunk == 0)
#endif
        {
        NT_MARK(chunk == 0 ? 0 : 2)
        __syncthreads();                                 // the previous chunk has been consumed
        NT_MARK(4)                                       // (stamp builds: the wait at this barrier = the waves' skew)
        {
            // all of a thread's pieces are requested before the first one is written to LDS (as a plain
            // loop the compiler waited for each 16-byte load in turn: eleven memory round trips per
            // chunk); the tail repeats the last piece instead of diverging
            constexpr int NST = (AZX_MAX_BOARD * AZX_MAX_BOARD * 16 + 255) / 256;
            constexpr int GRP = WIDE_STAGE_GROUP;            // pieces in flight per thread (register budget)
            const int last = ncells * 16 - 1;
            int tid_o = tid;                                 // opaque per chunk: the piece addresses are
            asm volatile("" : "+v"(tid_o));                  // recomputed here, not hoisted and spilled
#pragma unroll
            for (int j0 = 0; j0 < NST; j0 += GRP) {
                uint4 stg[GRP];
#pragma unroll
                for (int j = 0; j < GRP; ++j) {
                    const int idx = min(tid_o + 256 * (j0 + j), last);
                    // piece d of the LDS row (destination order, contiguous stores): k-group 2 (d & 1) + (d >> 3),
                    // part (d >> 2) & 1, k-half (d >> 1) & 1 -- a row's 16 lanes still read its two 128-byte lines whole
                    const int row = idx >> 4, d = idx & 15;
                    const size_t src = (size_t)row * rowg + (size_t)chunk * 128 + ((d >> 2) & 1) * (size_t)C * 2 +
                                       ((d >> 1) & 1) * 64 + (2 * (d & 1) + (d >> 3)) * 16;
                    if (j0 + j < NST) stg[j] = *reinterpret_cast<const uint4 *>(gin + src);
                }
#pragma unroll
                for (int j = 0; j < GRP; ++j) {
                    const int idx = min(tid_o + 256 * (j0 + j), last);
                    const int dst = IMG0 + (idx >> 4) * ROWB + (idx & 15) * 16;
                    if (j0 + j < NST) *reinterpret_cast<uint4 *>(smem + dst) = stg[j];
                }
            }
        }
        NT_MARK(5)                                       // (stamp builds: requests, their round trips, LDS writes)
        __syncthreads();
        NT_MARK(1)
        }
        // k-step t = 0..17 of this chunk: tap t/2, channels 32 (t%2) .. +31 of the chunk
        asm volatile("" : "+v"(tapok_lo), "+v"(tapok_hi), "+v"(rbase[0]), "+v"(rbase[1]), "+v"(rbase[2]),
                          "+v"(rbase[3]), "+v"(rbase[4]), "+v"(rbase[5]));
        auto qof = [&](int t) { return (((layer * 9 + t / 2) * NCH + chunk) * 2 + (t & 1)); };
        f16x8 wh_[NT], wl_[NT];
        f16x8 xh[MT], xl[MT];
        auto load_w = [&](int t, int nn, int part) {
#if AZX_WIDE_ABLATE & 4
            if (t > 0) return;
#endif
            const uint4 qq = *wptr(qof(t), nn, part);
            if (part) wl_[nn] = *reinterpret_cast<const f16x8 *>(&qq);
            else wh_[nn] = *reinterpret_cast<const f16x8 *>(&qq);
        };
        auto load_x = [&](int tt, int mm, int part) {
#if AZX_WIDE_ABLATE & 8
            if (tt > 0) return;
#endif
            const unsigned char *pa = smem + act_offset(tt >> 1, mm) + (tt & 1) * 32 + part * 64;
            if (part) xl[mm] = *reinterpret_cast<const f16x8 *>(pa);
            else xh[mm] = *reinterpret_cast<const f16x8 *>(pa);
        };
        if constexpr (NT == 2) {
            // two tiles per wave: a k-step is ONE pass of 36 MFMAs (6 position tiles x 2 channel tiles x 3); the weights
            // of k-step t + 1 go into the other of two register sets in the first MFMAs' shadows, a position tile's
            // activation fragments are reloaded for t + 1 one tile after its MFMAs have issued, the last tile lags
            f16x8 w2h[2][2], w2l[2][2];
            auto load_w2 = [&](int t, int nn, int part) {
                const uint4 qq = *wptr(qof(t), nn, part);
                if (part) w2l[t & 1][nn] = *reinterpret_cast<const f16x8 *>(&qq);
                else w2h[t & 1][nn] = *reinterpret_cast<const f16x8 *>(&qq);
            };
#pragma unroll
            for (int i = 0; i < 4; ++i) load_w2(0, i >> 1, i & 1);
#pragma unroll
            for (int m = 0; m < MT - 1; ++m) { load_x(0, m, 0); load_x(0, m, 1); }
#pragma unroll
            for (int t = 0; t < 18; ++t) {
#pragma unroll
                for (int q = 0; q < 6 * MT; ++q) {
                    const int m = q / 6, n = (q % 6) / 3, p = q % 3;
                    if (q == 1 || q == 4 || q == 7 || q == 10) {
                        const int idx = (q - 1) / 3;
                        if (t + 1 < 18) load_w2(t + 1, idx >> 1, idx & 1);
                    } else if (q == 13 || q == 16) {
                        load_x(t, MT - 1, q == 16);       // the lagging last tile
                    } else if (q >= 8 && (q % 6 == 2 || q % 6 == 5)) {
                        if (t + 1 < 18) load_x(t + 1, q / 6 - 1, q % 6 == 5);
                    }
                    const f16x8 wv = p == 1 ? w2l[t & 1][n] : w2h[t & 1][n];
                    const f16x8 xv = p == 2 ? xl[m] : xh[m];
                    if (!(WM == 0 ? (m == MT - 1 && n >= 1) : (m == 0 && n < 1)))
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, xv, acc[m][n], 0, 0, 0);
                    if (q % 3 == 2) __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) load_w(0, i >> 1, i & 1);
#pragma unroll
        for (int m = 0; m < MT - 1; ++m) { load_x(0, m, 0); load_x(0, m, 1); }
#pragma unroll
        for (int t = 0; t < 18; ++t) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int q = 0; q < 6 * MT; ++q) {                 // 36 MFMAs: 6 position tiles x 2 channel tiles x 3
                    const int m = q / 6, n = 2 * h + (q % 6) / 3, p = q % 3;
                    if (q == 1 || q == 4 || q == 7 || q == 10) {
                        const int idx = (q - 1) / 3;
                        if (h == 0) load_w(t, 2 + (idx >> 1), idx & 1);
                        else if (t + 1 < 18) load_w(t + 1, idx >> 1, idx & 1);
                    } else if (h == 0 && (q == 13 || q == 16)) {
                        load_x(t, MT - 1, q == 16);       // the lagging last tile
                    } else if (h == 1 && q >= 8 && (q % 6 == 2 || q % 6 == 5)) {
                        if (t + 1 < 18) load_x(t + 1, q / 6 - 1, q % 6 == 5);
                    }
                    const f16x8 wv = p == 1 ? wl_[n] : wh_[n];
                    const f16x8 xv = p == 2 ? xl[m] : xh[m];
                    if (!(WM == 0 ? (m == MT - 1 && n >= 2) : (m == 0 && n < 2)))
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, xv, acc[m][n], 0, 0, 0);
                    if (q % 3 == 2) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        }
    }

    };
    // TRAIN = 2: the epilogue's operands (raw_{l-1}, the skip gradient, the mask bytes) of all six position tiles are
    // requested together at the top of the epilogue.  (Measured and dropped, round 6: the first three tiles' 51 registers
    // requested before the k-loop, so that they arrive under it: 82.7 vs 83.4 us -- their round trip is not what the
    // epilogue costs.  profiles/r6_train_wide_ab.txt)
    constexpr int NPF = NT / 2;
    f32x4 er0[3][NPF][2], ek0[3][NPF][2];       // (the native vector type: arrays of HIP's float4 STRUCT handed to a lambda stay in scratch)
    unsigned int ea0[3][NPF];
    auto epi_request = [&](int m0, f32x4 (&er)[3][NPF][2], f32x4 (&ek)[3][NPF][2], unsigned int (&ea)[3][NPF]) __attribute__((always_inline)) {
#pragma unroll
        for (int mm = 0; mm < 3; ++mm)
#pragma unroll
            for (int np = 0; np < NPF; ++np) {
                const int rc = min(row0 + 16 * (m0 + mm) + lrow, ncells - 1);
                const size_t o = ((size_t)e * ncells + rc) * C + chan0(np);
                ea[mm][np] = F.mask[o >> 3];
                er[mm][np][0] = *reinterpret_cast<const f32x4 *>(F.raw + o);
                er[mm][np][1] = *reinterpret_cast<const f32x4 *>(F.raw + o + 4);
                if (F.skip) {        // (wave-uniform: a kernel argument)
                    ek[mm][np][0] = *reinterpret_cast<const f32x4 *>(F.skip + o);
                    ek[mm][np][1] = *reinterpret_cast<const f32x4 *>(F.skip + o + 4);
                } else {
                    ek[mm][np][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                    ek[mm][np][1] = ek[mm][np][0];
                }
            }
    };

    if (wm == 0) main_loop(std::integral_constant<int, 0>{});
    else main_loop(std::integral_constant<int, 1>{});

    if constexpr (TRAIN != 0) {
        // ---- training epilogue: raw fp32 out, per-board channel sums ---------------------------------------------
        constexpr int NP = NT / 2;                         // tile pairs: a lane's 8 consecutive channels each
        // tile n of position tile m is this wave's unless it is the shared tile's other half (see the k-loop)
        auto tile_ok = [&](int m, int n) -> bool { return !(wm == 0 ? (m == MT - 1 && n >= NT / 2) : (m == 0 && n < NT / 2)); };
        float s1[NP][8], s2[NP][8];
#pragma unroll
        for (int np = 0; np < NP; ++np)
#pragma unroll
            for (int j = 0; j < 8; ++j) { s1[np][j] = 0.f; s2[np][j] = 0.f; }
        float pM[NP][8], pI[NP][8], vmax = 0.f;
        if (TRAIN == 2) {
            // mean / 1/std of BN_{l-1} for this lane's channels, from the batch sums the forward pass filed
#pragma unroll
            for (int np = 0; np < NP; ++np)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const double *sm = F.sums + (size_t)(chan0(np) + j) * 4;
                    const double mn = sm[0] * (double)F.invN, vr = sm[1] * (double)F.invN - mn * mn;
                    pM[np][j] = (float)mn;
                    pI[np][j] = (float)(1.0 / sqrt((vr > 0 ? vr : 0) + 1e-5));
                }
        }
        // Three position tiles at a time, and (TRAIN = 2) all of their mask / xhat / skip pieces requested before the first
        // is used, unconditionally at a clamped row: behind `if (row is mine)` the compiler waits for each tile's loads at
        // the join -- six memory round trips at the end of a kernel that has nothing left to hide them under.
        f32x4 er1[3][NP][2], ek1[3][NP][2];
        unsigned int ea1[3][NP];
        if constexpr (TRAIN == 2) {
            epi_request(0, er0, ek0, ea0);
            epi_request(3, er1, ek1, ea1);
        }
        auto epi_group = [&](auto m0_tag, const f32x4 (&er)[3][NP][2], const f32x4 (&ek)[3][NP][2],
                             const unsigned int (&ea)[3][NP]) __attribute__((always_inline)) {
            constexpr int m0 = decltype(m0_tag)::value;
#pragma unroll
            for (int mm = 0; mm < 3; ++mm) {
                const int m = m0 + mm;
                const int r = row0 + 16 * m + lrow;
#pragma unroll
                for (int np = 0; np < NP; ++np) {
                    const bool ok0 = tile_ok(m, 2 * np), ok1 = tile_ok(m, 2 * np + 1);
                    if (r < ncells && (ok0 || ok1)) {
                        const size_t o = ((size_t)e * ncells + r) * C + chan0(np);
                        float vv[8];
                        if (TRAIN == 2) {
                            const f32x4 r0 = er[mm][np][0], r1 = er[mm][np][1];
                            const f32x4 k0 = ek[mm][np][0], k1 = ek[mm][np][1];
                            const unsigned int bits = ea[mm][np];
                            const float rv[8] = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
                            const float kv[8] = {k0[0], k0[1], k0[2], k0[3], k1[0], k1[1], k1[2], k1[3]};
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const bool ok = (j >> 2) ? ok1 : ok0;
                                vv[j] = ((bits >> j) & 1u) ? acc[m][2 * np + (j >> 2)][j & 3] * unscale + kv[j] : 0.f;
                                if (ok) {
                                    vmax = fmaxf(vmax, fabsf(vv[j]));
                                    s1[np][j] += vv[j];
                                    s2[np][j] += vv[j] * (rv[j] - pM[np][j]) * pI[np][j];
                                }
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const bool ok = (j >> 2) ? ok1 : ok0;
                                vv[j] = acc[m][2 * np + (j >> 2)][j & 3] * unscale;
                                if (ok) {
                                    s1[np][j] += vv[j];
                                    s2[np][j] += vv[j] * vv[j];
                                }
                            }
                        }
                        float *o32 = out32 + o;
                        if (ok0) *reinterpret_cast<float4 *>(o32) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                        if (ok1) *reinterpret_cast<float4 *>(o32 + 4) = make_float4(vv[4], vv[5], vv[6], vv[7]);
                    }
                }
            }
        };
        epi_group(std::integral_constant<int, 0>{}, er0, ek0, ea0);
        epi_group(std::integral_constant<int, 3>{}, er1, ek1, ea1);
        if (TRAIN == 2) {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
            // ONE atomic per block, below (through LDS with the channel sums).  One per WAVE -- 2 048 atomics on one word at
            // the end of a launch whose blocks all finish together -- was 16 of this kernel's 80 us (round 6, ablation:
            // profiles/r6_train_wide_ab.txt 9)
            if (!stat && lane == 0) atomicMax(F.gmax + ((blockIdx.y * gridDim.x + blockIdx.x) & 31), __float_as_uint(vmax));
        }
        if (stat) {
            // a channel's rows sit in the 16 lanes li of one lane group lh and in both position waves: lanes first
            // (xor 1, 2, 4, 8 stay inside the group), then the two waves through LDS (the chunk image is done with)
#pragma unroll
            for (int np = 0; np < NP; ++np)
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) {
                        s1[np][j] += __shfl_xor(s1[np][j], o);
                        s2[np][j] += __shfl_xor(s2[np][j], o);
                    }
            __syncthreads();                            // every wave has left the k-loop: the image can be overwritten
            constexpr int BC = 32 * NT;                 // channels of the block
            float2 *sst = reinterpret_cast<float2 *>(smem);     // [wm][BC]
            if (li == 0) {
#pragma unroll
                for (int np = 0; np < NP; ++np)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        sst[wm * BC + 16 * NT * wn + 32 * np + 8 * lh + j] = make_float2(s1[np][j], s2[np][j]);
            }
            float *wmax = reinterpret_cast<float *>(sst + 2 * BC);     // [4]: the waves' max |g|
            if (TRAIN == 2 && lane == 0) wmax[wave] = vmax;
            __syncthreads();
            if (tid < BC) {
                const float2 a = sst[tid], b = sst[BC + tid];
                stat[(size_t)e * C + co_base + tid] = make_float2(a.x + b.x, a.y + b.y);
            }
            if (TRAIN == 2 && tid == 0)      // F.gmax: 32 slots of this layer (train_kernels.hip: TW_GSLOTS) -- 512 blocks, 16 atomics a word
                atomicMax(F.gmax + ((blockIdx.y * gridDim.x + blockIdx.x) & 31), __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
        }
        return;
    } else {
    // ---- epilogue: ReLU -> split -> HBM, one 16-byte piece per lane and plane -----------------------
    NT_MARK(2)
    unsigned char *gout = reinterpret_cast<unsigned char *>(out) + (size_t)e * ncells * rowg;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int r = row0 + 16 * m + lrow;
#pragma unroll
        for (int np = 0; np < 2; ++np) {
            bool mine = !(wm == 0 ? (m == MT - 1 && np == 1) : (m == 0 && np == 0));   // else the other wave's
#if AZX_WIDE_ABLATE & 2
            mine = mine && acc[m][2 * np][0] == 12345.678f;
#endif
            if (r < ncells && mine) {
                float vv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) vv[j] = fmaxf(acc[m][2 * np + (j >> 2)][j & 3], 0.0f);
                if (AZX_SAT_TRACK) satmax = fmaxf(fmaxf(fmaxf(satmax, vv[0]), fmaxf(vv[1], vv[2])), fmaxf(fmaxf(vv[3], vv[4]), fmaxf(vv[5], fmaxf(vv[6], vv[7]))));
                uint4 h8, l8;
                split2_f16(vv[0], vv[1], h8.x, l8.x);
                split2_f16(vv[2], vv[3], h8.y, l8.y);
                split2_f16(vv[4], vv[5], h8.z, l8.z);
                split2_f16(vv[6], vv[7], h8.w, l8.w);
                const int c0 = chan0(np);
                *reinterpret_cast<uint4 *>(gout + (size_t)r * rowg + c0 * 2) = h8;
                *reinterpret_cast<uint4 *>(gout + (size_t)r * rowg + (size_t)C * 2 + c0 * 2) = l8;
                if (out32) {
                    float *o32 = out32 + ((size_t)e * ncells + r) * C + c0;
                    *reinterpret_cast<float4 *>(o32) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                    *reinterpret_cast<float4 *>(o32 + 4) = make_float4(vv[4], vv[5], vv[6], vv[7]);
                }
            }
        }
    }
#ifdef AZX_NET_STAMP
    __builtin_amdgcn_s_waitcnt(0);      // the stores have been acknowledged: their time belongs to the epilogue
#endif
    if (satmax > 65504.0f) atomicOr(P.sat_flag, 1u);     // an activation left the f16 range: its hi half is +inf (NetDev::sat_flag)
    NT_MARK(3)
    WT_FLUSH
    }
}

__global__ __launch_bounds__(256, 2) void k_conv_wide_f16x3_s16(NetDev P, int layer, const unsigned short *__restrict__ in,
                                                                unsigned short *out, const unsigned short *resid,
                                                                float *__restrict__ out32,
                                                                const int32_t *__restrict__ n_eval_ptr, int n_eval_host,
                                                                int e_base, int e_end) {
    conv_wide_s16_body<0>(P, layer, in, out, resid, out32, n_eval_ptr, n_eval_host, e_base, e_end, 1.f, nullptr);
}

// the training step's convolutions (TRAIN = 1 above): `w16` = one layer's fragments in the wide pack
// ([tap][chunk][half][ntile][hi, lo][lane][8]: k_tw_pack), boards [0, n_boards)
template <int NTW>
__global__ __launch_bounds__(256, 2) void k_conv_wide_train(int N, int C, const unsigned short *__restrict__ w16,
                                                            const unsigned short *__restrict__ in, float *__restrict__ out32,
                                                            int n_boards, const float *__restrict__ unscale, float2 *__restrict__ stat) {
    NetDev P;
    P.N = N; P.ncells = N * N; P.C = C; P.Wh16 = w16; P.bias = nullptr; P.sat_flag = nullptr;
    conv_wide_s16_body<1, NTW>(P, 0, in, nullptr, nullptr, out32, nullptr, n_boards, 0, n_boards, *unscale, stat);
}

template <int NTW>
__global__ __launch_bounds__(256, 2) void k_conv_wide_train_bwd(int N, int C, const unsigned short *__restrict__ w16,
                                                                const unsigned short *__restrict__ in, float *__restrict__ out32,
                                                                int n_boards, const float *__restrict__ unscale, float2 *__restrict__ stat,
                                                                WideBwdFuse F) {
    NetDev P;
    P.N = N; P.ncells = N * N; P.C = C; P.Wh16 = w16; P.bias = nullptr; P.sat_flag = nullptr;
    conv_wide_s16_body<2, NTW>(P, 0, in, nullptr, nullptr, out32, nullptr, n_boards, 0, n_boards, *unscale, stat, F);
}

// (the training convolutions run with two tiles per wave, board x 64 channels per block: the four-tile form measured
// 73 vs 61-64 us forward, 10.8-11.0 vs 10.4-10.6 ms per step, and was removed)

int azx_net_wide_train_conv_bwd(int N, int C, const unsigned short *w16, const unsigned short *in, float *g_out, int n_boards,
                                const float *unscale, float2 *pgsum, const unsigned char *mask, const float *raw, const float *skip,
                                const double *sums, float invN, unsigned int *gmax, hipStream_t st) {
    // (N * N + 2) * 272 bytes: 46.5 KB at 13x13, inside the 48 KB a kernel may ask for without raising its limit -- no
    // per-process "raised" flag to go stale on a second device
    const size_t lds = (size_t)(N * N + 2) * WIDE_ROWB;
    if (lds > 48 * 1024) return AZX_EINVAL;
    const WideBwdFuse F = {mask, raw, skip, sums, invN, gmax};
    hipLaunchKernelGGL(k_conv_wide_train_bwd<2>, dim3(8 * (C / 64), (n_boards + 7) / 8), dim3(256), lds, st, N, C, w16, in, g_out, n_boards, unscale, pgsum, F);
    return AZX_OK;
}

int azx_net_wide_train_conv(int N, int C, const unsigned short *w16, const unsigned short *in, float *out32, int n_boards,
                            const float *unscale, float2 *stat, hipStream_t st) {
    const size_t lds = (size_t)(N * N + 2) * WIDE_ROWB;      // <= 46.5 KB (see azx_net_wide_train_conv_bwd)
    if (lds > 48 * 1024) return AZX_EINVAL;
    hipLaunchKernelGGL(k_conv_wide_train<2>, dim3(8 * (C / 64), (n_boards + 7) / 8), dim3(256), lds, st, N, C, w16, in, out32, n_boards, unscale, stat);
    return AZX_OK;
}

// stem of the wide tower: the one-hot K = 27 product of k_tower_f16x3_s16's stem, one board x 128
// output channels per block, straight to the HBM activation layout
__global__ __launch_bounds__(256, 2) void k_stem_wide_f16x3(NetDev P, const uint8_t *__restrict__ ev_board,
                                                            unsigned short *out, float *__restrict__ out32,
                                                            const int32_t *__restrict__ n_eval_ptr, int n_eval_host) {
    constexpr int MW = WIDE_MW, NW = WIDE_NW;
    float satmax = 0.f;
    __shared__ unsigned char cells[(AZX_MAX_BOARD + 2) * (AZX_MAX_BOARD + 2) + 15];
    const int n_eval = n_eval_ptr ? *n_eval_ptr : n_eval_host;
    const int e = blockIdx.x;
    if (e >= n_eval) return;
    const int C = P.C, N = P.N, ncells = P.ncells, NT = C / 32, NH = N + 2;
    const int co_base = blockIdx.y * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int li = lane & 31, lh = lane >> 5;
    const size_t rowg = (size_t)C * 4;
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    const uint8_t *bd = ev_board + (size_t)e * AZX_CELL_STRIDE;
    for (int i = tid; i < NH * NH; i += 256) {
        const int y = i / NH - 1, x = i - (y + 1) * NH - 1;
        cells[i] = (y >= 0 && y < N && x >= 0 && x < N) ? bd[y * N + x] : (uint8_t)3;
    }
    __syncthreads();
    uint32_t onehot[MW];
#pragma unroll
    for (int m = 0; m < MW; ++m) {
        onehot[m] = 0u;
        const int r = (MW * wm + m) * 32 + li;
        if (r < ncells) {
            const int ry = r / N, rx = r - ry * N;
            const unsigned char *c0 = cells + ry * NH + rx;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const uint32_t v = c0[(tap / 3) * NH + tap % 3];
                onehot[m] |= (v < 3u ? 1u : 0u) << (3 * tap + v);
            }
        }
    }
    f32x16 acc[MW][NW];
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int n = 0; n < NW; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
    const uint4 *ws = reinterpret_cast<const uint4 *>(P.Ws);   // [kk][ntile][hi,lo][lane]
    const int nt0 = co_base / 32 + 2 * wn;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        f16x8 bfr[MW];
#pragma unroll
        for (int m = 0; m < MW; ++m) {
            const uint32_t byte = (onehot[m] >> (16 * kk + 8 * lh)) & 0xffu;
            uint4 q;
            q.x = ((byte >> 0) & 1u) * 0x3C00u | ((byte >> 1) & 1u) * 0x3C000000u;
            q.y = ((byte >> 2) & 1u) * 0x3C00u | ((byte >> 3) & 1u) * 0x3C000000u;
            q.z = ((byte >> 4) & 1u) * 0x3C00u | ((byte >> 5) & 1u) * 0x3C000000u;
            q.w = ((byte >> 6) & 1u) * 0x3C00u | ((byte >> 7) & 1u) * 0x3C000000u;
            bfr[m] = *reinterpret_cast<const f16x8 *>(&q);
        }
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            const uint4 *pa = ws + ((size_t)(kk * NT + nt0 + n) * 2) * 64 + lane;
            const uint4 qh = pa[0], ql = pa[64];
            const f16x8 wh8 = *reinterpret_cast<const f16x8 *>(&qh);
            const f16x8 wl8 = *reinterpret_cast<const f16x8 *>(&ql);
#pragma unroll
            for (int m = 0; m < MW; ++m) {
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh8, bfr[m], acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl8, bfr[m], acc[m][n], 0, 0, 0);
            }
        }
    }
    unsigned char *gout = reinterpret_cast<unsigned char *>(out) + (size_t)e * ncells * rowg;
#pragma unroll
    for (int n = 0; n < NW; ++n) {
        const int cb0 = co_base + 64 * wn + 32 * n + 4 * lh;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int cb = cb0 + 8 * g4;
            const float4 b4 = *reinterpret_cast<const float4 *>(P.stem_b + cb);
            const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int m = 0; m < MW; ++m) {
                const int r = (MW * wm + m) * 32 + li;
                if (r < ncells) {
                    f16x4 h4, l4;
                    float vv[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = fmaxf(acc[m][n][4 * g4 + j] + bv[j], 0.0f);
                        vv[j] = v;
                        if (AZX_SAT_TRACK) satmax = fmaxf(satmax, v);
                        _Float16 hi, lo;
                        split_f16(v, hi, lo);
                        h4[j] = hi;
                        l4[j] = lo;
                    }
                    *reinterpret_cast<f16x4 *>(gout + (size_t)r * rowg + cb * 2) = h4;
                    *reinterpret_cast<f16x4 *>(gout + (size_t)r * rowg + (size_t)C * 2 + cb * 2) = l4;
                    if (out32)
                        *reinterpret_cast<float4 *>(out32 + ((size_t)e * ncells + r) * C + cb) =
                            make_float4(vv[0], vv[1], vv[2], vv[3]);
                }
            }
        }
    }
    if (satmax > 65504.0f) atomicOr(P.sat_flag, 1u);     // an activation left the f16 range: its hi half is +inf (NetDev::sat_flag)
}

// ============================================================================================
// generic VALU fallback (any channel count): one thread per output element, activations in HBM
// ============================================================================================
__global__ void k_stem_generic(NetDev P, const uint8_t *ev_board, const int32_t *n_eval_ptr,
                               int n_eval_host, float *out) {
    const int n_eval = n_eval_ptr ? *n_eval_ptr : n_eval_host;
    const int C = P.C, N = P.N, ncells = P.ncells;
    const size_t total = (size_t)n_eval * ncells * C;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int co = idx % C;
        const int pos = (idx / C) % ncells;
        const size_t e = idx / ((size_t)C * ncells);
        const uint8_t *bd = ev_board + e * AZX_CELL_STRIDE;
        const int y = pos / N, x = pos - y * N;
        float acc = P.stem_b[co];
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            if (yy >= 0 && yy < N && xx >= 0 && xx < N)
                acc += P.stemT[(tap * 3 + bd[yy * N + xx]) * C + co];
        }
        out[idx] = fmaxf(acc, 0.0f);
    }
}

__global__ void k_conv_generic(NetDev P, int layer, const float *in, const float *residual,
                               const int32_t *n_eval_ptr, int n_eval_host, float *out) {
    const int n_eval = n_eval_ptr ? *n_eval_ptr : n_eval_host;
    const int C = P.C, N = P.N, ncells = P.ncells;
    const float *w = P.Wg + (size_t)layer * 9 * C * C;
    const size_t total = (size_t)n_eval * ncells * C;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int co = idx % C;
        const int pos = (idx / C) % ncells;
        const size_t e = idx / ((size_t)C * ncells);
        const int y = pos / N, x = pos - y * N;
        float acc = 0.0f;
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            if (yy < 0 || yy >= N || xx < 0 || xx >= N) continue;
            const float *ip = in + (e * ncells + yy * N + xx) * C;
            const float *wp = w + (size_t)tap * C * C + co;
            for (int ci = 0; ci < C; ++ci) acc += ip[ci] * wp[(size_t)ci * C];
        }
        float v = acc + P.bias[layer * C + co];
        if (residual) v += residual[idx];
        out[idx] = fmaxf(v, 0.0f);
    }
}

// ============================================================================================
// heads: value (conv1x1 C->2, BN, ReLU, FC 2N^2->64, ReLU, FC 64->1, tanh) and policy
// (conv1x1 C->4, BN, ReLU, FC 4N^2->N^2) -- network.py:77-84, :146; then, for the search, the
// masked softmax over the legal (= empty) cells, written by ORIGINAL cell index.
// One 192-thread block per board.
// ============================================================================================
// ============================================================================================
// k_heads_mfma: the heads' fully connected layers as fp32 MFMA GEMMs over a tile of boards.
// value_fc2 (2 n^2 -> 64) and move_fc (4 n^2 -> n^2) (network.py:79, :146) are 148 kFLOP per board -- 6 GFLOP per
// leaf batch of 40 960 -- which k_heads runs as scalar FMA chains (0.23 ms per batch, 2.9 % of a configs[2]
// move).  Here a 256-thread block takes 16 boards: their six head planes (from the fused tower, `hfeat`) are
// staged in LDS as the A operand ([board][k], row stride = 4 mod 64 floats: conflict-free ds_read_b128; 49 KB, so
// three blocks share a CU and hide each other's staging and softmax), the weights come from L2 pre-packed in
// B-fragment order, wave w owns output tiles 2w, 2w + 1 of move_fc and tile w of value_fc2
// (v_mfma_f32_16x16x4_f32: exact fp32 products).  A row of the product depends on its own board only, so a
// board's outputs do not depend on what else is in the tile.  Then value_fc3 + tanh and the masked softmax
// exactly as in k_heads.
// ============================================================================================
#define HM_MB 16
__global__ __launch_bounds__(256, 3) void k_heads_mfma(NetDev P, const float *__restrict__ hfeat,
                                                       const uint8_t *__restrict__ ev_board,
                                                       const int32_t *__restrict__ ev_flip,
                                                       const int32_t *__restrict__ n_eval_ptr, int n_eval_host,
                                                       float *__restrict__ logit_out, float *__restrict__ value_out,
                                                       float *__restrict__ prior_out) {
    extern __shared__ __align__(16) float hsm[];
    const int n_eval = n_eval_ptr ? *n_eval_ptr : n_eval_host;
    const int e0 = blockIdx.x * HM_MB;
    if (e0 >= n_eval) return;
    const int nb = min(HM_MB, n_eval - e0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int N = P.N, ncells = P.ncells, LDA = P.hm_lda;
    const int KV = 2 * ncells, KP = 4 * ncells, KVp = (KV + 15) & ~15, KPp = (KP + 15) & ~15;
    const int QV = KVp / 16, QP = KPp / 16, nf = 6 * ncells;
    float *A = hsm;                                   // [16][LDA]: value inputs at 0, policy inputs at KVp
    // what the softmax at the end needs from HBM -- the boards' empties and flip flags of this wave's four
    // boards -- is requested now, far ahead of its use
    uint8_t cellv[HM_MB / 4][2];
    int flipv[HM_MB / 4];
#pragma unroll
    for (int bi = 0; bi < HM_MB / 4; ++bi) {
        const int b = wave + 4 * bi, e = e0 + (b < nb ? b : 0);
        cellv[bi][0] = ev_board[(size_t)e * AZX_CELL_STRIDE + lane];
        cellv[bi][1] = ev_board[(size_t)e * AZX_CELL_STRIDE + 64 + lane];
        flipv[bi] = ev_flip[e];
    }
    // ---- stage the tile: the boards' 6 n^2 floats are one contiguous run of float2 (a board is 3 n^2 of them: 8-byte
    // aligned whatever n); all of a thread's loads in flight before the first LDS write.  Padding columns and
    // missing boards are zero.
    {
        const int nf2 = 3 * ncells, total = nb * nf2;
        const float2 *src = reinterpret_cast<const float2 *>(hfeat + (size_t)e0 * nf);
        constexpr int INFL = 24;
        for (int f0 = 0; f0 < total; f0 += 256 * INFL) {
            float2 v[INFL];
#pragma unroll
            for (int j = 0; j < INFL; ++j) {
                const int f = f0 + tid + 256 * j;
                v[j] = src[f < total ? f : total - 1];
            }
#pragma unroll
            for (int j = 0; j < INFL; ++j) {
                const int f = f0 + tid + 256 * j;
                if (f < total) {
                    const int b = f / nf2, i = 2 * (f - b * nf2);
                    *reinterpret_cast<float2 *>(A + (size_t)b * LDA + (i < KV ? i : KVp + (i - KV))) = v[j];
                }
            }
        }
        for (int b = wave; b < HM_MB; b += 4) {
            float *row = A + (size_t)b * LDA;
            if (b < nb) {
                if (lane < KVp - KV) row[KV + lane] = 0.0f;
                if (lane < KPp - KP) row[KVp + KP + lane] = 0.0f;
            } else {
                for (int i = lane; i < KVp + KPp; i += 64) row[i] = 0.0f;
            }
        }
    }
    __syncthreads();
    // ---- move_fc: output tiles 2 wave, 2 wave + 1 (16 logits each), K = 4 n^2; k = 16 t + 4 lk + s of group t ----
    const int NTP = (ncells + 15) / 16;               // <= 8 (n <= 11)
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    f32x4_ accp[2] = {f32x4_{0.f, 0.f, 0.f, 0.f}, f32x4_{0.f, 0.f, 0.f, 0.f}};
    constexpr int AH = 6;                             // weight fragments this many groups ahead
    {
        const int t0 = min(2 * wave, NTP - 1), t1 = min(2 * wave + 1, NTP - 1);
        const float4 *w0 = reinterpret_cast<const float4 *>(P.hmP) + (size_t)t0 * QP * 64 + lane;
        const float4 *w1 = reinterpret_cast<const float4 *>(P.hmP) + (size_t)t1 * QP * 64 + lane;
        const float *ap = A + (size_t)li * LDA + KVp + 4 * lk;
        float4 bq0[AH], bq1[AH];
#pragma unroll
        for (int u = 0; u < AH; ++u) { bq0[u] = w0[(size_t)min(u, QP - 1) * 64]; bq1[u] = w1[(size_t)min(u, QP - 1) * 64]; }
        if (2 * wave < NTP)
        for (int q0 = 0; q0 < QP; q0 += AH) {
#pragma unroll
            for (int u = 0; u < AH; ++u) {
                const int q = q0 + u;
                const float4 b0 = bq0[u], b1 = bq1[u];
                bq0[u] = w0[(size_t)min(q + AH, QP - 1) * 64];
                bq1[u] = w1[(size_t)min(q + AH, QP - 1) * 64];
                if (q < QP) {
                    const float4 a4 = *reinterpret_cast<const float4 *>(ap + 16 * q);
                    accp[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b0.x, accp[0], 0, 0, 0);
                    accp[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b1.x, accp[1], 0, 0, 0);
                    accp[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b0.y, accp[0], 0, 0, 0);
                    accp[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b1.y, accp[1], 0, 0, 0);
                    accp[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b0.z, accp[0], 0, 0, 0);
                    accp[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b1.z, accp[1], 0, 0, 0);
                    accp[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b0.w, accp[0], 0, 0, 0);
                    accp[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b1.w, accp[1], 0, 0, 0);
                }
            }
        }
    }
    // ---- value_fc2: output tile = wave (16 of the 64 units), K = 2 n^2 ---------------------------------------
    f32x4_ accv = f32x4_{0.f, 0.f, 0.f, 0.f};
    {
        const float4 *wv = reinterpret_cast<const float4 *>(P.hmV) + (size_t)wave * QV * 64 + lane;
        const float *ap = A + (size_t)li * LDA + 4 * lk;
        float4 bq[AH];
#pragma unroll
        for (int u = 0; u < AH; ++u) bq[u] = wv[(size_t)min(u, QV - 1) * 64];
        for (int qb = 0; qb < QV; qb += AH) {
#pragma unroll
            for (int u = 0; u < AH; ++u) {
                const int q = qb + u;
                const float4 b4 = bq[u];
                bq[u] = wv[(size_t)min(q + AH, QV - 1) * 64];
                if (q < QV) {
                    const float4 a4 = *reinterpret_cast<const float4 *>(ap + 16 * q);
                    accv = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, accv, 0, 0, 0);
                    accv = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, accv, 0, 0, 0);
                    accv = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, accv, 0, 0, 0);
                    accv = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, accv, 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();                                  // every wave is done reading the feature tile: its memory is reused
    float (*lg)[AZX_CELL_STRIDE] = reinterpret_cast<float (*)[AZX_CELL_STRIDE]>(hsm);            // [16][192] logits
    float (*h2)[64] = reinterpret_cast<float (*)[64]>(hsm + HM_MB * AZX_CELL_STRIDE);            // [16][64] fc2 sums
    // C/D layout of 16x16x4: col = lane & 15 (output unit), row = 4 (lane >> 4) + reg (board)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int j = 16 * (2 * wave + tt) + li;
        if (2 * wave + tt < NTP && j < ncells) {
            const float bias = P.mfcb[j];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = 4 * lk + r;
                const float logit = accp[tt][r] + bias;
                lg[b][j] = logit;
                if (b < nb) logit_out[(size_t)(e0 + b) * AZX_CELL_STRIDE + j] = logit;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) h2[4 * lk + r][16 * wave + li] = accv[r];
    __syncthreads();
    if (tid < nb) {                                   // + bias, ReLU, value_fc3 + tanh (network.py:79-81)
        float acc = 0.f;
        for (int i = 0; i < 64; ++i) acc += fmaxf(h2[tid][i] + P.fc2b[i], 0.f) * P.fc3w[i];
        value_out[e0 + tid] = tanhf(acc + P.fc3b[0]);
    }
    if (!prior_out) return;
    // masked softmax over the legal cells of the network-frame board (network.py:147-151), prior =
    // exp(log_softmax) (mcts.py:210): one wavefront per board, cells lane, lane + 64 (n <= 11)
#pragma unroll
    for (int bi = 0; bi < HM_MB / 4; ++bi) {
        const int b = wave + 4 * bi;
        if (b >= nb) break;
        const int e = e0 + b;
        float x[2];
        bool legal[2];
        float mx = -INFINITY;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int cell = s2 * 64 + lane;
            legal[s2] = cell < ncells && cellv[bi][s2] == 0;
            x[s2] = legal[s2] ? lg[b][cell] : -INFINITY;
            mx = fmaxf(mx, x[s2]);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) sum += legal[s2] ? expf(x[s2] - mx) : 0.f;
        const float lse = mx + logf(wave_sum(sum));
        const int flip = flipv[bi];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int cell = s2 * 64 + lane;
            if (cell < ncells) {
                int oc = cell;
                if (flip) {                            // back to the mover's frame (hex.py:107-111)
                    const int i = cell / N, j = cell - i * N;
                    oc = (N - 1 - j) * N + (N - 1 - i);
                }
                prior_out[(size_t)e * AZX_CELL_STRIDE + oc] = legal[s2] ? expf(x[s2] - lse) : 0.f;
            }
        }
    }
}

#ifndef HEADS_BPB
#define HEADS_BPB 8
#endif
#ifndef HEADS_KSPLIT
#define HEADS_KSPLIT 1
#endif
__global__ __launch_bounds__(192 * HEADS_KSPLIT) void k_heads(NetDev P, const float *__restrict__ act,
                                               const float *__restrict__ hfeat,
                                               const uint8_t *__restrict__ ev_board,
                                               const int32_t *__restrict__ ev_flip,
                                               const int32_t *__restrict__ n_eval_ptr,
                                               int n_eval_host, float *__restrict__ logit_out,
                                               float *__restrict__ value_out,
                                               float *__restrict__ prior_out) {
    // HEADS_BPB boards per block share every FC weight load (the policy FC matrix alone is 234 KB).  8 is
    // the measured optimum: 12 / 16 boards per block halve the weight traffic out of the L2 but leave fewer
    // blocks to hide each thread's serial 484-step accumulation (+0.14 / +0.25 ms per launch, round 2)
    extern __shared__ __align__(16) float hsm[];
    const int n_eval = n_eval_ptr ? *n_eval_ptr : n_eval_host;
    const int e0 = blockIdx.x * HEADS_BPB;
    if (e0 >= n_eval) return;
    const int nb = min(HEADS_BPB, n_eval - e0);
    const int tid = threadIdx.x;
    // the two fully connected layers can be split over HEADS_KSPLIT thread groups along their input
    // dimension (group 0 adds the partial sums and finishes); measured 1 = 2 = 3 within noise (round 2: the
    // kernel is bound neither by the L2 weight traffic nor by the length of a thread's accumulation chain), so 1
    const int part = tid / 192, t = tid - 192 * part;
    const int C = P.C, N = P.N, ncells = P.ncells;
    typedef float (*rowsB)[HEADS_BPB];
    rowsB hv = reinterpret_cast<rowsB>(hsm);                                  // [2 ncells][board]
    rowsB hp = reinterpret_cast<rowsB>(hsm + 2 * ncells * HEADS_BPB);         // [4 ncells][board]
    rowsB h2 = reinterpret_cast<rowsB>(hsm + 6 * ncells * HEADS_BPB);         // [64][board]
    float (*lg)[AZX_CELL_STRIDE] = reinterpret_cast<float (*)[AZX_CELL_STRIDE]>(hsm + (6 * ncells + 64) * HEADS_BPB);
    float *psum = hsm + (6 * ncells + 64) * HEADS_BPB + HEADS_BPB * AZX_CELL_STRIDE;   // [KSPLIT-1][(64 + 192) * BPB]

    // ---- 1x1 convs + folded BN + ReLU (network.py:77, :83); flatten order (c, h, w) -----------
    if (hfeat != nullptr) {                           // already done by the tower kernel
        // board by board, a thread's features of a board requested together (no index division, no
        // load-wait-store chain per element)
        const int nf = 6 * ncells;
        constexpr int NTH = 192 * HEADS_KSPLIT;
        constexpr int NFT = (6 * AZX_MAX_CELLS + NTH - 1) / NTH;
#pragma unroll 2
        for (int b = 0; b < HEADS_BPB; ++b) {
            float v[NFT];
            const float *src = hfeat + (size_t)(e0 + (b < nb ? b : 0)) * nf;
#pragma unroll
            for (int j = 0; j < NFT; ++j) {
                const int f = tid + NTH * j;
                v[j] = src[f < nf ? f : nf - 1];
            }
#pragma unroll
            for (int j = 0; j < NFT; ++j) {
                const int f = tid + NTH * j;
                if (f < nf) {
                    const float x = b < nb ? v[j] : 0.0f;
                    if (f < 2 * ncells) hv[f][b] = x; else hp[f - 2 * ncells][b] = x;
                }
            }
        }
    } else if (tid < ncells) {
        for (int b = 0; b < HEADS_BPB; ++b) {
            float v0 = 0.f, v1 = 0.f, p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
            if (b < nb) {
                const float4 *a = reinterpret_cast<const float4 *>(act + ((size_t)(e0 + b) * ncells + tid) * C);
                for (int c4 = 0; c4 < C / 4; ++c4) {
                    const float4 x = a[c4];
                    const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c = 4 * c4 + j;
                        v0 += xs[j] * P.wv[c];
                        v1 += xs[j] * P.wv[C + c];
                        p0 += xs[j] * P.wp[c];
                        p1 += xs[j] * P.wp[C + c];
                        p2 += xs[j] * P.wp[2 * C + c];
                        p3 += xs[j] * P.wp[3 * C + c];
                    }
                }
                for (int c = (C / 4) * 4; c < C; ++c) {      // channel counts not divisible by 4
                    const float x = act[((size_t)(e0 + b) * ncells + tid) * C + c];
                    v0 += x * P.wv[c]; v1 += x * P.wv[C + c];
                    p0 += x * P.wp[c]; p1 += x * P.wp[C + c]; p2 += x * P.wp[2 * C + c]; p3 += x * P.wp[3 * C + c];
                }
            }
            hv[tid][b] = fmaxf(v0 + P.bv[0], 0.f);
            hv[ncells + tid][b] = fmaxf(v1 + P.bv[1], 0.f);
            hp[tid][b] = fmaxf(p0 + P.bp[0], 0.f);
            hp[ncells + tid][b] = fmaxf(p1 + P.bp[1], 0.f);
            hp[2 * ncells + tid][b] = fmaxf(p2 + P.bp[2], 0.f);
            hp[3 * ncells + tid][b] = fmaxf(p3 + P.bp[3], 0.f);
        }
    }
    __syncthreads();
    float acc2[HEADS_BPB], accm[HEADS_BPB];
#pragma unroll
    for (int b = 0; b < HEADS_BPB; ++b) { acc2[b] = 0.f; accm[b] = 0.f; }
    if (t < 64) {                                     // value_fc2 (network.py:79): this group's share of the inputs
        const int i0 = (2 * ncells * part) / HEADS_KSPLIT, i1 = (2 * ncells * (part + 1)) / HEADS_KSPLIT;
#pragma unroll 8                                     // eight weight loads in flight per thread
        for (int i = i0; i < i1; ++i) {
            const float w = P.fc2T[i * 64 + t];
#pragma unroll
            for (int b4 = 0; b4 < HEADS_BPB / 4; ++b4) {
                const float4 x = *reinterpret_cast<const float4 *>(&hv[i][4 * b4]);
                acc2[4 * b4] += x.x * w; acc2[4 * b4 + 1] += x.y * w; acc2[4 * b4 + 2] += x.z * w; acc2[4 * b4 + 3] += x.w * w;
            }
        }
    }
    if (t < ncells) {                                 // move_fc (network.py:146)
        const int i0 = (4 * ncells * part) / HEADS_KSPLIT, i1 = (4 * ncells * (part + 1)) / HEADS_KSPLIT;
#pragma unroll 8
        for (int i = i0; i < i1; ++i) {
            const float w = P.mfcT[(size_t)i * AZX_CELL_STRIDE + t];
#pragma unroll
            for (int b4 = 0; b4 < HEADS_BPB / 4; ++b4) {
                const float4 x = *reinterpret_cast<const float4 *>(&hp[i][4 * b4]);
                accm[4 * b4] += x.x * w; accm[4 * b4 + 1] += x.y * w; accm[4 * b4 + 2] += x.z * w; accm[4 * b4 + 3] += x.w * w;
            }
        }
    }
    if (HEADS_KSPLIT > 1) {
        if (part > 0) {
            float *ps = psum + (size_t)(part - 1) * (64 + 192) * HEADS_BPB;
#pragma unroll
            for (int b = 0; b < HEADS_BPB; ++b) {
                if (t < 64) ps[b * 64 + t] = acc2[b];
                if (t < ncells) ps[64 * HEADS_BPB + b * 192 + t] = accm[b];
            }
        }
        __syncthreads();
        if (part == 0) {
            for (int pp = 0; pp < HEADS_KSPLIT - 1; ++pp) {
                const float *ps = psum + (size_t)pp * (64 + 192) * HEADS_BPB;
#pragma unroll
                for (int b = 0; b < HEADS_BPB; ++b) {
                    if (t < 64) acc2[b] += ps[b * 64 + t];
                    if (t < ncells) accm[b] += ps[64 * HEADS_BPB + b * 192 + t];
                }
            }
        }
    }
    if (part == 0 && t < 64) {                        // + bias, ReLU
#pragma unroll
        for (int b = 0; b < HEADS_BPB; ++b) h2[t][b] = fmaxf(acc2[b] + P.fc2b[t], 0.f);
    }
    if (part == 0 && t < ncells) {
        const float bias = P.mfcb[t];
#pragma unroll
        for (int b = 0; b < HEADS_BPB; ++b) {
            const float logit = accm[b] + bias;
            lg[b][t] = logit;
            if (b < nb) logit_out[(size_t)(e0 + b) * AZX_CELL_STRIDE + t] = logit;
        }
    }
    __syncthreads();
    if (tid < nb) {                                   // value_fc3 + tanh (network.py:80-81)
        float acc = 0.f;
        for (int i = 0; i < 64; ++i) acc += h2[i][tid] * P.fc3w[i];
        value_out[e0 + tid] = tanhf(acc + P.fc3b[0]);
    }
    if (!prior_out) return;
    // masked softmax over the legal cells of the network-frame board (network.py:147-151),
    // prior = exp(log_softmax) (mcts.py:210); one wavefront per board, cells lane, lane+64, lane+128
    const int lane = tid & 63, wave = tid >> 6;
    for (int b = wave; b < nb; b += 3 * HEADS_KSPLIT) {
        const int e = e0 + b;
        float x[3];
        bool legal[3];
        float mx = -INFINITY;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int cell = s * 64 + lane;
            legal[s] = cell < ncells && ev_board[(size_t)e * AZX_CELL_STRIDE + cell] == 0;
            x[s] = legal[s] ? lg[b][cell] : -INFINITY;
            mx = fmaxf(mx, x[s]);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < 3; ++s) sum += legal[s] ? expf(x[s] - mx) : 0.f;
        const float lse = mx + logf(wave_sum(sum));
        const int flip = ev_flip[e];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int cell = s * 64 + lane;
            if (cell < ncells) {
                int oc = cell;
                if (flip) {                            // back to the mover's frame (hex.py:107-111)
                    const int i = cell / N, j = cell - i * N;
                    oc = (N - 1 - j) * N + (N - 1 - i);
                }
                prior_out[(size_t)e * AZX_CELL_STRIDE + oc] = legal[s] ? expf(x[s] - lse) : 0.f;
            }
        }
    }
}

// ============================================================================================
// host side
// ============================================================================================


// Every tower kernel takes its LDS image as dynamic shared memory above the 64 KB default: the limit is raised to
// the CU's 160 KB for all of them whenever a network is created (per device and idempotent; a process-wide "done"
// flag would leave a second device, or a later network with a larger image, on the first one's setting -- which
// ROCm 7.2 happens not to enforce, but the contract is the opt-in).
static int raise_lds_limits() {
    const int cap = 160 * 1024;
    const void *fns[] = {(const void *)k_tower_f16x3_s16, (const void *)k_conv_wide_f16x3_s16,
                         (const void *)k_tower_mfma<64, 4, 2, 1, 2>, (const void *)k_tower_mfma<64, 6, 1, 2, 2>,
                         (const void *)k_tower_mfma<32, 6, 2, 2, 1>, (const void *)k_heads_mfma, (const void *)k_heads};
    for (const void *f : fns)
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, cap) != hipSuccess)
            return nfail(AZX_EHIP, "net: raising a tower kernel's dynamic LDS limit failed");
    return AZX_OK;
}

int azx_net_create(AzxNet **out, int N, int blocks, int chans, int max_evals, hipStream_t st) {
    if (N < 2 || N > AZX_MAX_BOARD) return nfail(AZX_EINVAL, "net: board size out of range");
    if (blocks < 0 || chans < 1) return nfail(AZX_EINVAL, "net: bad num_blocks/base_chans");
    if (int rc = raise_lds_limits()) return rc;
    AzxNet *net = new AzxNet();
    memset(&net->d, 0, sizeof net->d);
    net->d.N = N;
    net->d.ncells = N * N;
    net->d.C = chans;
    net->d.blocks = blocks;
    net->d.layers = 2 * blocks;
    net->max_evals = max_evals;
    net->stream = st;
    const int ncells = N * N;
    // pick the fused MFMA tower when its tiling covers (C, N)
    const char *force = getenv("AZX_TOWER");
    const bool want_fp32 = force && !strcmp(force, "fp32");
    if (chans == 64 && ncells <= 121 && blocks >= 1 && !want_fp32) net->tower_variant = 4;   // k_tower_f16x3
    else if (chans == 64 && ncells <= 128) net->tower_variant = 1;   // <64,4,2,1,2>
    else if (chans == 64 && ncells <= 192) net->tower_variant = 2;   // <64,6,1,2,2>
    else if (chans == 32 && ncells <= 192) net->tower_variant = 3;   // <32,6,2,2,1>
    else if (chans % 128 == 0 && ncells <= 192 && !want_fp32) net->tower_variant = 5;   // k_conv_wide_f16x3 per layer
    net->use_mfma = net->tower_variant != 0;
    { const char *v = getenv("AZX_WIDE_STREAMS"); net->opt_wsplit = std::min(4, std::max(1, v ? atoi(v) : 2)); }
    { const char *v = getenv("AZX_HEADS"); net->opt_heads_mfma = !(v && !strcmp(v, "valu")); }
    { const char *v = getenv("AZX_PACK"); net->pack_on_host = v && !strcmp(v, "host"); }
    {
        char b[260];
        const char *tower = "k_stem_generic + k_conv_generic (VALU)";
        if (net->tower_variant == 4) tower = "k_tower_f16x3_s16";
        else if (net->tower_variant == 5) tower = "k_stem_wide_f16x3 + k_conv_wide_f16x3_s16 per layer";
        else if (net->tower_variant == 1) tower = "k_tower_mfma<64,4,2,1,2> (fp32 MFMA)";
        else if (net->tower_variant == 2) tower = "k_tower_mfma<64,6,1,2,2> (fp32 MFMA)";
        else if (net->tower_variant == 3) tower = "k_tower_mfma<32,6,2,2,1> (fp32 MFMA)";
        const bool hm = net->tower_variant == 4 && net->opt_heads_mfma && ncells <= 128;
        snprintf(b, sizeof b, "%s + %s (%dx%d on %dx%d; AZX_TOWER=%s AZX_WIDE_STREAMS=%d AZX_HEADS=%s AZX_PACK=%s)",
                 tower, hm ? "k_heads_mfma" : "k_heads", blocks, chans, N, N, want_fp32 ? "fp32" : "default",
                 net->opt_wsplit, net->opt_heads_mfma ? "mfma" : "valu", net->pack_on_host ? "host" : "device");
        net->info = b;
    }
    const size_t E = max_evals;
    net->act = nalloc<float>(net, E * ncells * chans);
    net->logit = nalloc<float>(net, E * AZX_CELL_STRIDE);
    if (net->tower_variant == 4) net->hfeat = nalloc<float>(net, E * 6 * ncells);
    net->hb_board = nalloc<uint8_t>(net, E * AZX_CELL_STRIDE);
    net->hb_flip = nalloc<int32_t>(net, E);
    net->hb_value = nalloc<float>(net, E);
    if (!net->use_mfma) {
        net->act2 = nalloc<float>(net, E * ncells * chans);
        net->act3 = nalloc<float>(net, E * ncells * chans);
    }
    if (net->tower_variant == 5) {
        net->wideX = nalloc<unsigned short>(net, E * ncells * chans * 2);
        net->wideY = nalloc<unsigned short>(net, E * ncells * chans * 2);
    }
    if (!net->act || !net->logit || !net->hb_board || !net->hb_flip || !net->hb_value ||
        (!net->use_mfma && (!net->act2 || !net->act3)) || (net->tower_variant == 5 && (!net->wideX || !net->wideY))) {
        azx_net_destroy(net);
        return nfail(AZX_ENOMEM, "net: hipMalloc failed");
    }
    const int bpb = net->tower_variant == 2 ? 1 : 2;
    net->lds_bytes = (size_t)bpb * 2 * (ncells + 1) * (chans + 4) * sizeof(float);
    if (net->tower_variant == 4) net->lds_bytes = (size_t)F16X3_BPB * 128 * 272 + 2 * 272;   // boards + the zero rows (two in k_tower_f16x3_s16)
    if (net->tower_variant == 5) net->lds_bytes = (size_t)(ncells + 2) * WIDE_ROWB;   // two zero rows + the chunk image
    net->d.sat_flag = nalloc<uint32_t>(net, 4);
    if (!net->d.sat_flag || hipDeviceSynchronize() != hipSuccess) {     // (the zero fills ran on the null stream)
        azx_net_destroy(net);
        return nfail(AZX_ENOMEM, "net: hipMalloc failed");
    }
    *out = net;
    return AZX_OK;
}

void azx_net_set_stream(AzxNet *net, hipStream_t st, const uint32_t *mask, int words) {
    if (!net) return;
    for (int i = 0; i < 3; ++i) {
        if (net->stream2[i]) { (void)hipStreamSynchronize(net->stream2[i]); (void)hipStreamDestroy(net->stream2[i]); }
        if (net->ev_join[i]) (void)hipEventDestroy(net->ev_join[i]);
        net->stream2[i] = nullptr;
        net->ev_join[i] = nullptr;
    }
    if (net->ev_fork) (void)hipEventDestroy(net->ev_fork);
    net->ev_fork = nullptr;           // run_net makes the side streams again on first use
    net->streams_ok = false;
    net->cu_mask.assign(mask, mask + (mask ? words : 0));
    net->stream = st;
}

void azx_net_destroy(AzxNet *net) {
    if (net) {
        for (int i = 0; i < 3; ++i) {
            if (net->stream2[i]) { (void)hipStreamSynchronize(net->stream2[i]); (void)hipStreamDestroy(net->stream2[i]); }
            if (net->ev_join[i]) (void)hipEventDestroy(net->ev_join[i]);
            net->stream2[i] = nullptr;
            net->ev_join[i] = nullptr;
        }
        if (net->ev_fork) (void)hipEventDestroy(net->ev_fork);
        net->ev_fork = nullptr;
    }
#ifdef AZX_NET_STAMP
    {
        unsigned long long h[10] = {0};
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wide_stamp), sizeof h) == hipSuccess && h[7]) {
            static const char *nm[6] = {"prologue (bias + residual, setup)", "barrier after staging", "k-loops (MFMA)", "epilogue (stores acknowledged)",
                                        "barrier before staging (skew)", "staging: requests, round trips, LDS writes"};
            unsigned long long tot = 0;
            for (int r = 0; r < 6; ++r) tot += h[r];
            fprintf(stderr, "k_conv_wide_f16x3_s16 stamps over %llu waves: %.0f cycles/wave, shader clock %.0f MHz during the kernel\n",
                    h[7], (double)tot / h[7], h[8] ? 100.0 * (double)tot / (double)h[8] : 0.0);
            for (int r = 0; r < 6; ++r)
                fprintf(stderr, "  %-44s %6.1f%%  %9.0f cycles/wave\n", nm[r], 100.0 * h[r] / tot, (double)h[r] / h[7]);
            unsigned long long z[10] = {0};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wide_stamp), z, sizeof z);
        }
    }
    {
        unsigned long long h[10] = {0};
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tower_stamp), sizeof h) == hipSuccess && h[7]) {
            static const char *nm[7] = {"stem", "residual load/setup", "k-loop (MFMA)", "barrier after k-loop",
                                        "epilogue", "barrier after epilogue", "output store"};
            unsigned long long tot = 0;
            for (int r = 0; r < 7; ++r) tot += h[r];
            fprintf(stderr, "k_tower_f16x3_s16 stamps over %llu blocks: %.0f cycles/block, shader clock %.0f MHz during the kernel\n",
                    h[7], (double)tot / h[7], h[8] ? 100.0 * (double)tot / (double)h[8] : 0.0);
            for (int r = 0; r < 7; ++r)
                fprintf(stderr, "  %-24s %6.1f%%  %9.0f cycles/block\n", nm[r], 100.0 * h[r] / tot, (double)h[r] / h[7]);
            {
                std::vector<unsigned long long> tr(3 * 32768);
                (void)hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_tower_trace), tr.size() * sizeof(unsigned long long));
                std::map<unsigned long long, std::vector<std::pair<unsigned long long, unsigned long long>>> cu;
                unsigned long long t0 = ~0ull, t1 = 0;
                size_t nblk = 0;
                for (int b = 0; b < 32768; ++b)
                    if (tr[3 * b + 2]) {
                        cu[tr[3 * b]].push_back({tr[3 * b + 1], tr[3 * b + 2]});
                        t0 = std::min(t0, tr[3 * b + 1]);
                        t1 = std::max(t1, tr[3 * b + 2]);
                        ++nblk;
                    }
                double busy = 0, mn = 1e30, mx = 0, lastend_min = 1e30, lastend_max = 0;
                size_t cmin = 1 << 30, cmax = 0;
                for (auto &kv : cu) {
                    double b = 0, le = 0;
                    for (auto &iv : kv.second) { b += (double)(iv.second - iv.first); le = std::max(le, (double)(iv.second - t0)); }
                    busy += b; mn = std::min(mn, b); mx = std::max(mx, b);
                    cmin = std::min(cmin, kv.second.size()); cmax = std::max(cmax, kv.second.size());
                    lastend_min = std::min(lastend_min, le); lastend_max = std::max(lastend_max, le);
                }
                fprintf(stderr, "  last launch: %zu blocks on %zu CUs, span %.1f us, blocks/CU %zu..%zu, mean resident blocks/CU %.2f, "
                        "per-CU block-time %.0f..%.0f us, CU finish %.0f..%.0f us\n", nblk, cu.size(), (t1 - t0) / 100.0, cmin, cmax,
                        busy / ((double)(t1 - t0) * cu.size()), mn / 100.0, mx / 100.0, lastend_min / 100.0, lastend_max / 100.0);
            }
            int nb = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_tower_f16x3_s16, F16X3_BPB * 128, net->lds_bytes);
            hipFuncAttributes fa;
            memset(&fa, 0, sizeof fa);
            (void)hipFuncGetAttributes(&fa, (const void *)k_tower_f16x3_s16);
            fprintf(stderr, "  runtime occupancy: %d blocks/CU (dynamic LDS %zu B, static %zu B, %d VGPRs, max threads %d)\n", nb,
                    net->lds_bytes, (size_t)fa.sharedSizeBytes, fa.numRegs, fa.maxThreadsPerBlock);
        }
    }
#endif
    if (!net) return;
    for (void *p : net->allocs) (void)hipFree(p);
    if (net->raw_tab_host) (void)hipHostFree((void *)net->raw_tab_host);
    if (net->wmax_host) (void)hipHostFree((void *)net->wmax_host);
    delete net;
}

bool azx_net_ready(const AzxNet *net) { return net && net->ready; }


// tower + heads over boards[0 .. n) (n read from n_eval_ptr on the device when given)
static void run_net(AzxNet *net, const uint8_t *boards, const int32_t *flip, const int32_t *n_eval_ptr,
                    int n_host, int max_n, float *logit, float *value, float *prior, hipStream_t st) {
    const NetDev &d = net->d;
    if (max_n <= 0) return;
    const float *hfeat = nullptr;    // set when the tower kernel already produced the heads' conv planes
    if (net->use_mfma) {
        const size_t lds = net->lds_bytes;
        if (net->tower_variant == 4) {
            const dim3 grid((max_n + F16X3_BPB - 1) / F16X3_BPB), block(F16X3_BPB * 128);
            hipLaunchKernelGGL(k_tower_f16x3_s16, grid, block, lds, st, d, boards, n_eval_ptr, n_host, (float *)nullptr, net->hfeat);
            hfeat = net->hfeat;
        } else if (net->tower_variant == 5) {
            const dim3 grid(max_n, d.C / 128), block(256);
            hipLaunchKernelGGL(k_stem_wide_f16x3, grid, block, 0, st, d, boards, net->wideX,
                               d.blocks == 0 ? net->act : (float *)nullptr, n_eval_ptr, n_host);
            // A layer is one launch over all boards, and a launch ends with a partly filled round of blocks
            // (9 964 blocks over 512 slots: 19.46 rounds, 2.7 % of a layer idle) -- 38 times per batch.  The
            // boards are independent, so the batch is cut into AZX_WIDE_STREAMS parts that run their 38 launches
            // on separate streams: one part's tail round fills up with the others' blocks, whichever layer
            // those are in.
            const int wsplit = net->opt_wsplit;
            {
                int parts = wsplit;
                if (parts > 1 && !net->ev_fork) {
                    bool ok = hipEventCreateWithFlags(&net->ev_fork, hipEventDisableTiming) == hipSuccess;
                    for (int i = 0; i < 3 && ok; ++i)
                        ok = (net->cu_mask.empty() ? hipStreamCreateWithFlags(&net->stream2[i], hipStreamNonBlocking)
                                                   : hipExtStreamCreateWithCUMask(&net->stream2[i], (uint32_t)net->cu_mask.size(), net->cu_mask.data())) == hipSuccess &&
                             hipEventCreateWithFlags(&net->ev_join[i], hipEventDisableTiming) == hipSuccess;
                    net->streams_ok = ok;
                }
                if (parts > 1 && !net->streams_ok) parts = 1;
                const int per = (((max_n + parts - 1) / parts + 7) / 8) * 8;     // boards per part, whole groups of 8
                if (per >= max_n) parts = 1;
                // the stem (and everything before it) is done; if the fork cannot be recorded or waited for, a
                // later part's convolutions could start before the stem has finished: fall back to one stream
                if (parts > 1 && hipEventRecord(net->ev_fork, st) != hipSuccess) parts = 1;
                for (int part = 1; part < parts; ++part)
                    if (hipStreamWaitEvent(net->stream2[part - 1], net->ev_fork, 0) != hipSuccess) { parts = 1; break; }
                const int per_ok = parts > 1 ? per : max_n;
                for (int part = 0; part < parts; ++part) {
                    hipStream_t s = part ? net->stream2[part - 1] : st;
                    const int e0 = part * per_ok, e1 = std::min(max_n, e0 + per_ok);
                    if (e1 <= e0) continue;
                    const dim3 g(8 * (d.C / 128), (e1 - e0 + 7) / 8);
                    for (int b = 0; b < d.blocks; ++b) {
                        hipLaunchKernelGGL(k_conv_wide_f16x3_s16, g, block, lds, s, d, 2 * b, (const unsigned short *)net->wideX, net->wideY,
                                           (const unsigned short *)nullptr, (float *)nullptr, n_eval_ptr, n_host, e0, e1);
                        hipLaunchKernelGGL(k_conv_wide_f16x3_s16, g, block, lds, s, d, 2 * b + 1, (const unsigned short *)net->wideY, net->wideX,
                                           (const unsigned short *)net->wideX, b == d.blocks - 1 ? net->act : (float *)nullptr, n_eval_ptr, n_host, e0, e1);
                    }
                    if (part) {     // the heads read every board; if the join cannot be expressed on the streams, block
                        if (hipEventRecord(net->ev_join[part - 1], s) != hipSuccess ||
                            hipStreamWaitEvent(st, net->ev_join[part - 1], 0) != hipSuccess)
                            (void)hipStreamSynchronize(s);
                    }
                }
            }
        } else if (net->tower_variant == 1) {
            hipLaunchKernelGGL((k_tower_mfma<64, 4, 2, 1, 2>), dim3((max_n + 1) / 2), dim3(256), lds, st, d, boards, n_eval_ptr, n_host, net->act);
        } else if (net->tower_variant == 2) {
            hipLaunchKernelGGL((k_tower_mfma<64, 6, 1, 2, 2>), dim3(max_n), dim3(256), lds, st, d, boards, n_eval_ptr, n_host, net->act);
        } else {
            hipLaunchKernelGGL((k_tower_mfma<32, 6, 2, 2, 1>), dim3((max_n + 1) / 2), dim3(256), lds, st, d, boards, n_eval_ptr, n_host, net->act);
        }
    } else {
        const int grid = 2048;
        hipLaunchKernelGGL(k_stem_generic, dim3(grid), dim3(256), 0, st, d, boards, n_eval_ptr, n_host, net->act);
        float *x = net->act, *y = net->act2, *z = net->act3;
        for (int b = 0; b < d.blocks; ++b) {
            hipLaunchKernelGGL(k_conv_generic, dim3(grid), dim3(256), 0, st, d, 2 * b, x, (const float *)nullptr, n_eval_ptr, n_host, y);
            hipLaunchKernelGGL(k_conv_generic, dim3(grid), dim3(256), 0, st, d, 2 * b + 1, y, x, n_eval_ptr, n_host, z);
            std::swap(x, z);
        }
        if (x != net->act)   // heads read net->act
            (void)hipMemcpyAsync(net->act, x, (size_t)max_n * d.ncells * d.C * sizeof(float), hipMemcpyDeviceToDevice, st);
    }
    if (hfeat != nullptr && net->opt_heads_mfma && d.ncells <= 128) {
        // the fused tower left the six head planes: the FC layers run as fp32 MFMA GEMMs over tiles of 32 boards
        const size_t hl = std::max((size_t)HM_MB * d.hm_lda, (size_t)HM_MB * AZX_CELL_STRIDE + (size_t)HM_MB * 64) * sizeof(float);
        hipLaunchKernelGGL(k_heads_mfma, dim3((max_n + HM_MB - 1) / HM_MB), dim3(256), hl, st, d, hfeat, boards, flip, n_eval_ptr, n_host, logit, value, prior);
    } else {
        const size_t hl = ((size_t)(6 * d.ncells + 64) * HEADS_BPB + (size_t)HEADS_BPB * AZX_CELL_STRIDE +
                           (size_t)(HEADS_KSPLIT - 1) * (64 + 192) * HEADS_BPB) * sizeof(float);
        hipLaunchKernelGGL(k_heads, dim3((max_n + HEADS_BPB - 1) / HEADS_BPB), dim3(192 * HEADS_KSPLIT), hl, st, d, net->act, hfeat, boards, flip, n_eval_ptr, n_host, logit, value, prior);
    }
}

const char *azx_net_kernel_info(const AzxNet *net) { return net->info.c_str(); }

// The split-f16 towers OR 1 into d.sat_flag when an activation left the f16 range (its `hi` half became +inf).
// Read where the host synchronises anyway; the flag is cleared so the next call starts clean.
int azx_net_check_range(AzxNet *net, hipStream_t st) {
    if (!net || !net->d.sat_flag) return AZX_OK;
    uint32_t flag = 0;
    if (hipMemcpyAsync(&flag, net->d.sat_flag, sizeof flag, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return nfail(AZX_EHIP, "net: reading the activation range flag failed");
    if (!flag) return AZX_OK;
    (void)hipMemsetAsync(net->d.sat_flag, 0, sizeof flag, st);
    return nfail(AZX_ERANGE, "net: an activation of the residual tower exceeded the f16 range (65504) of the split-f16 "
                             "kernels -- the evaluations of this call are not valid (AZX_TOWER=fp32 runs the fp32 MFMA tower)");
}

void azx_net_eval(AzxNet *net, const DevEngine &e, hipStream_t st) {
    run_net(net, e.ev_board, e.ev_flip, e.n_eval, 0, net->max_evals, net->logit, e.ev_value, e.ev_prior, st);
}

int azx_net_forward_host(AzxNet *net, int B, int K, const int32_t *boards, const int32_t *legal_moves,
                         float *value, float *logprob, hipStream_t st) {
    if (!net->ready) return nfail(AZX_ESTATE, "net: azx_set_weights has not been called");
    if (B < 0 || K < 0 || (B && (!boards || !legal_moves || !value || !logprob)))
        return nfail(AZX_EINVAL, "net: bad forward arguments");
    const NetDev &d = net->d;
    const int ncells = d.ncells;
    std::vector<uint8_t> hb((size_t)net->max_evals * AZX_CELL_STRIDE);
    std::vector<float> hl((size_t)net->max_evals * AZX_CELL_STRIDE), hv(net->max_evals);
    for (int b0 = 0; b0 < B; b0 += net->max_evals) {
        const int nb = std::min(net->max_evals, B - b0);
        std::fill(hb.begin(), hb.end(), 0);
        for (int i = 0; i < nb; ++i)
            for (int c = 0; c < ncells; ++c) {
                const int32_t v = boards[(size_t)(b0 + i) * ncells + c];
                if (v < 0 || v > 2) return nfail(AZX_EINVAL, "net: board values must be 0/1/2");
                hb[(size_t)i * AZX_CELL_STRIDE + c] = (uint8_t)v;
            }
        if (hipMemcpyAsync(net->hb_board, hb.data(), (size_t)nb * AZX_CELL_STRIDE, hipMemcpyHostToDevice, st) != hipSuccess)
            return nfail(AZX_EHIP, "net: H2D copy failed");
        run_net(net, net->hb_board, net->hb_flip, nullptr, nb, nb, net->logit, net->hb_value, nullptr, st);
        if (hipGetLastError() != hipSuccess) return nfail(AZX_EHIP, "net: kernel launch failed");
        if (hipMemcpyAsync(hl.data(), net->logit, (size_t)nb * AZX_CELL_STRIDE * sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(hv.data(), net->hb_value, (size_t)nb * sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
            return nfail(AZX_EHIP, "net: D2H copy failed");
        for (int i = 0; i < nb; ++i) {
            value[b0 + i] = hv[i];
            // gather + pad mask + log_softmax over the padded row (network.py:147-151)
            const int32_t *lm = legal_moves + (size_t)(b0 + i) * K;
            float *lp = logprob + (size_t)(b0 + i) * K;
            float mx = -INFINITY;
            for (int j = 0; j < K; ++j) {
                if (lm[j] < 0 || lm[j] > ncells) return nfail(AZX_EINVAL, "net: legal move out of range");
                const int tile = lm[j] > 0 ? lm[j] - 1 : 0;
                lp[j] = lm[j] == 0 ? -99.0f : hl[(size_t)i * AZX_CELL_STRIDE + tile];
                mx = std::max(mx, lp[j]);
            }
            double sum = 0.0;
            for (int j = 0; j < K; ++j) sum += std::exp((double)lp[j] - (double)mx);
            const float lse = (float)((double)mx + std::log(sum));
            for (int j = 0; j < K; ++j) lp[j] = lp[j] - lse;
        }
    }
    return AZX_OK;
}
