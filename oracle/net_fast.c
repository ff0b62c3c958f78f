/* oracle/net_fast.c -- the TIMED host path of bench.py's cpu_baseline: the same network as net.c with the
 * 3x3 convolutions blocked for the host's vector units.  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * net.c's direct convolution (the parity path, pinned against golden G3) streams a layer's whole weight tensor
 * once per board position; it runs at about a third of the per-core rate of the reference's own CPU path
 * (torch's conv2d), which made the baseline a straw man.  Here a block of PB consecutive
 * positions of one board row times OB output channels is accumulated in vector registers while the weights
 * stream by once per block, with fused multiply-adds; the activations live in a zero-padded (n+2) x (n+2) image so the
 * nine taps need no bounds tests.  Same operation order per output element as net.c (tap-major, then input
 * channel), so the two paths differ only by the FMA's single rounding (tests/test_oracle_golden.py holds them
 * together at 1e-5).  The compiled-in vector width is picked at load time (target_clones). */
#include "net_priv.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define OB 32     /* output channels per register block */
#define VL 16     /* floats per vector value (one zmm; two ymm on AVX2) */
#define PBMAX 11
typedef float vf __attribute__((vector_size(4 * VL)));

#if defined(__x86_64__)
#define CLONES __attribute__((target_clones("avx512f", "fma", "default")))
#else
#define CLONES
#endif

/* in / out / residual: padded images [(n+2)*(n+2) + PBMAX][channels], halo zero; w: [tap][cin][cout].
 * PB positions of one board row x OB output channels per register block (PB x 2 zmm accumulators); the loop over
 * output-channel tiles is outermost so that a tile's weights (9 cin OB floats: 295 KB at 256 channels) stay in
 * the core's L2 while the whole board streams past them. */
#define DEFINE_CONV(PB)                                                                                          \
CLONES static void conv3x3_blocked_##PB(int n, int cin, int cout, const float *in, const float *w,              \
                                        const float *scale, const float *shift, const float *residual,           \
                                        float *out) {                                                            \
    const int np = n + 2;                                                                                        \
    for (int o0 = 0; o0 < cout; o0 += OB) {                                                                      \
        const int ob = cout - o0 < OB ? cout - o0 : OB;                                                          \
        for (int y = 0; y < n; ++y)                                                                              \
            for (int x0 = 0; x0 < n; x0 += PB) {                                                                 \
                const int px = n - x0 < PB ? n - x0 : PB;                                                        \
                float acc[PB][OB];                                                                               \
                if (ob == OB) {                                                                                  \
                    vf va[PB][OB / VL];                                                                          \
                    for (int p = 0; p < PB; ++p)                                                                 \
                        for (int j = 0; j < OB / VL; ++j) va[p][j] = (vf){0};                                    \
                    for (int tap = 0; tap < 9; ++tap) {                                                          \
                        const float *ip = in + (size_t)((y + tap / 3) * np + x0 + tap % 3) * cin;                \
                        const float *wp = w + (size_t)tap * cin * cout + o0;                                     \
                        for (int i = 0; i < cin; ++i) {                                                          \
                            vf wv[OB / VL];                                                                      \
                            memcpy(wv, wp + (size_t)i * cout, sizeof wv);      /* unaligned vector loads */      \
                            _Pragma("GCC unroll 16")                                                             \
                            for (int p = 0; p < PB; ++p) {                                                       \
                                const float a = ip[(size_t)p * cin + i];                                         \
                                _Pragma("GCC unroll 2")                                                          \
                                for (int j = 0; j < OB / VL; ++j) va[p][j] += a * wv[j];                         \
                            }                                                                                    \
                        }                                                                                        \
                    }                                                                                            \
                    memcpy(acc, va, sizeof acc);                                                                 \
                } else {                                                                                         \
                    for (int p = 0; p < PB; ++p)                                                                 \
                        for (int o = 0; o < OB; ++o) acc[p][o] = 0.0f;                                           \
                    for (int tap = 0; tap < 9; ++tap) {                                                          \
                        const float *ip = in + (size_t)((y + tap / 3) * np + x0 + tap % 3) * cin;                \
                        const float *wp = w + (size_t)tap * cin * cout + o0;                                     \
                        for (int i = 0; i < cin; ++i)                                                            \
                            for (int p = 0; p < PB; ++p)                                                         \
                                for (int o = 0; o < ob; ++o)                                                     \
                                    acc[p][o] += ip[(size_t)p * cin + i] * wp[(size_t)i * cout + o];             \
                    }                                                                                            \
                }                                                                                                \
                for (int p = 0; p < px; ++p) {                                                                   \
                    const size_t pos = (size_t)((y + 1) * np + x0 + 1 + p);                                      \
                    float *op = out + pos * cout + o0;                                                           \
                    const float *rp = residual ? residual + pos * cout + o0 : NULL;                              \
                    for (int o = 0; o < ob; ++o) {                                                               \
                        float v = acc[p][o] * scale[o0 + o] + shift[o0 + o];                                     \
                        if (rp) v += rp[o];                                                                      \
                        op[o] = v > 0.0f ? v : 0.0f;                                                             \
                    }                                                                                            \
                }                                                                                                \
            }                                                                                                    \
    }                                                                                                            \
}
DEFINE_CONV(6)
DEFINE_CONV(7)
DEFINE_CONV(11)

/* a board row in as few, as full register blocks as the 32 vector registers allow: 11 = 11, 13 = 7 + 6, else sixes */
static void conv3x3_blocked(int n, int cin, int cout, const float *in, const float *w, const float *scale,
                            const float *shift, const float *residual, float *out) {
    if (n == 11) conv3x3_blocked_11(n, cin, cout, in, w, scale, shift, residual, out);
    else if (n == 13 || n == 7) conv3x3_blocked_7(n, cin, cout, in, w, scale, shift, residual, out);
    else conv3x3_blocked_6(n, cin, cout, in, w, scale, shift, residual, out);
}

void onet_forward_fast(const onet_t *cnet, int B, int K, const int32_t *boards, const int32_t *legal_moves,
                       float *value, float *logprob) {
    onet_t *net = (onet_t *)cnet;
    onet_finalize(net);
    const int n = net->n, n2 = n * n, np = n + 2, C = net->chans;
    const size_t img = (size_t)np * np + PBMAX + 2;
    float *x0 = (float *)calloc(img * 4, sizeof(float));
    float *a = (float *)calloc(img * C, sizeof(float));
    float *b = (float *)calloc(img * C, sizeof(float));
    float *c = (float *)calloc(img * C, sizeof(float));
    float *flat = (float *)malloc(sizeof(float) * n2 * C);
    float *vh = (float *)malloc(sizeof(float) * 2 * n2);
    float *ph = (float *)malloc(sizeof(float) * 4 * n2);
    float *logit = (float *)malloc(sizeof(float) * n2);
    for (int s = 0; s < B; ++s) {
        const int32_t *bd = boards + (size_t)s * n2;
        for (int y = 0; y < n; ++y)                        /* network.py:141-142; the halo stays zero */
            for (int x = 0; x < n; ++x)
                for (int ch = 0; ch < 4; ++ch)
                    x0[((size_t)(y + 1) * np + x + 1) * 4 + ch] = net->emb[bd[y * n + x] * 4 + ch];
        conv3x3_blocked(n, 4, C, x0, net->w_stem, net->s_stem, net->b_stem, NULL, a);
        for (int blk = 0; blk < net->blocks; ++blk) {      /* network.py:31-39 */
            conv3x3_blocked(n, C, C, a, net->w_blk[2 * blk], net->s_blk[2 * blk], net->b_blk[2 * blk], NULL, b);
            conv3x3_blocked(n, C, C, b, net->w_blk[2 * blk + 1], net->s_blk[2 * blk + 1], net->b_blk[2 * blk + 1], a, c);
            float *t = a; a = c; c = t;
        }
        for (int y = 0; y < n; ++y)
            memcpy(flat + (size_t)y * n * C, a + ((size_t)(y + 1) * np + 1) * C, sizeof(float) * n * C);
        onet_heads(net, flat, K, legal_moves + (size_t)s * K, value + s, logprob + (size_t)s * K, vh, ph, logit);
    }
    free(x0); free(a); free(b); free(c); free(flat); free(vh); free(ph); free(logit);
}

/* mcts.py:208-215 over the blocked forward: the evaluator obench_selfplay times */
void oeval_net_fast(void *ctx, int n, int B, int K, const int32_t *boards, const int32_t *legal_moves,
                    float *value, float *prior) {
    (void)n;
    onet_forward_fast((const onet_t *)ctx, B, K, boards, legal_moves, value, prior);
    for (size_t i = 0; i < (size_t)B * K; ++i) prior[i] = expf(prior[i]);
}
