/* oracle/net.c -- CPU restatement of azalea/network.py's inference forward (HexNetwork).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Plain fp32 direct convolution, eval-mode BatchNorm. */
#include "net_priv.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

onet_t *onet_new(int n, int blocks, int chans) {
    onet_t *net = (onet_t *)calloc(1, sizeof(onet_t));
    net->n = n; net->blocks = blocks; net->chans = chans;
    return net;
}

static void free_packed(onet_t *net) {
    if (!net->ready) return;
    free(net->emb); free(net->w_stem); free(net->s_stem); free(net->b_stem);
    for (int i = 0; i < 2 * net->blocks; ++i) { free(net->w_blk[i]); free(net->s_blk[i]); free(net->b_blk[i]); }
    free(net->w_blk); free(net->s_blk); free(net->b_blk);
    free(net->w_vc); free(net->s_vc); free(net->b_vc);
    free(net->w_pc); free(net->s_pc); free(net->b_pc);
    net->ready = 0;
}

void onet_free(onet_t *net) {
    if (!net) return;
    free_packed(net);
    for (int i = 0; i < net->nt; ++i) { free(net->names[i]); free(net->data[i]); }
    free(net);
}

int onet_set(onet_t *net, const char *name, const float *data, int64_t count) {
    free_packed(net);
    int slot = -1;
    for (int i = 0; i < net->nt; ++i) if (!strcmp(net->names[i], name)) slot = i;
    if (slot < 0) {
        if (net->nt >= ONET_MAX_TENSORS) return -1;
        slot = net->nt++;
        net->names[slot] = strdup(name);
        net->data[slot] = NULL;
    }
    free(net->data[slot]);
    net->data[slot] = (float *)malloc(sizeof(float) * count);
    memcpy(net->data[slot], data, sizeof(float) * count);
    net->count[slot] = count;
    return 0;
}

static const float *get(const onet_t *net, const char *name, int64_t want) {
    for (int i = 0; i < net->nt; ++i)
        if (!strcmp(net->names[i], name)) {
            if (want >= 0 && net->count[i] != want) {
                fprintf(stderr, "oracle net: tensor %s has %lld elements, want %lld\n", name,
                        (long long)net->count[i], (long long)want);
                abort();
            }
            return net->data[i];
        }
    fprintf(stderr, "oracle net: missing tensor %s\n", name);
    abort();
}

/* eval-mode BatchNorm2d as y = x*scale + shift (eps = 1e-5, torch default; network.py:21) */
static void fold_bn(const onet_t *net, const char *prefix, int c, float **scale, float **shift) {
    char nm[256];
    snprintf(nm, sizeof nm, "%s.weight", prefix);       const float *w = get(net, nm, c);
    snprintf(nm, sizeof nm, "%s.bias", prefix);         const float *b = get(net, nm, c);
    snprintf(nm, sizeof nm, "%s.running_mean", prefix); const float *m = get(net, nm, c);
    snprintf(nm, sizeof nm, "%s.running_var", prefix);  const float *v = get(net, nm, c);
    *scale = (float *)malloc(sizeof(float) * c);
    *shift = (float *)malloc(sizeof(float) * c);
    for (int i = 0; i < c; ++i) {
        double s = (double)w[i] / sqrt((double)v[i] + 1e-5);
        (*scale)[i] = (float)s;
        (*shift)[i] = (float)((double)b[i] - (double)m[i] * s);
    }
}

/* torch conv weight [cout][cin][kh][kw] -> [kh][kw][cin][cout] */
static float *pack_conv(const float *w, int cout, int cin, int ks) {
    float *p = (float *)malloc(sizeof(float) * cout * cin * ks * ks);
    for (int o = 0; o < cout; ++o)
        for (int i = 0; i < cin; ++i)
            for (int y = 0; y < ks; ++y)
                for (int x = 0; x < ks; ++x)
                    p[((y * ks + x) * cin + i) * cout + o] = w[((o * cin + i) * ks + y) * ks + x];
    return p;
}

void onet_finalize(onet_t *net) {
    if (net->ready) return;
    const int C = net->chans, n2 = net->n * net->n;
    char nm[256];
    net->emb = (float *)malloc(sizeof(float) * 12);
    memcpy(net->emb, get(net, "encoder.weight", 12), sizeof(float) * 12);
    net->w_stem = pack_conv(get(net, "conv1.weight", (int64_t)C * 4 * 9), C, 4, 3);
    fold_bn(net, "bn1", C, &net->s_stem, &net->b_stem);
    net->w_blk = (float **)calloc(2 * net->blocks, sizeof(float *));
    net->s_blk = (float **)calloc(2 * net->blocks, sizeof(float *));
    net->b_blk = (float **)calloc(2 * net->blocks, sizeof(float *));
    for (int b = 0; b < net->blocks; ++b)
        for (int h = 0; h < 2; ++h) {
            snprintf(nm, sizeof nm, "resblocks.%d.conv%d.weight", b, h + 1);
            net->w_blk[2 * b + h] = pack_conv(get(net, nm, (int64_t)C * C * 9), C, C, 3);
            snprintf(nm, sizeof nm, "resblocks.%d.bn%d", b, h + 1);
            fold_bn(net, nm, C, &net->s_blk[2 * b + h], &net->b_blk[2 * b + h]);
        }
    net->w_vc = pack_conv(get(net, "value_conv1.weight", (int64_t)2 * C), 2, C, 1);
    fold_bn(net, "value_bn1", 2, &net->s_vc, &net->b_vc);
    net->w_pc = pack_conv(get(net, "move_conv1.weight", (int64_t)4 * C), 4, C, 1);
    fold_bn(net, "move_bn1", 4, &net->s_pc, &net->b_pc);
    net->fc2_w = get(net, "value_fc2.weight", (int64_t)64 * 2 * n2);
    net->fc2_b = get(net, "value_fc2.bias", 64);
    net->fc3_w = get(net, "value_fc3.weight", 64);
    net->fc3_b = get(net, "value_fc3.bias", 1);
    net->mfc_w = get(net, "move_fc.weight", (int64_t)n2 * 4 * n2);
    net->mfc_b = get(net, "move_fc.bias", n2);
    net->ready = 1;
}

/* 3x3, pad 1, no bias; in/out are [pos][chan]; then BN scale/shift, optional residual, ReLU */
static void conv3x3_bn(int n, int cin, int cout, const float *in, const float *w,
                       const float *scale, const float *shift, const float *residual,
                       float *out) {
    float acc[512];
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) {
            for (int o = 0; o < cout; ++o) acc[o] = 0.0f;
            for (int ky = 0; ky < 3; ++ky) {
                int iy = y + ky - 1;
                if (iy < 0 || iy >= n) continue;
                for (int kx = 0; kx < 3; ++kx) {
                    int ix = x + kx - 1;
                    if (ix < 0 || ix >= n) continue;
                    const float *ip = in + (size_t)(iy * n + ix) * cin;
                    const float *wp = w + (size_t)((ky * 3 + kx) * cin) * cout;
                    for (int i = 0; i < cin; ++i) {
                        float a = ip[i];
                        const float *wr = wp + (size_t)i * cout;
                        for (int o = 0; o < cout; ++o) acc[o] += a * wr[o];
                    }
                }
            }
            float *op = out + (size_t)(y * n + x) * cout;
            const float *rp = residual ? residual + (size_t)(y * n + x) * cout : NULL;
            for (int o = 0; o < cout; ++o) {
                float v = acc[o] * scale[o] + shift[o];
                if (rp) v += rp[o];
                op[o] = v > 0.0f ? v : 0.0f;
            }
        }
}

/* value / policy heads + masked log_softmax of one position (network.py:77-84, :146-151) */
void onet_heads(const onet_t *net, const float *a, int K, const int32_t *lm, float *value, float *lp,
                float *vh, float *ph, float *logit) {
    const int n2 = net->n * net->n, C = net->chans;
    /* value head, network.py:77-81; flatten order (c, h, w) */
    for (int p = 0; p < n2; ++p)
        for (int o = 0; o < 2; ++o) {
            float acc = 0.0f;
            for (int i = 0; i < C; ++i) acc += a[(size_t)p * C + i] * net->w_vc[i * 2 + o];
            float v = acc * net->s_vc[o] + net->b_vc[o];
            vh[o * n2 + p] = v > 0.0f ? v : 0.0f;
        }
    float h2[64];
    for (int o = 0; o < 64; ++o) {
        float acc = 0.0f;
        const float *wr = net->fc2_w + (size_t)o * 2 * n2;
        for (int i = 0; i < 2 * n2; ++i) acc += vh[i] * wr[i];
        acc += net->fc2_b[o];
        h2[o] = acc > 0.0f ? acc : 0.0f;
    }
    float v3 = 0.0f;
    for (int i = 0; i < 64; ++i) v3 += h2[i] * net->fc3_w[i];
    v3 += net->fc3_b[0];
    *value = tanhf(v3);
    /* policy head, network.py:83-84, :146-151 */
    for (int p = 0; p < n2; ++p)
        for (int o = 0; o < 4; ++o) {
            float acc = 0.0f;
            for (int i = 0; i < C; ++i) acc += a[(size_t)p * C + i] * net->w_pc[i * 4 + o];
            float v = acc * net->s_pc[o] + net->b_pc[o];
            ph[o * n2 + p] = v > 0.0f ? v : 0.0f;
        }
    for (int t = 0; t < n2; ++t) {
        float acc = 0.0f;
        const float *wr = net->mfc_w + (size_t)t * 4 * n2;
        for (int i = 0; i < 4 * n2; ++i) acc += ph[i] * wr[i];
        logit[t] = acc + net->mfc_b[t];
    }
    float mx = -INFINITY;
    for (int j = 0; j < K; ++j) {
        int tile = lm[j] > 0 ? lm[j] - 1 : 0;          /* clamp(min=0), network.py:147 */
        lp[j] = lm[j] == 0 ? -99.0f : logit[tile];     /* network.py:150 */
        if (lp[j] > mx) mx = lp[j];
    }
    double sum = 0.0;
    for (int j = 0; j < K; ++j) sum += exp((double)lp[j] - (double)mx);
    float lse = (float)((double)mx + log(sum));
    for (int j = 0; j < K; ++j) lp[j] = lp[j] - lse;   /* network.py:151 */
}

/* HexNetwork.forward (network.py:134-152) over Network.forward (network.py:68-85) */
void onet_forward(const onet_t *cnet, int B, int K, const int32_t *boards,
                  const int32_t *legal_moves, float *value, float *logprob) {
    onet_t *net = (onet_t *)cnet;
    onet_finalize(net);
    const int n = net->n, n2 = n * n, C = net->chans;
    float *x0 = (float *)malloc(sizeof(float) * n2 * 4);
    float *a = (float *)malloc(sizeof(float) * n2 * C);
    float *b = (float *)malloc(sizeof(float) * n2 * C);
    float *c = (float *)malloc(sizeof(float) * n2 * C);
    float *vh = (float *)malloc(sizeof(float) * 2 * n2);
    float *ph = (float *)malloc(sizeof(float) * 4 * n2);
    float *logit = (float *)malloc(sizeof(float) * n2);
    for (int s = 0; s < B; ++s) {
        const int32_t *bd = boards + (size_t)s * n2;
        for (int p = 0; p < n2; ++p)                       /* network.py:141-142 */
            for (int ch = 0; ch < 4; ++ch) x0[p * 4 + ch] = net->emb[bd[p] * 4 + ch];
        conv3x3_bn(n, 4, C, x0, net->w_stem, net->s_stem, net->b_stem, NULL, a);  /* :73 */
        for (int blk = 0; blk < net->blocks; ++blk) {      /* network.py:31-39 */
            conv3x3_bn(n, C, C, a, net->w_blk[2 * blk], net->s_blk[2 * blk], net->b_blk[2 * blk], NULL, b);
            conv3x3_bn(n, C, C, b, net->w_blk[2 * blk + 1], net->s_blk[2 * blk + 1], net->b_blk[2 * blk + 1], a, c);
            float *t = a; a = c; c = t;
        }
        onet_heads(net, a, K, legal_moves + (size_t)s * K, value + s, logprob + (size_t)s * K, vh, ph, logit);
    }
    free(x0); free(a); free(b); free(c); free(vh); free(ph); free(logit);
}

/* mcts.py:208-215: value and prior = exp(moves_logprob) */
void oeval_net(void *ctx, int n, int B, int K, const int32_t *boards,
               const int32_t *legal_moves, float *value, float *prior) {
    (void)n;
    onet_forward((const onet_t *)ctx, B, K, boards, legal_moves, value, prior);
    for (size_t i = 0; i < (size_t)B * K; ++i) prior[i] = expf(prior[i]);
}
