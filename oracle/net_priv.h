/* oracle/net_priv.h -- the network's private layout, shared by net.c (parity path) and net_fast.c (timed path).
 * TEST INFRASTRUCTURE ONLY (see oracle.h). */
#ifndef ORACLE_NET_PRIV_H
#define ORACLE_NET_PRIV_H
#include "oracle.h"

#define ONET_MAX_TENSORS 512

struct onet {
    int n, blocks, chans;
    int nt;
    char *names[ONET_MAX_TENSORS];
    float *data[ONET_MAX_TENSORS];
    int64_t count[ONET_MAX_TENSORS];
    /* packed, built lazily by finalize() */
    int ready;
    float *emb;         /* [3][4]                  network.py:125 */
    float *w_stem;      /* [3][3][4][C]            network.py:47 */
    float *s_stem, *b_stem;           /* folded BN scale/shift, network.py:48 */
    float **w_blk;      /* 2*blocks x [3][3][C][C] network.py:20-23 */
    float **s_blk, **b_blk;
    float *w_vc, *s_vc, *b_vc;        /* [C][2]   network.py:54-55 */
    float *w_pc, *s_pc, *b_pc;        /* [C][4]   network.py:59-60 */
    const float *fc2_w, *fc2_b, *fc3_w, *fc3_b, *mfc_w, *mfc_b;   /* network.py:56-57, :127 */
};

void onet_finalize(onet_t *net);   /* pack the weights (idempotent) */
/* value / policy heads + masked log_softmax of one position from its final activations a[pos][chan]
 * (network.py:77-84, :146-151); scratch: vh[2 n^2], ph[4 n^2], logit[n^2] */
void onet_heads(const onet_t *net, const float *a, int K, const int32_t *lm, float *value, float *lp,
                float *vh, float *ph, float *logit);
#endif
