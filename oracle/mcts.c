/* oracle/mcts.c -- CPU restatement of azalea/search_tree.py + azalea/mcts.py.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Compile with -ffp-contract=off: the PUCT score
 * (mcts.py:132-135) is evaluated op by op in float32 by numpy and tie-breaks depend on it. */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- tree: search_tree.py:43-71 -------------------------------------------------------- */
otree_t *otree_new(int64_t cap) {
    otree_t *t = (otree_t *)calloc(1, sizeof(otree_t));
    t->cap = cap;
    t->parent = (int32_t *)malloc(sizeof(int32_t) * cap);
    t->first_child = (int32_t *)malloc(sizeof(int32_t) * cap);
    t->num_children = (int32_t *)malloc(sizeof(int32_t) * cap);
    t->num_visits = (float *)malloc(sizeof(float) * cap);
    t->total_value = (float *)malloc(sizeof(float) * cap);
    t->prior_prob = (float *)malloc(sizeof(float) * cap);
    otree_reset(t);
    return t;
}

void otree_free(otree_t *t) {
    if (!t) return;
    free(t->parent); free(t->first_child); free(t->num_children);
    free(t->num_visits); free(t->total_value); free(t->prior_prob);
    free(t);
}

/* search_tree.py:59-71 */
void otree_reset(otree_t *t) {
    t->num_nodes = 1;
    t->root_id = 0;
    t->parent[0] = -1;
    t->first_child[0] = -1;
    t->num_children[0] = -1;
    t->num_visits[0] = 0.0f;
    t->total_value[0] = 0.0f;
    t->prior_prob[0] = 1.0f;
}

/* search_tree.py:115-132: re-root at an evaluated child, else forget everything */
void otree_move(otree_t *t, int move_id) {
    if (t->num_children[t->root_id] < 0) { otree_reset(t); return; }
    int32_t node = t->first_child[t->root_id] + move_id;
    if (t->num_children[node] < 0) otree_reset(t);
    else t->root_id = node;
}

/* search_tree.py:254-274 */
static int create_child_nodes(otree_t *t, int32_t id, int k, const float *prior) {
    if (t->num_nodes + k > t->cap) return -1;   /* SearchTreeFull */
    int32_t first = (int32_t)t->num_nodes;
    t->num_nodes += k;
    t->first_child[id] = first;
    t->num_children[id] = k;
    for (int j = 0; j < k; ++j) {
        int32_t c = first + j;
        t->parent[c] = id;
        t->first_child[c] = -1;
        t->num_children[c] = -1;
        t->num_visits[c] = 0.0f;
        t->total_value[c] = 0.0f;
        t->prior_prob[c] = prior[j];
    }
    return 0;
}

/* numpy's float32 pairwise add-reduce for n < 128 (np.sum at mcts.py:132 and :287) */
static float np_sum_f32(const float *a, int n) {
    if (n < 8) {
        float r = 0.0f;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    float r[8];
    int i;
    for (i = 0; i < 8; ++i) r[i] = a[i];
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

/* mcts.py:119-136 score_actions + np.argmax (mcts.py:112): returns the chosen child index */
static int score_and_pick(const otree_t *t, int32_t node, float c32, double noise_scale,
                          const double *noise) {
    int32_t fc = t->first_child[node];
    int k = t->num_children[node];
    const float *nv = t->num_visits + fc;
    const float *tv = t->total_value + fc;
    const float *pp = t->prior_prob + fc;
    float sum = (k < 128) ? np_sum_f32(nv, k) : 0.0f;
    if (k >= 128) { for (int j = 0; j < k; ++j) sum += nv[j]; } /* integers: order-free */
    float root = sqrtf(sum);
    float keep32 = (float)(1.0 - noise_scale);   /* python float -> weak scalar -> f32 */
    int best = 0;
    float best_score = 0.0f;
    for (int j = 0; j < k; ++j) {
        float P = pp[j];
        if (noise_scale != 0.0) {
            /* mcts.py:129-131: f32 product, promoted to f64 for the add, rounded back to f32 */
            float kept = keep32 * P;
            P = (float)((double)kept + noise_scale * noise[j]);
        }
        float gap = root / (1.0f + nv[j]);          /* mcts.py:132 */
        float U = (c32 * P) * gap;                  /* mcts.py:133 */
        float W = -tv[j];                           /* search_tree.py:203 */
        float Q = W / (nv[j] < 1.0f ? 1.0f : nv[j]); /* mcts.py:134 */
        float score = Q + U;                        /* mcts.py:135 */
        if (j == 0 || score > best_score) { best = j; best_score = score; }
    }
    return best;
}

/* mcts.py:79-92 (one leaf): leaf and every ancestor except the root */
static void virtual_loss(otree_t *t, int32_t node, float amount) {
    while (node != t->root_id) {
        t->num_visits[node] += amount;
        t->total_value[node] += amount;
        node = t->parent[node];
    }
}

/* mcts.py:258-293 sample_paths, with select_batch (:46-76), evaluate_batch's host half
 * (:168-200), expand_batch (:226-239) and backup_batch (:242-255) */
int osearch(otree_t *t, const ohex_t *game, oeval_fn eval, void *ctx, const osearch_cfg_t *cfg,
            osearch_stats_t *st) {
    const int n = game->n, cells = n * n, bs = cfg->batch_size;
    const float c32 = (float)cfg->c_puct;
    memset(st, 0, sizeof(*st));

    int32_t *leaf_node = (int32_t *)malloc(sizeof(int32_t) * bs);
    ohex_t *leaf_game = (ohex_t *)malloc(sizeof(ohex_t) * bs);
    int32_t *ubatch_board = (int32_t *)malloc(sizeof(int32_t) * bs * cells);
    int32_t *ubatch_moves = (int32_t *)malloc(sizeof(int32_t) * bs * cells);
    int32_t *tmp_moves = (int32_t *)malloc(sizeof(int32_t) * cells);
    float *value = (float *)malloc(sizeof(float) * bs);
    float *nt_value = (float *)malloc(sizeof(float) * bs);
    float *prior = (float *)malloc(sizeof(float) * bs * cells);
    int *nch = (int *)malloc(sizeof(int) * bs);
    int *nt_index = (int *)malloc(sizeof(int) * bs);
    int rc = 0;
    float search_value = 0.0f;   /* python int 0 += np.float32 -> float32 accumulator */

    /* mcts.py:272-273 + :18-27 evaluate_root (its value is discarded) */
    if (t->num_children[t->root_id] < 0) {
        int k = ohex_legal_moves(game, tmp_moves);
        if (ohex_result(game) != 0) {
            rc = create_child_nodes(t, t->root_id, 0, prior);
        } else {
            if (game->color == 2)
                ohex_flip_board_moves(n, game->board, tmp_moves, k, ubatch_board, ubatch_moves);
            else {
                memcpy(ubatch_board, game->board, sizeof(int32_t) * cells);
                memcpy(ubatch_moves, tmp_moves, sizeof(int32_t) * k);
            }
            eval(ctx, n, 1, k, ubatch_board, ubatch_moves, nt_value, prior);
            st->n_eval += 1;
            rc = create_child_nodes(t, t->root_id, k, prior);
        }
        if (rc) goto done;
    }

    int num_batches = cfg->simulations / bs + 1;   /* mcts.py:268 */
    for (int b = 0; b < num_batches; ++b) {
        /* ---- select_batch: mcts.py:62-72 ---- */
        for (int i = 0; i < bs; ++i) {
            ohex_t g = *game;                          /* snapshot/restore: search_tree.py:150-154 */
            int32_t node = t->root_id;
            double ns = cfg->noise_scale;
            const double *noise = NULL;
            if (ns != 0.0) {
                if (st->n_select >= cfg->noise_rows) { rc = -2; goto done; }
                noise = cfg->noise + st->n_select * (int64_t)t->num_children[t->root_id];
            }
            while (t->num_children[node] > 0) {        /* not leaf: mcts.py:105 */
                int child = score_and_pick(t, node, c32, ns, noise);
                st->sum_depth += 1;
                st->sum_k_interior += t->num_children[node];
                int k = ohex_legal_moves(&g, tmp_moves);
                (void)k;
                ohex_step(&g, tmp_moves[child]);        /* search_tree.py:306-308 */
                node = t->first_child[node] + child;
                ns = 0.0;                               /* mcts.py:114 */
            }
            st->n_select += 1;
            virtual_loss(t, node, 1.0f);               /* mcts.py:68 */
            leaf_node[i] = node;
            leaf_game[i] = g;
        }
        for (int i = 0; i < bs; ++i) virtual_loss(t, leaf_node[i], -1.0f);   /* mcts.py:72 */

        /* ---- deduplicate_leaves: mcts.py:139-152 (keep first occurrence) ---- */
        int nu = 0;
        for (int i = 0; i < bs; ++i) {
            int dup = 0;
            for (int j = 0; j < nu; ++j) if (leaf_node[j] == leaf_node[i]) { dup = 1; break; }
            if (!dup) { leaf_node[nu] = leaf_node[i]; leaf_game[nu] = leaf_game[i]; ++nu; }
        }

        /* ---- evaluate_batch: mcts.py:155-217 ---- */
        int K = 0, nnt = 0;
        for (int i = 0; i < nu; ++i) {
            nch[i] = ohex_legal_moves(&leaf_game[i], tmp_moves);
            if (nch[i] > K) K = nch[i];
        }
        for (int i = 0; i < nu; ++i) {
            if (ohex_result(&leaf_game[i]) != 0) {
                value[i] = -1.0f;                      /* mcts.py:194-195 */
                st->n_terminal_evals += 1;
                continue;
            }
            int32_t *bd = ubatch_board + (size_t)nnt * cells;
            int32_t *mv = ubatch_moves + (size_t)nnt * K;
            int k = ohex_legal_moves(&leaf_game[i], tmp_moves);
            memset(mv, 0, sizeof(int32_t) * K);
            if (leaf_game[i].color == 2) {             /* state.color == 1: mcts.py:178-181 */
                ohex_flip_board_moves(n, leaf_game[i].board, tmp_moves, k, bd, mv);
            } else {
                memcpy(bd, leaf_game[i].board, sizeof(int32_t) * cells);
                memcpy(mv, tmp_moves, sizeof(int32_t) * k);
            }
            nt_index[nnt++] = i;
        }
        if (nnt) {
            eval(ctx, n, nnt, K, ubatch_board, ubatch_moves, nt_value, prior);
            st->n_eval += nnt;
            for (int j = 0; j < nnt; ++j) value[nt_index[j]] = nt_value[j];
        }

        /* ---- expand_batch: mcts.py:226-239 ---- */
        for (int i = 0, j = 0; i < nu; ++i) {
            int is_nt = (j < nnt && nt_index[j] == i);
            if (t->num_children[leaf_node[i]] != 0) {   /* not already terminal */
                const float *p = is_nt ? prior + (size_t)j * K : prior;
                rc = create_child_nodes(t, leaf_node[i], nch[i], p);
                if (rc) goto done;
                st->sum_k_leaf += nch[i];
            }
            if (is_nt) ++j;
        }

        /* ---- backup_batch: mcts.py:242-255 ---- */
        for (int i = 0; i < nu; ++i) {
            int32_t node = leaf_node[i];
            float v = value[i];
            for (;;) {
                t->total_value[node] += v;
                t->num_visits[node] += 1.0f;
                v = -v;
                if (node == t->root_id) break;
                node = t->parent[node];
            }
        }
        search_value += np_sum_f32(value, nu);          /* mcts.py:287 */
    }
    st->search_value = (double)(search_value / (float)(num_batches * bs));   /* mcts.py:291 */

done:
    st->status = rc;
    free(leaf_node); free(leaf_game); free(ubatch_board); free(ubatch_moves); free(tmp_moves);
    free(value); free(nt_value); free(prior); free(nch); free(nt_index);
    return rc;
}

/* ---- built-in evaluators ------------------------------------------------------------- */
void oeval_uniform(void *vctx, int n, int B, int K, const int32_t *boards,
                   const int32_t *legal_moves, float *value, float *prior) {
    const ouniform_ctx_t *c = (const ouniform_ctx_t *)vctx;
    for (int i = 0; i < B; ++i) {
        int k = 0;
        for (int j = 0; j < K; ++j) if (legal_moves[(size_t)i * K + j] > 0) ++k;
        float p = c->prior_by_k ? c->prior_by_k[k] : 1.0f / (float)k;
        for (int j = 0; j < K; ++j) prior[(size_t)i * K + j] = j < k ? p : 0.0f;
        if (c->hash_value) {
            uint32_t h = ofnv1a(boards + (size_t)i * n * n, n * n);
            value[i] = (float)((double)(h & 0xffff) / 32768.0 - 1.0);
        } else {
            value[i] = 0.0f;
        }
    }
}

void oeval_tape(void *vctx, int n, int B, int K, const int32_t *boards,
                const int32_t *legal_moves, float *value, float *prior) {
    otape_ctx_t *c = (otape_ctx_t *)vctx;
    (void)n; (void)boards;
    for (int i = 0; i < B; ++i) {
        /* the tape also holds terminal rows (value -1, k 0): skip them, they never reach here */
        while (c->pos < c->len && c->nch[c->pos] == 0) c->pos++;
        if (c->pos >= c->len) { c->mismatch = 1; value[i] = 0; continue; }
        int k = 0;
        for (int j = 0; j < K; ++j) if (legal_moves[(size_t)i * K + j] > 0) ++k;
        if (k != c->nch[c->pos]) c->mismatch = 1;
        value[i] = c->value[c->pos];
        const float *p = c->prior + c->off[c->pos];
        for (int j = 0; j < K; ++j) prior[(size_t)i * K + j] = j < k ? p[j] : 0.0f;
        c->pos++;
    }
}
