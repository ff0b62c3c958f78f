/* oracle/selfplay.c -- whole-game self-play loop over the CPU restatement, one game per thread.
 * TEST INFRASTRUCTURE ONLY (see oracle.h): this is the timed host-core baseline of bench.py
 * ("cpu_baseline", kind "port"; the network forward is net_fast.c's blocked one).  The loop restates azalea/play_game.py:44-67 around
 * azalea/policy.py:132-176; the RNG is a local xoshiro256** (the reference's numpy RandomState
 * is only reproduced in parity tests, where the host draws the noise and the move). */
#include "oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct { uint64_t s[4]; } rng_t;

static uint64_t splitmix(uint64_t *x) {
    uint64_t z = (*x += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
static void rng_seed(rng_t *r, uint64_t seed) { for (int i = 0; i < 4; ++i) r->s[i] = splitmix(&seed); }
static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static uint64_t rng_next(rng_t *r) {
    uint64_t *s = r->s, res = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return res;
}
static double rng_uniform(rng_t *r) { return ((rng_next(r) >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
static double rng_normal(rng_t *r) {
    double u = rng_uniform(r), v = rng_uniform(r);
    return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v);
}
/* Marsaglia-Tsang, with the alpha<1 boost */
static double rng_gamma(rng_t *r, double alpha) {
    if (alpha < 1.0) return rng_gamma(r, alpha + 1.0) * pow(rng_uniform(r), 1.0 / alpha);
    double d = alpha - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (;;) {
        double x = rng_normal(r), v = 1.0 + c * x;
        if (v <= 0) continue;
        v = v * v * v;
        double u = rng_uniform(r);
        if (log(u) < 0.5 * x * x + d - d * v + d * log(v)) return d * v;
    }
}

typedef struct {
    const obench_cfg_t *cfg;
    const onet_t *net;
    int tid;
    int64_t games, plies, selects, evals;
} worker_t;

static int next_game;
static pthread_mutex_t lock = PTHREAD_MUTEX_INITIALIZER;

static void *worker(void *arg) {
    worker_t *w = (worker_t *)arg;
    const obench_cfg_t *cfg = w->cfg;
    const int n = cfg->n, cells = n * n;
    int sel_per_move = (cfg->simulations / cfg->batch_size + 1) * cfg->batch_size;
    /* the reference never frees nodes: <= (sims+1) expansions per ply, k shrinking by one per ply */
    otree_t *tree = otree_new((int64_t)(sel_per_move + 1) * cells * (cells + 1) / 2 + 1024);
    double *noise = (double *)malloc(sizeof(double) * sel_per_move * cells);
    int32_t lm[OHEX_MAXC];
    double probs[OHEX_MAXC];
    ouniform_ctx_t uctx = {0, NULL};
    for (;;) {
        pthread_mutex_lock(&lock);
        int gi = next_game < cfg->n_games ? next_game++ : -1;
        pthread_mutex_unlock(&lock);
        if (gi < 0) break;
        rng_t rng;
        rng_seed(&rng, cfg->seed + (uint64_t)gi);
        ohex_t game;
        ohex_init(&game, n);
        otree_reset(tree);
        int ply = 0;
        if (cfg->start_max > 0) {          /* random legal prefix: mid-game start (see obench_cfg_t) */
            int want = (int)(rng_next(&rng) % (uint64_t)(cfg->start_max + 1));
            while (ply < want) {
                int k = ohex_legal_moves(&game, lm);
                ohex_t trial = game;
                ohex_step(&trial, lm[rng_next(&rng) % (uint64_t)k]);
                if (ohex_result(&trial)) break;
                game = trial;
                ++ply;
            }
        }
        const int ply0 = ply;
        for (; ply - ply0 < cfg->max_plies && !ohex_result(&game); ++ply) {
            int k = ohex_legal_moves(&game, lm);
            if (cfg->noise_scale != 0.0) {            /* mcts.py:128, one draw per select_leaf */
                for (int s = 0; s < sel_per_move; ++s) {
                    double sum = 0.0, *row = noise + (size_t)s * k;
                    for (int j = 0; j < k; ++j) { row[j] = rng_gamma(&rng, cfg->noise_alpha); sum += row[j]; }
                    for (int j = 0; j < k; ++j) row[j] = sum > 0 ? row[j] / sum : 1.0 / k;
                }
            }
            osearch_cfg_t sc = {cfg->simulations, cfg->batch_size, cfg->c_puct, cfg->noise_scale,
                                noise, sel_per_move};
            osearch_stats_t st;
            int rc = cfg->use_net ? osearch(tree, &game, oeval_net_fast, (void *)w->net, &sc, &st)
                                  : osearch(tree, &game, oeval_uniform, &uctx, &sc, &st);
            if (rc) break;                             /* SearchTreeFull: game skipped */
            w->selects += st.n_select;
            w->evals += st.n_eval;
            /* search_tree.py:327-344 as_distribution + policy.py:142-160 */
            const float *nv = tree->num_visits + tree->first_child[tree->root_id];
            double T = ply >= cfg->exploration_depth ? 0.0 : cfg->temperature;
            double mx = 0, z = 0;
            for (int j = 0; j < k; ++j) if (nv[j] > mx) mx = nv[j];
            for (int j = 0; j < k; ++j) {
                probs[j] = T > 0 ? (nv[j] > 0 ? pow(nv[j], 1.0 / T) : 0.0) : (nv[j] == mx ? 1.0 : 0.0);
                z += probs[j];
            }
            double u = rng_uniform(&rng) * z, acc = 0;
            int move_id = k - 1;
            for (int j = 0; j < k; ++j) { acc += probs[j]; if (u < acc) { move_id = j; break; } }
            otree_move(tree, move_id);                 /* policy.py:170-176 */
            ohex_step(&game, lm[move_id]);
        }
        w->plies += ply - ply0;
        w->games += 1;
    }
    otree_free(tree);
    free(noise);
    return NULL;
}

int obench_selfplay(const obench_cfg_t *cfg, const onet_t *net, obench_out_t *out) {
    int nt = cfg->n_threads > 0 ? cfg->n_threads : 1;
    if (nt > 256) nt = 256;
    pthread_t th[256];
    worker_t ws[256];
    if (cfg->use_net && net) {   /* pack weights before the threads race to do it */
        int32_t bd[OHEX_MAXC] = {0}, mv[1] = {1};
        float v, lp;
        onet_forward(net, 1, 1, bd, mv, &v, &lp);
    }
    next_game = 0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < nt; ++i) {
        memset(&ws[i], 0, sizeof(ws[i]));
        ws[i].cfg = cfg; ws[i].net = net; ws[i].tid = i;
        pthread_create(&th[i], NULL, worker, &ws[i]);
    }
    memset(out, 0, sizeof(*out));
    for (int i = 0; i < nt; ++i) {
        pthread_join(th[i], NULL);
        out->games += ws[i].games; out->plies += ws[i].plies;
        out->selects += ws[i].selects; out->evals += ws[i].evals;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    out->seconds = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    return 0;
}
