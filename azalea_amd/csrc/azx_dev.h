// azx_dev.h -- device-side data model shared by the tree and network kernels (gfx950 only).
//
// HBM layout, per game slot g (see DESIGN.md "Data layout"):
//   cells[g][SLOTS*64]  u32 per board cell: bits0-1 colour (0 empty, 1 X, 2 O),
//                       bits2-3 edge flags of the cell's group (low edge / high edge),
//                       bits8..  group label (cell index of the stone that last merged it).
//                       One wavefront owns one game; lane l holds cells l, l+64, l+128.
//   arena[a][g][cap]    Node (16 B): {num_visits, total_value, prior_prob, link}.
//                       link >= 0: first child id (children are contiguous, in ascending
//                       tile order); -1: unevaluated; <= -2: terminal (first_child value
//                       the reference would hold = -2 - link).  The reference's parent[] and
//                       num_children[] (search_tree.py:48-50) are implied: a Hex node with k
//                       legal moves has children with k-1, and paths are recorded on descent.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/azx.h"

#define AZX_LINK_UNEVAL (-1)
#define AZX_LINK_TERM(fc) (-2 - (fc))
#define AZX_LABEL_NONE 0x3FFFFFu

struct alignas(16) Node {
    float nv;      // num_visits   (float32 on purpose: search_tree.py:53)
    float tv;      // total_value  (own perspective, search_tree.py:37-38)
    float pp;      // prior_prob   (parent's perspective, search_tree.py:39-40)
    int32_t link;
};

struct alignas(64) TreeHdr {
    int32_t num_nodes;
    int32_t root_id;
    int32_t root_k;        // legal moves at the root
    int32_t k0;            // legal moves at node 0 (for the six-array dump)
    int32_t arena;         // which ping-pong arena holds the live tree
    int32_t status;        // 0 ok, 1 SearchTreeFull
    int32_t batches_left;  // select batches still to run in the current search
    int32_t pending;       // leaves selected and waiting for expand/backup
    int32_t pending_root;  // the pending leaf is the root evaluation (no backup)
    int32_t select_count;  // select_leaf calls so far in this search (noise row index)
    float   search_value;  // mcts.py:287 accumulator (float32)
    int32_t slow_div;      // a backed-up value was outside the range the unscaled divide is exact for
    int32_t defer_compact; // play mode: compact the arena (keep the subtree under root_id) before the next search
    int32_t dropped;       // nodes discarded by arena compactions since the last reset: num_nodes + dropped is the
                           // reference's never-reclaimed node count (search_tree.py:112 'search_tree_nodes')
    int32_t pad[2];
};

struct alignas(64) GameHdr {
    int32_t color;     // side to move, 1 = X, 2 = O
    int32_t winner;    // 0 / 1 / 2
    int32_t ply;
    int32_t active;    // slot takes part in search/advance
    int64_t uid;       // global game index (seeds the device RNG stream)
    int32_t move_id;   // device-chosen child index of the last azx_play step
    int32_t n_rows;    // replay rows this game has written so far
    int32_t gen;       // games this slot has started so far: uid = slot + n_games * gen (a fixed
                       // seed reproduces every game whatever order the GPU schedules the slots in)
    int32_t ply0;      // ply the game started from (azx_reset with a move prefix; 0 for restarts): its replay
                       // row r was recorded at ply ply0 + r
    int32_t parked;    // play mode: the game is finished but the harvest queue had no room for its rows; the slot
                       // waits (active = 0) until the host has drained the queue (k_advance in unpark mode)
    int32_t pad[5];
};

enum {   // counters[] slots
    CTR_SELECTS = 0, CTR_SUM_DEPTH, CTR_SUM_K_INT, CTR_SUM_K_LEAF, CTR_EVALS, CTR_TERM_EVALS,
    CTR_GAMES, CTR_ERRORS, CTR_PLIES, CTR_ROWS, CTR_COUNT = 16
};

struct DevEngine {
    int32_t N, ncells, G, bs, cap, slots;
    int32_t selects_per_search;   // (simulations / bs + 1) * bs, mcts.py:268
    float c_puct;
    int32_t evaluator, flags;
    // games
    uint32_t *cells;        // [G][slots*64]
    GameHdr *ghdr;          // [G]
    // trees
    Node *arena[2];         // [G][cap] each
    TreeHdr *thdr;          // [G]
    // per-leaf scratch of the batch in flight
    int32_t *leaf_node;     // [G][bs]
    int32_t *leaf_len;      // [G][bs] path length (nodes below the root)
    int32_t *leaf_eval;     // [G][bs] index into ev_* or -1 (terminal)
    int32_t *leaf_link;     // [G][bs] the leaf's link when it was selected (-1 or terminal code)
    int32_t *leaf_cells;    // [G][bs] first cell of the path | last cell << 16
    uint64_t *leaf_mask;    // [G][bs][4] empties bitmask at the leaf (original frame)
    int32_t *path;          // [G][bs][ncells]
    // evaluation requests / results (packed by atomic counter)
    uint8_t *ev_board;      // [E][AZX_CELL_STRIDE] network input (first player's view)
    int32_t *ev_src;        // [E] g*bs + i
    int32_t *ev_flip;       // [E] 1 if the board was flipped (mover is O)
    float *ev_value;        // [E]
    float *ev_prior;        // [E][AZX_CELL_STRIDE] by ORIGINAL cell index
    int32_t *n_eval;        // [1]
    // noise (parity: host rows; throughput: device RNG)
    const double *noise;    // [G][n_select][noise_stride] or null
    int32_t n_select, noise_stride;
    double noise_scale;
    float noise_alpha;
    const float *gamma_tab;    // [AZX_GAMMA_TAB_FLOATS] device gamma sampler table for noise_alpha (mcts_kernels.hip)
    int32_t device_noise;
    uint64_t seed;
    int32_t uid_stride, uid_offset;   // uid = (slot + G * gen) * uid_stride + uid_offset (azx_config.game_index_*)
    const float *prior_by_k;   // [ncells+1]
    int32_t prior_default;     // 1: prior_by_k is the default float32 1/k table
    unsigned long long *counters;   // [G][CTR_COUNT] per-game (no atomics); the host sums over games
    // throughput mode (azx_play): per-slot replay rows of the game in progress ...
    int32_t exploration_depth;
    float temperature;
    uint8_t *row_board;     // [G][ncells][AZX_CELL_STRIDE] absolute colours before the move
    float *row_prob;        // [G][ncells][AZX_CELL_STRIDE] moves_prob dense by child index
    int32_t *row_k;         // [G][ncells]
    float *row_meta;        // [G][ncells][8] per-ply search metrics of the game in progress: search_value, root
                            // width, log-probability of the move drawn, (first-row flag), mean root-child visits,
                            // tree nodes (search_tree.py:109-112; play_game.py:41-43 averages them per game)
    // ... and the output queue finished games are appended to (whole games only)
    int64_t q_cap;
    int32_t q_ring;         // 1: wrap around instead of stalling (bench)
    uint8_t *q_board;       // [Q][AZX_CELL_STRIDE]
    float *q_prob;          // [Q][AZX_CELL_STRIDE]
    int32_t *q_color, *q_k; // [Q]
    float *q_reward;        // [Q]
    int64_t *q_uid;         // [Q]
    float *q_meta;          // [Q][8] the rows' per-ply search metrics (AZX_ROW_METRICS floats per row)
    unsigned long long *q_count;   // [1] rows appended
    double *stat_sums;      // [G][8] per game: search_value, root_width, action_logprob, reward_last
};

// ---- wave64 reductions on DPP (no LDS crossbar round trips) ---------------------------------
// row_shr 1/2/4/8 build an inclusive scan inside each 16-lane row, row_bcast15 / row_bcast31
// fold the rows; the total lands in lane 63 and is broadcast with v_readlane.
#define AZX_DPP(old, v, ctrl, rmask) \
    __builtin_amdgcn_update_dpp((old), (v), (ctrl), (rmask), 0xf, false)

__device__ __forceinline__ float wave_sum(float v) {
    v += __int_as_float(AZX_DPP(0, __float_as_int(v), 0x111, 0xf));
    v += __int_as_float(AZX_DPP(0, __float_as_int(v), 0x112, 0xf));
    v += __int_as_float(AZX_DPP(0, __float_as_int(v), 0x114, 0xf));
    v += __int_as_float(AZX_DPP(0, __float_as_int(v), 0x118, 0xf));
    v += __int_as_float(AZX_DPP(0, __float_as_int(v), 0x142, 0xa));
    v += __int_as_float(AZX_DPP(0, __float_as_int(v), 0x143, 0xc));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_sum_i(int v) {
    v += AZX_DPP(0, v, 0x111, 0xf);
    v += AZX_DPP(0, v, 0x112, 0xf);
    v += AZX_DPP(0, v, 0x114, 0xf);
    v += AZX_DPP(0, v, 0x118, 0xf);
    v += AZX_DPP(0, v, 0x142, 0xa);
    v += AZX_DPP(0, v, 0x143, 0xc);
    return __builtin_amdgcn_readlane(v, 63);
}
// unsigned max: 0 is the identity, so bound_ctrl DPP reads need no 'old' operand
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#define AZX_DPPZ(v, ctrl, rmask) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (rmask), 0xf, true)
    uint32_t t;
    t = AZX_DPPZ(v, 0x111, 0xf); v = t > v ? t : v;
    t = AZX_DPPZ(v, 0x112, 0xf); v = t > v ? t : v;
    t = AZX_DPPZ(v, 0x114, 0xf); v = t > v ? t : v;
    t = AZX_DPPZ(v, 0x118, 0xf); v = t > v ? t : v;
    t = AZX_DPPZ(v, 0x142, 0xa); v = t > v ? t : v;
    t = AZX_DPPZ(v, 0x143, 0xc); v = t > v ? t : v;
#undef AZX_DPPZ
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ float wave_max(float v) {
    const int ninf = 0xff800000;   // -inf: identity of max
    v = fmaxf(v, __int_as_float(AZX_DPP(ninf, __float_as_int(v), 0x111, 0xf)));
    v = fmaxf(v, __int_as_float(AZX_DPP(ninf, __float_as_int(v), 0x112, 0xf)));
    v = fmaxf(v, __int_as_float(AZX_DPP(ninf, __float_as_int(v), 0x114, 0xf)));
    v = fmaxf(v, __int_as_float(AZX_DPP(ninf, __float_as_int(v), 0x118, 0xf)));
    v = fmaxf(v, __int_as_float(AZX_DPP(ninf, __float_as_int(v), 0x142, 0xa)));
    v = fmaxf(v, __int_as_float(AZX_DPP(ninf, __float_as_int(v), 0x143, 0xc)));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// ---- Hex rules on one wavefront: azalea/game/hex.py:137-231 ------------------------------
// The reference flood-fills the last mover's group (hex.py:204-231); here every cell carries
// its group's label and edge flags, so placing a stone is one O(1) wave-parallel relabel
// (a union of <= 6 neighbouring groups) and the win test is "merged flags == both edges".
// Board geometry in constant memory (scalar loads): for every board size N = 2..13 and every cell
// four u64: the neighbour bitmask (hex.py:190-195 neighbourhood) over cell slots 0..2, and the
// cell's own edge bits (bit0/1: row 0 / row N-1 for colour 1, bit2/3: column 0 / N-1 for colour 2).
#define AZX_GEO_CELLS 820            // sum of N^2 for N = 2..13 is 818
extern __constant__ uint64_t c_geo[AZX_GEO_CELLS * 4];
// float32 sqrt of the integers 0..AZX_SQRT_TAB-1 (correctly rounded, filled by the host): the
// sum of child visit counts under the square root of mcts.py:132 is always a small integer
#define AZX_SQRT_TAB 4096
extern __constant__ float c_sqrt[AZX_SQRT_TAB];
__host__ __device__ inline int azx_geo_base(int N) {   // cells of all smaller boards
    int b = 0;
    for (int n = 2; n < N; ++n) b += n * n;
    return b;
}

template <int SLOTS>
struct HexWave {
    uint32_t c[SLOTS];        // per-lane cells: colour | group flags << 2 | group label << 8
    uint64_t occ[2][SLOTS];   // wave-uniform bitboards of the X / O stones
    int color;                // 1 = X to move, 2 = O (hex.py:148)
    int winner;               // hex.py:149
    int gbase;                // this board size's offset into c_geo

    __device__ __forceinline__ void geom(int N, int lane) { (void)lane; gbase = azx_geo_base(N); }
    __device__ __forceinline__ void load(const uint32_t *p, int lane) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            c[s] = p[s * 64 + lane];
            occ[0][s] = __ballot((c[s] & 3u) == 1u);
            occ[1][s] = __ballot((c[s] & 3u) == 2u);
        }
    }
    __device__ __forceinline__ void store(uint32_t *p, int lane) const {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) p[s * 64 + lane] = c[s];
    }
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) { c[s] = 0; occ[0][s] = 0ull; occ[1][s] = 0ull; }
        color = 1;
        winner = 0;
    }
    // empties bitmask, slot s (hex.py:151-159: legal moves are the empty cells while winner==0):
    // pure scalar arithmetic on the stone bitboards
    __device__ __forceinline__ uint64_t empties(int s, int lane, int ncells) const {
        (void)lane;
        const int nv = ncells - 64 * s;
        const uint64_t valid = nv >= 64 ? ~0ull : (nv <= 0 ? 0ull : ((1ull << nv) - 1ull));
        return ~(occ[0][s] | occ[1][s]) & valid;
    }
    // hex.py:172-179 step + :204-231 check_win.  `cell` must be wave-uniform, empty, winner==0.
    // The same-colour neighbours are the AND of the cell's neighbour mask (constant memory) with
    // the mover's bitboard -- scalar work; their groups' edge flags are OR-ed and their groups
    // relabelled with one vector compare per neighbouring stone.
    __device__ __forceinline__ void step(int cell, int N, int lane) {
        (void)N;
        const int col = color;
        const uint64_t *g = c_geo + (size_t)(gbase + cell) * 4;
        uint32_t flags = (uint32_t)(g[3] >> (2 * (col - 1))) & 3u;
        uint64_t nb[SLOTS];
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            nb[s] = g[s] & (col == 1 ? occ[0][s] : occ[1][s]);
            uint64_t m = nb[s];
            while (m) {
                const int j = (int)__ffsll((long long)m) - 1;
                m &= m - 1;
                flags |= ((uint32_t)__builtin_amdgcn_readlane((int)c[s], j) >> 2) & 3u;
            }
        }
        const uint32_t nv = (uint32_t)col | (flags << 2) | ((uint32_t)cell << 8);
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            uint64_t m = nb[s];
            while (m) {
                const int j = (int)__ffsll((long long)m) - 1;
                m &= m - 1;
                const uint32_t lab = (uint32_t)__builtin_amdgcn_readlane((int)c[s], j) >> 8;
#pragma unroll
                for (int t = 0; t < SLOTS; ++t)
                    if ((int)(c[t] & 3u) == col && (c[t] >> 8) == lab) c[t] = nv;
            }
        }
        const uint64_t bit = 1ull << (cell & 63);
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            if (s * 64 + lane == cell) c[s] = nv;
            if (s == (cell >> 6)) {
                if (col == 1) occ[0][s] |= bit; else occ[1][s] |= bit;
            }
        }
        winner = (flags == 3u) ? col : 0;
        color = 3 - col;
    }
};

