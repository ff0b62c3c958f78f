/*
 * azx.h -- C ABI of the MI355X-native batched Hex self-play engine (libazx_hip.so).
 *
 * This is the drop-in boundary for azalea's MCTS self-play hot path.  The reference
 * (jseppanen/azalea) has no FFI layer: its seam is the duck-typed Python surface
 * Player.read / Policy.choose_action / SearchTree.search.  Each entry point below names
 * the reference interface (file:line under azalea/) it replaces; azalea_amd/ binds them
 * with ctypes and INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions: plain pointers and sizes only (no torch types).  Every function returns
 * 0 on success or a negative AZX_E* code; azx_last_error() gives the message.  Host
 * buffers are caller-allocated.  An engine handle is NOT thread-safe: one host thread
 * per handle, all device work stream-ordered on the engine's own HIP stream, calls block.
 *
 * Units follow the reference: a "move" is flat tile index + 1 (0 = padding,
 * game/hex.py:151-159); a "move_id"/child index i is the i-th legal move in ascending
 * tile order (search_tree.py:298-308); colour 1 = X/first player, 2 = O.
 */
#ifndef AZX_H
#define AZX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZX_MAX_BOARD 13          /* 11x11 and 13x13 are the BASELINE configs */
#define AZX_MAX_CELLS (AZX_MAX_BOARD * AZX_MAX_BOARD)
#define AZX_CELL_STRIDE 192       /* per-position row stride of dense [*, cells] buffers */
#define AZX_MAX_BATCH 16          /* search_batch_size upper bound */
#define AZX_ROW_METRICS 8         /* floats per replay row returned by azx_play_row_metrics */
/* bytes of one fixed-size replay record (azx_rows_pack / azx_replay_put_records) for a board of `cells` cells */
#define AZX_RECORD_BYTES(cells) ((size_t)((16 + 5 * (cells) + 15) / 16 * 16))

enum {
    AZX_OK = 0,
    AZX_EINVAL = -1,     /* bad argument / configuration */
    AZX_EHIP = -2,       /* HIP runtime failure (message has the hipError string) */
    AZX_ENOMEM = -3,
    AZX_ESTATE = -4,     /* call sequence error (e.g. apply without select) */
    AZX_ENODEV = -5,     /* no MI355X visible: there is no CPU fallback */
    AZX_ERANGE = -6      /* a folded weight or a tower activation is outside the f16 range of the split-f16 kernels
                            (the reference computes in fp32 throughout, network.py:68-85): azx_set_weights rejects such
                            weights; a call whose evaluations overflowed reports it and its results are not valid */
};

/* evaluator = what plays the role of Network.run inside mcts.evaluate_batch (mcts.py:202-215) */
enum {
    AZX_EVAL_RESNET = 0,    /* HexNetwork forward on device (network.py:134-152) */
    AZX_EVAL_UNIFORM = 1,   /* uniform priors 1/k, value 0, evaluated inline (BASELINE config 2) */
    AZX_EVAL_UNIFORM_HASH = 2, /* parity stub: prior_by_k table + fnv1a value hash, inline */
    AZX_EVAL_EXTERNAL = 3   /* host supplies value/priors per leaf between select and apply */
};

enum {
    AZX_FLAG_NO_COMPACT = 1  /* keep the reference's never-free arena + moving root_id
                                (search_tree.py:115-132) instead of compacting on advance */
};

typedef struct {
    int32_t board_size;          /* N: policy.py:51 */
    int32_t n_games;             /* concurrent game slots on this GPU */
    int32_t simulations;         /* policy.py:55 */
    int32_t search_batch_size;   /* policy.py:56 (<= AZX_MAX_BATCH) */
    float   exploration_coef;    /* c_puct, policy.py:57 (f32: numpy weak-scalar semantics) */
    int32_t exploration_depth;   /* policy.py:58 */
    double  noise_alpha;         /* policy.py:59 */
    double  noise_scale;         /* policy.py:60 */
    double  temperature;         /* policy.py:61 */
    int32_t evaluator;           /* AZX_EVAL_* */
    int32_t num_blocks;          /* policy.py:52 (resnet only) */
    int32_t base_chans;          /* policy.py:53 (resnet only; 64 or multiples of 32) */
    int32_t nodes_per_game;      /* tree arena capacity per game (SearchTreeFull beyond it) */
    int32_t flags;               /* AZX_FLAG_* */
    int32_t device;              /* HIP device ordinal */
    uint64_t seed;               /* base seed; game uid u draws from stream seed+u */
    /* multi-GPU sharding: the j-th game this engine starts (j = slot + n_games * games
     * the slot has started before) has global index uid = j * game_index_stride + game_index_offset.
     * Rank r of W passes (W, r): a fixed seed then plays the same set of games whatever W is.
     * 0 stride = 1 (single engine). */
    int32_t game_index_stride;
    int32_t game_index_offset;
} azx_config;

typedef struct azx_engine azx_engine;

const char *azx_last_error(void);
int azx_version(void);            /* ABI revision: 6 = azx_reserve_cus, azx_replay_put_records_async (self-play beside training);
                                   * 5 = AZX_ERANGE, azx_debug_weights, weights packed on the device
                                   * (4 = 8-float row metrics, azx_kernel_info, azx_debug_set_queue_cap;
                                   *  3 = azx_config.game_index_*, azx_play_stats.sum_game_length) */

/* Policy.initialize / Policy.reset (policy.py:36-63, :76-80): allocate device arenas. */
int azx_create(const azx_config *cfg, azx_engine **out);
void azx_destroy(azx_engine *e);

/* Policy.net weights (policy.py:65-74, network.py:42-61,:120-132): copy + pack the
 * state_dict tensors.  names[i] are state_dict keys; ptrs[i] point at contiguous fp32 data,
 * on the device (torch tensor.data_ptr()) when on_device != 0, else on the host. */
int azx_set_weights(azx_engine *e, int n_tensors, const char *const *names,
                    const void *const *ptrs, const int64_t *counts, int on_device);
/* The trainer changes its weights every step (policy_trainer.py:85-90) and the engine is refreshed on every
 * Player.read, so azx_set_weights folds / splits / re-orders ON THE DEVICE, reading device tensors in place
 * (host arrays are staged first); blocking.  AZX_ERANGE: a BatchNorm-folded weight is not finite or -- for the
 * split-f16 towers -- beyond the f16 range (65504); the message names the tensor, nothing is installed and the
 * engine has no weights until a valid set arrives.
 * Debug / tests: packed operand number `which` as the kernels read it -- *nbytes = its size (-1 past the last one),
 * its name in `name`, its bytes in `out` when cap suffices.  AZX_PACK=host (read at azx_create) selects the scalar
 * host reference packer; the two are bit-identical (tests/test_gpu_weights.py). */
int azx_debug_weights(azx_engine *e, int which, void *out, int64_t cap, int64_t *nbytes, char *name, int name_cap);

/* AZX_EVAL_UNIFORM_HASH only: prior_by_k[k] = the f32 prior a k-move position gets. */
int azx_set_prior_table(azx_engine *e, const float *prior_by_k, int count);

/* HexGame.reset + SearchTree.reset (hex.py:47-49, search_tree.py:59-71) for the listed
 * slots (NULL = all), then replay `n_moves[i]` moves from `moves` (row stride `stride`). */
int azx_reset(azx_engine *e, const int32_t *slots, int n_slots, const int32_t *moves,
              const int32_t *n_moves, int stride);

/* ---- one search, split the way mcts.sample_paths is (mcts.py:258-293) ------------------ */

/* Mark which slots take part in azx_search* / azx_advance (active[n_games], 0 = parked).  All
 * slots are active after azx_create / azx_reset.  A tournament (evaluation.py:46-80) searches, per
 * agent, only the games whose turn it is. */
int azx_set_active(azx_engine *e, const int32_t *active);

/* SearchTree.search for every active slot from its current root (search_tree.py:73-113).
 * noise: NULL -> no noise if noise_scale==0, else device RNG Dirichlet (throughput mode);
 * else host Dirichlet draws [n_games][n_select][noise_stride] f64, one row per select_leaf
 * (mcts.py:126-131) -- parity mode keeps numpy's RandomState on the host.
 * With AZX_EVAL_EXTERNAL this returns AZX_ESTATE: drive the phases below instead. */
int azx_search(azx_engine *e, const double *noise, int n_select, int noise_stride,
               double noise_scale);

/* external-evaluator phases (parity tests; also how a custom network would plug in):
 *   begin -> [pending>0: get_leaves, put_evals] -> repeat { step -> get_leaves -> put_evals }
 * azx_search_begin emits the root evaluation requests (mcts.py:272-273),
 * azx_search_step applies the pending evaluations (expand mcts.py:226-239, backup :242-255)
 * and, if batches remain, selects the next batch (mcts.py:46-76).  *n_pending receives the
 * number of positions waiting for evaluation; *done is 1 when all batches are applied. */
int azx_search_begin(azx_engine *e, const double *noise, int n_select, int noise_stride,
                     double noise_scale, int *n_pending);
int azx_search_step(azx_engine *e, int *n_pending, int *done);
/* pending positions in (slot, leaf order) order: boards[n][cells] int32 already flipped to
 * the first player's view (mcts.py:178-181), legal_moves[n][cells] padded with 0 (flipped
 * tiles, original order), slot[n], k[n]. */
int azx_get_leaves(azx_engine *e, int cap, int32_t *boards, int32_t *legal_moves,
                   int32_t *slot, int32_t *k, int *n_out);
/* value[n], prior[n][cells] (prior j = child j, first k entries) in the same order */
int azx_put_evals(azx_engine *e, int n, const float *value, const float *prior);
/* AZX_EVAL_RESNET driven through the phase API: the device network has already evaluated the
 * pending positions; read its (value, prior) back in azx_get_leaves order (call that first).
 * This is the "evaluation tape" parity tests replay through the CPU oracle. */
int azx_get_evals(azx_engine *e, int cap, float *value, float *prior, int *n_out);

/* ---- results ---------------------------------------------------------------------------- */

/* root.move_stats + root stats (search_tree.py:106-112, :192-204): per slot, child arrays
 * [n_games][cells] dense by child index (first k valid). Any pointer may be NULL. */
int azx_get_root(azx_engine *e, int32_t *k, int32_t *legal_moves, float *child_visits,
                 float *child_value, float *child_prior, float *root_visits,
                 float *root_value, int32_t *num_nodes, float *search_value);
/* per slot: 0 ok, 1 = the arena overflowed during the last search (SearchTreeFull,
 * search_tree.py:258-259): the caller skips the game as parallel_player.py:73-76 does */
int azx_get_status(azx_engine *e, int32_t *status);
/* per slot: SearchTree.num_nodes as the reference counts it (search_tree.py:112 'search_tree_nodes'): nodes
 * allocated since the tree's last reset and never reclaimed (search_tree.py:115-132) -- the arena compaction
 * here frees nodes, azx_get_root's num_nodes is the arena's live count. */
int azx_get_tree_nodes(azx_engine *e, int32_t *nodes);

/* game state per slot: HexGame.state (hex.py:55-60): board [n_games][cells] int32,
 * color (0/1), result (0/1/3), ply. */
int azx_get_games(azx_engine *e, int32_t *board, int32_t *color, int32_t *result,
                  int32_t *ply);

/* Policy.execute_action + HexGame.step (policy.py:170-176, hex.py:172-179): per slot the
 * chosen child index (move_id), or -1 to leave the slot untouched. */
int azx_advance(azx_engine *e, const int32_t *move_ids);

/* raw tree of one slot in the reference's six-array model (search_tree.py:48-55) */
int azx_tree_dump(azx_engine *e, int slot, int cap, int32_t *parent, int32_t *first_child,
                  int32_t *num_children, float *num_visits, float *total_value,
                  float *prior_prob, int32_t *num_nodes, int32_t *root_id);

/* ---- network only: Network.run inference branch (network.py:87-91,:103-105) ------------- */
int azx_forward(azx_engine *e, int B, int K, const int32_t *boards,
                const int32_t *legal_moves, float *value, float *moves_logprob);

/* ---- rules only: HexGameImpl.step/legal_moves/result over move lists (hex.py:151-179) --- */
/* moves[n_games][stride]; outputs per ply p < length: result after the move, number of
 * legal moves before it, and the empties bitmask before it (4 x u64 per ply).
 * board_size need not match any engine. */
int azx_hex_replay(int device, int board_size, int n_games, const int32_t *moves,
                   const int32_t *length, int stride, int32_t *result_out,
                   int32_t *nlegal_out, uint64_t *empties_out, int32_t *final_board);

/* ---- throughput mode: Player.read (parallel_player.py:24-28, :41-52) -------------------- */
typedef struct {
    int64_t positions;        /* rows written */
    int64_t games;            /* games finished (metrics['games']) */
    int64_t game_errors;      /* SearchTreeFull games skipped (parallel_player.py:73-76) */
    int64_t plies;            /* engine moves executed over all slots */
    int64_t selects;          /* select_leaf calls */
    int64_t evals;            /* positions evaluated */
    int64_t sum_depth, sum_k_interior, sum_k_leaf;   /* roofline byte model inputs */
    double  sum_search_value, sum_root_width, sum_action_logprob, sum_reward_last;
    double  seconds;          /* device time of the call (hipEvents on the engine stream) */
    double  mcts_seconds;     /* device time in the tree kernels */
    int64_t mcts_launches;    /* engine moves those launches covered (one per k_mcts search launch) */
    double  net_seconds;      /* device time in the network kernels (tower + heads), per batch */
    int64_t net_launches;
    int64_t mcts_kernel_launches;   /* kernel launches behind mcts_seconds (k_play: many moves each) */
    double  sum_game_length;  /* plies of the finished games, counted from the empty board (metrics: moves_per_game
                                 of games that started from an azx_reset prefix; rows only cover the plies searched) */
} azx_play_stats;

/* Self-play until >= min_positions rows from FINISHED games are available (whole games
 * only, like batch_examples) or max_plies engine steps have run (0 = unbounded).
 * Rows: board int32[cap][cells] (absolute colours), color[cap], nlegal[cap],
 * moves_prob f32[cap][cells] dense by child index, reward f32[cap], game_uid[cap]. */
int azx_play(azx_engine *e, int64_t min_positions, int64_t max_plies, int64_t cap,
             int32_t *board, int32_t *color, int32_t *nlegal, float *moves_prob,
             float *reward, int64_t *game_uid, azx_play_stats *stats);

/* per-ply search metrics of the rows the last azx_play / azx_play_device / azx_replay_fill call harvested, in row order:
 * metrics[n][AZX_ROW_METRICS] = {search_value (mcts.py:291), search_root_width (search_tree.py:110), log-probability
 * of the move drawn (play_game.py:43), 1 on the first row of a game else 0, search_root_visits (mean child
 * visits, search_tree.py:110), search_tree_nodes (nodes allocated since the tree's last reset, never reclaimed:
 * search_tree.py:112), search_root_children (search_tree.py:111), 0}.
 * play_game averages them over a game's plies and Player.read sums those means over the games it returns
 * (play_game.py:73-76, parallel_player.py:50-51). */
int azx_play_row_metrics(azx_engine *e, int64_t cap, float *metrics, int64_t *n_out);

/* bench hook: run `plies` lock-step engine moves on all slots (device RNG, finished games
 * restart in place), no row transfer; fills stats. */
int azx_play_steps(azx_engine *e, int64_t plies, azx_play_stats *stats);

/* ---- device-resident replay buffer ------------------------------------------
 * Replaces, for a trainer that keeps its minibatches on the GPU: ReplayBuffer.put's wrap-around
 * FIFO (azalea/replay_buffer.py:134-149), Player.read feeding it (parallel_player.py:41-52,
 * replay_buffer.py:121-132) and the DataLoader + prep.torch_batch_replays collate
 * (policy_trainer.py:51-56, prep.py:24-39).  Rows live in a fixed-capacity ring in HBM.
 * The fresh-example accounting of ReplayBuffer.consume (a float counter) stays with the caller. */

/* allocate (or replace) the ring: `capacity` rows, empty */
int azx_replay_create(azx_engine *e, int64_t capacity);
/* capacity, rows held (<= capacity) and the next write position */
int azx_replay_state(azx_engine *e, int64_t *capacity, int64_t *size, int64_t *write_idx);
/* restore size/write position (ReplayBuffer.load_state_dict, replay_buffer.py:160-165) */
int azx_replay_set_state(azx_engine *e, int64_t size, int64_t write_idx);
/* ReplayBuffer.put of n host rows (same row layout as azx_play's outputs): written at the write
 * position in order, wrapping; when n > capacity only the last `capacity` rows survive, as with
 * the reference's recursive put. */
int azx_replay_put(azx_engine *e, int64_t n, const int32_t *board, const int32_t *color,
                   const int32_t *nlegal, const float *moves_prob, const float *reward);
/* Player.read(min_positions) + ReplayBuffer.put without a host round trip: plays whole games
 * (throughput mode, like azx_play) until >= min_positions rows were harvested and moves them
 * into the ring.  *rows_out = rows added. */
int azx_replay_fill(azx_engine *e, int64_t min_positions, int64_t max_plies, int64_t *rows_out,
                    azx_play_stats *stats);
/* ---- multi-GPU replay exchange -------------------------------------------------------
 * Replaces the reference's only data-parallel return path -- pickled ReplayDataFrames over worker pipes
 * (process_pool.py:31-47, parallel_player.py:41-52) -- by one all-gather of fixed-size records between
 * DEVICE buffers: every rank plays its share (azx_play_device), packs the harvested rows into records
 * (azx_rows_pack), the caller all-gathers the record buffers (RCCL over xGMI) and appends all of them to
 * its ring (azx_replay_put_records).  Record layout (AZX_RECORD_BYTES(cells) bytes):
 *   0 game_uid i64 | 8 reward f32 | 12 color i16 | 14 nlegal i16 | 16 moves_prob f32[cells] |
 *   16+4*cells board u8[cells] | zero padding to a multiple of 16. */

/* Player.read(min_positions) that leaves the rows in the engine's harvest queue (whole games only,
 * like azx_play); *rows_out = rows now queued, valid until the next azx_play* / azx_replay_fill call. */
int azx_play_device(azx_engine *e, int64_t min_positions, int64_t max_plies, int64_t *rows_out,
                    azx_play_stats *stats);
/* queue rows [first, first+n) -> records in the DEVICE buffer records_dev (n * AZX_RECORD_BYTES). Blocking. */
int azx_rows_pack(azx_engine *e, int64_t first, int64_t n, void *records_dev);
/* ReplayBuffer.put of n records held in a DEVICE buffer: FIFO with the wrap-around / overflow behaviour
 * of azx_replay_put.  Blocking. */
int azx_replay_put_records(azx_engine *e, int64_t n, const void *records_dev);
/* The same put ENQUEUED on the caller's stream (a hipStream_t) and not synchronised: ordered with the
 * azx_replay_collate_async reads and the training step on that stream, and independent of the engine's own stream --
 * so a trainer thread can take rows into the ring while another host thread is inside azx_play_device on the same
 * handle (the one pairing of concurrent calls a handle allows; play-ahead self-play, replay_buffer.py:121-132 over
 * process_pool.py:29-47).  records_dev must stay valid until the stream has passed this point. */
int azx_replay_put_records_async(azx_engine *e, int64_t n, const void *records_dev, void *hip_stream);
/* Leave `cus_per_xcd` compute units of every XCD free of this engine's kernels (0 = use all): the engine's streams are
 * re-made with a CU mask (hipExtStreamCreateWithCUMask), so a training step on another stream finds empty CUs at
 * once instead of queueing behind resident tower blocks.  Rounded up to a multiple of 4: an XCD deals workgroups
 * round-robin to its 4 shader engines, so taking CUs from fewer than all of them costs the same throughput as taking
 * one from each (tools/microbench/cu_mask.hip).  Call between plays (the streams are drained).  *reserved_out = CUs
 * actually left free on the whole device. */
int azx_reserve_cus(azx_engine *e, int cus_per_xcd, int *reserved_out);

/* prep.batch_replays of the rows `indices[0..batch)` (host array; each < rows held) into DEVICE
 * buffers with row stride board_size^2: color i64[batch], legal_moves i32[batch][cells] (ascending
 * tile+1, zero padded), result i64[batch] (always 0), board i32[batch][cells],
 * moves_prob f32[batch][cells] (zero padded), reward f32[batch].  *max_k_out (host) = the batch's
 * largest legal-move count: the reference's collate pads legal_moves/moves_prob to exactly that
 * width, so callers slice [:, :max_k].  Blocking. */
int azx_replay_collate(azx_engine *e, int64_t batch, const int64_t *indices, int64_t *color_dev,
                       int32_t *legal_moves_dev, int64_t *result_dev, int32_t *board_dev,
                       float *moves_prob_dev, float *reward_dev, int32_t *max_k_out);

/* azx_replay_collate enqueued on the CALLER's stream (hipStream_t; NULL = default) and NOT synchronised, without
 * max_k (the consumer takes full-width rows): a trainer whose step runs on that stream (azx_train_step) queues collate
 * + step and moves on.  `indices` is copied before the call returns.  Ring writes (azx_replay_fill / azx_replay_put*,
 * blocking calls on the engine's stream) must be ordered after these reads by the caller: synchronise the stream
 * before a refill. */
int azx_replay_collate_async(azx_engine *e, int64_t batch, const int64_t *indices, int64_t *color_dev,
                             int32_t *legal_moves_dev, int64_t *result_dev, int32_t *board_dev,
                             float *moves_prob_dev, float *reward_dev, void *hip_stream);

/* NOT the reference's batch (off by default).  The reference trains on the boards as the replay rows hold them --
 * absolute colours, the second player's positions included (policy_trainer.py:84-85 -> network.py:92-102: no use of
 * `color`) -- while its search hands the network every position in the FIRST player's view (mcts.py:178-181:
 * flip_player_board_moves on the rows with color == 1).  With on != 0 both collates hand out the rows of the second
 * player in that view: colours swapped, board mirrored along the anti-diagonal, each legal move mapped with it in its
 * original list position (moves_prob stays aligned; `color` still reports the mover).  The training distribution then
 * is the distribution the search evaluates. */
int azx_replay_set_mover_view(azx_engine *e, int on);

/* float32 arithmetic self-test (tests): the tree kernels need IEEE-rounded sqrt and divide and
 * no FMA contraction (mcts.py:132-135).  sq=sqrtf(a), dv=a/(1+b), mul=(0.75f*a)*b+a. */
int azx_selftest_arith(int device, int n, const float *a, const float *b, float *sq, float *dv,
                       float *mul);

/* self-test of the two shortcuts in the search kernel's score path (mcts.py:132-134): quot[i] =
 * num[i]/den[i] through its unscaled reciprocal-refine divide, and sqrt_tab[i] = the constant-memory
 * sqrt table entry of the integer den[i] (0 beyond the table).  Both must equal the IEEE results
 * for den in [1, 2^24] and num 0 or 2^-100 <= |num| <= 2^100. */
int azx_selftest_divide(int device, int n, const float *num, const float *den, float *quot,
                        float *sqrt_tab);

/* throughput mode draws its Dirichlet noise on the device (mcts.py:128 uses numpy): n_rows
 * draws of Dirichlet(alpha * 1_k), k <= 128, exactly as the search kernel generates them, for
 * distribution tests. */
int azx_selftest_dirichlet(int device, double alpha, int k, int n_rows, uint32_t seed, float *out);

/* throughput mode's move draw on its own (tests): runs the device stand-in for as_distribution +
 * rng.multinomial (search_tree.py:327-344, policy.py:142-160) on every active slot's CURRENT root
 * statistics (call azx_search first) and returns the child index it drew, move_id[n_games] (-1: no draw),
 * and the moves_prob row it recorded for the replay, moves_prob[n_games][cells] dense by child index.
 * The slots are left with that row appended and the move pending: azx_reset them afterwards. */
int azx_debug_choose(azx_engine *e, int32_t *move_id, float *moves_prob);

/* raw device counters (16 x u64) since engine creation: selects, sum_depth, sum_k_interior,
 * sum_k_leaf, evals, terminal evals, games, errors, plies, rows, then diagnostic slots */
int azx_debug_counters(azx_engine *e, uint64_t *out16);
/* the same counters per game slot, not summed: out[n_games][16] (diagnostics: load balance) */
int azx_debug_counters_raw(azx_engine *e, uint64_t *out, int64_t n_games);

/* Which kernels this engine launches, as one line of text ("tree=... play=... tower=... heads=..."): the
 * diagnostic switches AZX_MCTS_GENERIC / AZX_NO_PERSISTENT / AZX_TOWER /
 * AZX_WIDE_STREAMS are read ONCE, by azx_create, into the engine; this reports what they selected so a run
 * can prove which kernels it used.  Returns the length of the full text (it is truncated to cap - 1 bytes). */
int azx_kernel_info(azx_engine *e, char *buf, int cap);
/* tests: bound the harvest queue of the following azx_play* calls to `rows` rows (0 = no bound) so that finished
 * games find it full and park (parallel_player.py has no counterpart: its pipes block instead). */
int azx_debug_set_queue_cap(azx_engine *e, int64_t rows);

/* ---- the training step on the device ---------------------------------------------------------
 * Replaces policy_trainer.supervised_step(train=True) (azalea/policy_trainer.py:123-142: zero_grad, Network.run with
 * compute_loss, backward, optimizer.step) for HexNetwork (network.py:68-102, :120-152) under torch.optim.SGD
 * (momentum, weight decay): forward in TRAIN mode (BatchNorm on batch statistics, running statistics and
 * num_batches_tracked updated), the reference's loss, backward, and the SGD update written IN PLACE into the caller's
 * parameter and momentum tensors -- hand-written MFMA kernels queued on the caller's stream (the convolutions on the
 * split-f16 arithmetic of the self-play tower, fp32 accumulate, operands scaled per layer by powers of two: results
 * at fp32 accuracy whatever the magnitudes; environment AZX_TRAIN_FWD / _BWD / _WGRAD=fp32 selects exact-fp32 MFMA
 * kernels per pass).  The trainer keeps owning its tensors (PyTorch holds them); this handle owns the activations
 * and scratch.  16 / 32 / 64 channels on boards up to 11x11, and 128 / 256 channels on boards from 3x3 to 13x13 -- and
 * 64 channels on 12x12 / 13x13 -- through the
 * wide step (the self-play wide convolution kernel in its TRAIN modes for forward and backward-data, elementwise
 * BatchNorm / ReLU passes on split-f16 images); any batch (built for the reference's 128); other shapes are
 * AZX_EINVAL.  PERMANENT LIMIT of this ABI revision: 16 / 32 channels on 12x12 / 13x13 boards -- the narrow kernels
 * tile a board as 128 position rows and keep it whole in LDS beside 1024-wide head planes; 144 / 169 cells need the wide
 * step's 176-row tiling, whose convolution blocks are 64 output channels wide.  policy_trainer.train runs the stock
 * PyTorch step there (the reference's own, shape-agnostic: policy_trainer.py:123-142), correct and ~4x slower.  The
 * reference's width (64) and BASELINE's configs (6x64 on 11x11, 19x256 on 13x13) are covered on every board. */
typedef struct {
    int32_t board_size, num_blocks, base_chans;   /* policy.py:51-53 */
    int32_t batch_size;                           /* config batch_size (hex11_train_config.yml: 128) */
    int32_t device;
} azx_train_config;
typedef struct azx_trainer azx_trainer;
int azx_train_create(const azx_train_config *cfg, azx_trainer **out);
void azx_train_destroy(azx_trainer *t);
/* Every state_dict entry of the module by name (network.py:42-61, :120-132), as DEVICE pointers that stay valid and
 * are updated in place: parameters (fp32) with their SGD momentum buffers momentum[i] (fp32, same shape, zero before
 * the first step = torch's lazily created buffer), BatchNorm running_mean / running_var (fp32) and
 * num_batches_tracked (int64) with momentum[i] = NULL.  Call again after the tensors were re-allocated. */
int azx_train_bind(azx_trainer *t, int n, const char *const *names, void *const *tensors, const int64_t *counts,
                   void *const *momentum);
/* the step's static input buffers (device), row stride board_size^2, the layout azx_replay_collate writes:
 * board i32[B][cells], legal_moves i32[B][cells] (ascending tile + 1, zero padded), moves_prob f32[B][cells] (by child
 * index, zero padded), reward f32[B] */
int azx_train_inputs(azx_trainer *t, int32_t **board, int32_t **legal_moves, float **moves_prob, float **reward);
/* device buffers holding the last step's results: loss f32[3] = {total, value, moves} (network.py:92-102),
 * value f32[B], moves_logprob f32[B][cells] (entry j = log-probability of legal move j; padding as the reference's
 * -99 logits) */
int azx_train_outputs(azx_trainer *t, float **loss3, float **value, float **moves_logprob);
/* One optimizer step on the bound tensors with the inputs currently in the input buffers, enqueued on `hip_stream`
 * (a hipStream_t; NULL = the default stream) and NOT synchronised: order it after whatever filled the inputs and
 * before whatever reads the weights.  lr / momentum / weight_decay: torch.optim.SGD's (policy_trainer.py:44-49). */
int azx_train_step(azx_trainer *t, float lr, float momentum, float weight_decay, void *hip_stream);
/* tests: internal buffer by name ("raw<l>", "act<l>", "g<l>" [B][cells][C]; "sums"; "grad:<state_dict name>") */
int azx_train_debug(azx_trainer *t, const char *name, void *out, int64_t cap, int64_t *nbytes);

/* engine stream (hipStream_t) so callers can bracket work with HIP events */
void *azx_stream(azx_engine *e);

#ifdef __cplusplus
}
#endif
#endif /* AZX_H */
