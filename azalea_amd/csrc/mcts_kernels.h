// mcts_kernels.h -- launchers of the tree/rules kernels (mcts_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "azx_dev.h"

#define MODE_BEGIN 1    // start of a search: reset counters, evaluate the root if needed
#define MODE_APPLY 2    // expand + backup the pending leaves (evaluations are in ev_*)
#define MODE_SELECT 4   // select the next batch and emit evaluation requests
#define MODE_INLINE 8   // evaluator is inline (uniform priors): loop all batches in one launch

size_t azx_mcts_lds_bytes(int ncells, int bs);
void azx_launch_mcts(const DevEngine &E, int mode, int num_batches, hipStream_t st, bool force_generic);
bool azx_mcts_fast_path(const DevEngine &E, int mode, bool force_generic);   // would this launch take k_mcts<S, FAST>?
// whole moves in one launch (k_play); returns false (nothing launched) when not `allowed` or not applicable
bool azx_launch_play(const DevEngine &E, int num_batches, int steps, hipStream_t st, bool allowed);
void azx_launch_reset(const DevEngine &E, const int32_t *slots, int n_slots, const int32_t *moves,
                      const int32_t *n_moves, int stride, int assign_uid, hipStream_t st);
void azx_launch_advance(const DevEngine &E, const int32_t *move_ids, int play_mode, hipStream_t st);
void azx_launch_gather_root(const DevEngine &E, int32_t *k_out, int32_t *legal, float *cv, float *cw,
                            float *cp, float *rv, float *rw, int32_t *nn, float *sv, hipStream_t st);
void azx_launch_choose(const DevEngine &E, hipStream_t st);
void azx_launch_hex_replay(int N, int n_games, const int32_t *moves, const int32_t *length, int stride,
                           int32_t *result_out, int32_t *nlegal_out, uint64_t *empties_out,
                           int32_t *final_board, hipStream_t st);
void azx_launch_arith(const float *a, const float *b, float *sq, float *dv, float *mul, int n,
                      hipStream_t st);
void azx_launch_divide_test(const float *num, const float *den, float *q, float *rt, int n, hipStream_t st);
void azx_launch_noise_test(float alpha, const float *tab, int k, int n_rows, uint32_t seed, float *out, hipStream_t st);
// inverse-CDF correction table of the device gamma sampler for one alpha: AZX_GAMMA_TAB_FLOATS floats
#define AZX_GAMMA_TAB_FLOATS 802
void azx_gamma_table(double alpha, float *tab);
int azx_init_geometry(int device);   // board geometry tables -> constant memory (once per device)
