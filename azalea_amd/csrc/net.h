// net.h -- HexNetwork inference forward on gfx950 (net_kernels.hip): interface used by azx_capi.cpp
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "azx_dev.h"

struct AzxNet;

int azx_net_create(AzxNet **out, int N, int blocks, int chans, int max_evals, hipStream_t st);
void azx_net_destroy(AzxNet *net);
// the engine re-made its stream (azx_reserve_cus): use `st` from now on and make the wide tower's side streams on the
// same CU mask (`mask` words, 0 words = all CUs); the old side streams are drained and dropped
void azx_net_set_stream(AzxNet *net, hipStream_t st, const uint32_t *mask, int words);
const char *azx_net_error();
int azx_net_set_weights(AzxNet *net, int n, const char *const *names, const void *const *ptrs,
                        const int64_t *counts, int on_device);
bool azx_net_ready(const AzxNet *net);
// AZX_ERANGE when a split-f16 tower launch since the last check saw an activation beyond the f16 range (blocks on st)
int azx_net_check_range(AzxNet *net, hipStream_t st);
// packed buffer number `which` (name, bytes, contents); *nbytes = -1 past the last one
int azx_net_debug_weights(AzxNet *net, int which, void *out, int64_t cap, int64_t *nbytes, char *name, int name_cap);
const char *azx_net_kernel_info(const AzxNet *net);   // which tower / heads kernels this net launches
// evaluate the packed requests ev_board[0 .. *d.n_eval) -> ev_value, ev_prior (by original cell)
void azx_net_eval(AzxNet *net, const DevEngine &d, hipStream_t st);
// Network.run on host arrays (network.py:87-105): value[B], moves_logprob[B][K]
int azx_net_forward_host(AzxNet *net, int B, int K, const int32_t *boards,
                         const int32_t *legal_moves, float *value, float *logprob, hipStream_t st);

// One 3x3 convolution of a wide tower (C a multiple of 128, boards up to 13x13) for the TRAINING step (train_wide.hip):
// k_conv_wide_f16x3_s16's staging and k-loop on a split-f16 image `in` [boards][cells][C hi | C lo] and one layer's
// fragments `w16` in the wide pack; writes the raw fp32 product times `*unscale` (a device word: the scales are made on
// the device) to out32 [boards][cells][C] and, when `stat` is given, per-board per-channel (sum, sum of squares) pairs
// [boards][C].
int azx_net_wide_train_conv(int N, int C, const unsigned short *w16, const unsigned short *in, float *out32, int n_boards,
                            const float *unscale, float2 *stat, hipStream_t st);
// The backward-data convolution of the same step with the next elementwise pass fused into its epilogue: writes
// g_{l-1} = (conv^T product [+ skip]) masked by act_{l-1} > 0 (`mask`: one bit per element, [boards][cells][C / 8] bytes, k_tw_bnact's), this board's (sum g, sum g xhat_{l-1}) pairs to pgsum
// [boards][C] (xhat from raw_{l-1} and BN_{l-1}'s batch sums `sums` [C][4 doubles], invN = 1 / (boards x cells)) and
// max |g| into *gmax (atomicMax on the float's bits).
int azx_net_wide_train_conv_bwd(int N, int C, const unsigned short *w16, const unsigned short *in, float *g_out, int n_boards,
                                const float *unscale, float2 *pgsum, const unsigned char *mask, const float *raw, const float *skip,
                                const double *sums, float invN, unsigned int *gmax, hipStream_t st);
