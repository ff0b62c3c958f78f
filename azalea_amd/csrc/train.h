// train.h -- the reference's training step (azalea/policy_trainer.py:123-142 over network.py:68-102) as hand-written
// gfx950 kernels (train_kernels.hip): interface used by azx_capi.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct AzxTrain;

const char *azx_trn_error();
int azx_trn_create(AzxTrain **out, int N, int blocks, int chans, int batch, int device);
void azx_trn_destroy(AzxTrain *t);
// every state_dict entry by name: parameters with their SGD momentum buffers (fp32, same shapes), BatchNorm running
// statistics (fp32) and num_batches_tracked (int64) with momentum[i] = NULL.  All device pointers, used in place.
int azx_trn_bind(AzxTrain *t, int n, const char *const *names, void *const *ptrs, const int64_t *counts,
                 void *const *momentum);
// the step's static input buffers: board i32[B][cells], legal_moves i32[B][cells], moves_prob f32[B][cells], reward f32[B]
int azx_trn_inputs(AzxTrain *t, int32_t **board, int32_t **legal_moves, float **moves_prob, float **reward);
// outputs of the last step (device): loss f32[3] = total, value, moves; value f32[B]; moves_logprob f32[B][cells]
int azx_trn_outputs(AzxTrain *t, float **loss3, float **value, float **logprob);
// one optimizer step on the bound tensors, enqueued on `st` (not synchronised)
int azx_trn_step(AzxTrain *t, float lr, float momentum, float weight_decay, hipStream_t st);
// internal buffers by name, for layer-by-layer tests ("raw3", "act3", "g3", "sums", "grad:<tensor>", ...)
int azx_trn_debug(AzxTrain *t, const char *name, void *out, int64_t cap, int64_t *nbytes);
