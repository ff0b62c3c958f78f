// replay_kernels.h -- launchers of the device replay ring kernels (replay_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/azx.h"

struct ReplayRows {          // struct of arrays over replay rows (harvest queue or ring)
    uint8_t *board;          // [rows][AZX_CELL_STRIDE]
    float *prob;             // [rows][AZX_CELL_STRIDE]
    int32_t *color, *k;      // [rows]
    float *reward;           // [rows]
};

void azx_launch_replay_put(const ReplayRows &src, const ReplayRows &ring, long long n, long long cap,
                           long long write_idx, hipStream_t st);
void azx_launch_replay_collate(const ReplayRows &ring, const long long *idx, int B, int ncells,
                               long long *color, int32_t *legal, long long *result, int32_t *board,
                               float *prob, float *reward, int32_t *max_k, int mover_n, hipStream_t st);
void azx_launch_rows_export(const uint8_t *qb, const float *qp, long long n, int ncells, int32_t *board,
                            float *prob, hipStream_t st);
void azx_launch_rows_pack(const ReplayRows &src, const long long *uid, long long first, long long n, int ncells,
                          uint8_t *rec, hipStream_t st);
void azx_launch_records_put(const uint8_t *rec, const ReplayRows &ring, long long n, long long cap,
                            long long write_idx, int ncells, hipStream_t st);
