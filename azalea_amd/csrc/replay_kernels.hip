// replay_kernels.hip -- device-resident replay FIFO and minibatch collate for gfx950 (MI355X).
//
// The reference keeps replay rows as Python lists on the host, overwrites the oldest rows on
// refill (azalea/replay_buffer.py:134-149 ReplayBuffer.put) and builds minibatches in DataLoader
// workers with prep.batch_replays (azalea/prep.py:24-39; policy_trainer.py:51-56).  Here the rows
// never leave HBM: self-play appends whole games to the engine's harvest queue, k_replay_put
// moves them into a fixed-capacity ring in FIFO order, and k_replay_collate gathers the rows a
// sampler picked into the dense, zero-padded batch tensors the training step reads.
//
// Row format (same as the harvest queue, azx_dev.h): board u8[AZX_CELL_STRIDE] absolute colours
// before the move, moves_prob f32[AZX_CELL_STRIDE] dense by child index (zero beyond k), colour
// of the mover (0/1), k legal moves, reward (mover's perspective).  Both kernels are pure
// HBM copies: one wavefront per row, 16-byte accesses where the layout allows.
#include "replay_kernels.h"

// rows [skip, n) of src go to ring slots (write_idx + i) mod cap.  Rows before `skip` would be
// overwritten by later rows of the same put (n > cap), exactly what the sequential reference
// put leaves behind.
__global__ __launch_bounds__(64) void k_replay_put(ReplayRows src, ReplayRows ring, long long n,
                                                   long long skip, long long cap, long long write_idx) {
    const int lane = threadIdx.x;
    const long long i = skip + blockIdx.x;
    if (i >= n) return;
    const long long dst = (write_idx + i) % cap;
    // board: 192 B = 48 dwords; moves_prob: 192 floats = 48 float4
    if (lane < AZX_CELL_STRIDE / 4) {
        reinterpret_cast<uint32_t *>(ring.board + dst * AZX_CELL_STRIDE)[lane] =
            reinterpret_cast<const uint32_t *>(src.board + i * AZX_CELL_STRIDE)[lane];
        reinterpret_cast<float4 *>(ring.prob + dst * AZX_CELL_STRIDE)[lane] =
            reinterpret_cast<const float4 *>(src.prob + i * AZX_CELL_STRIDE)[lane];
    }
    if (lane == 0) {
        ring.color[dst] = src.color[i];
        ring.k[dst] = src.k[i];
        ring.reward[dst] = src.reward[i];
    }
}

// prep.batch_replays for the rows idx[0..B): one wavefront per batch row.  Outputs have row
// stride `ncells` (the caller slices [:, :max_k], the batch maximum prep.pad would pad to):
//   color i64[B], legal_moves i32[B][ncells] (ascending tile+1 of the empty cells, hex.py:151-159,
//   zero padded), result i64[B] (0: replay rows are positions of games in progress),
//   board i32[B][ncells], moves_prob f32[B][ncells] (zero padded), reward f32[B].
// mover_n > 0 (azx_replay_set_mover_view; NOT the reference's batch): the rows of the second player come out in the
// view the SEARCH evaluates them in (mcts.py:178-181, hex.py flip_player_board_moves) -- colours swapped, the board
// mirrored along the anti-diagonal, every legal move mapped with it in its original list position (so moves_prob
// stays aligned) -- on a board of mover_n x mover_n cells.
__global__ __launch_bounds__(64) void k_replay_collate(ReplayRows ring, const long long *idx, int B,
                                                       int ncells, long long *color, int32_t *legal,
                                                       long long *result, int32_t *board,
                                                       float *prob, float *reward, int32_t *max_k, int mover_n) {
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    if (b >= B) return;
    const long long r = idx[b];
    const uint8_t *rb = ring.board + r * AZX_CELL_STRIDE;
    const float *rp = ring.prob + r * AZX_CELL_STRIDE;
    const int k = ring.k[r];
    const bool flip = mover_n > 0 && ring.color[r] == 1;
    int base = 0;
#pragma unroll
    for (int s = 0; s < AZX_CELL_STRIDE / 64; ++s) {
        const int c = s * 64 + lane;
        const bool on = c < ncells;
        const int v = on ? (int)rb[c] : 3;
        const unsigned long long empties = __ballot(v == 0);
        int fc = c;                                                  // the cell's place in the view handed out
        if (flip && on) {
            const int row = c / mover_n, col = c - row * mover_n;
            fc = (mover_n - 1 - col) * mover_n + (mover_n - 1 - row);
        }
        if (on) {
            board[(size_t)b * ncells + fc] = flip && v ? 3 - v : v;
            prob[(size_t)b * ncells + c] = c < k ? rp[c] : 0.0f;
            if (c >= k) legal[(size_t)b * ncells + c] = 0;          // padding (prep.py:70-86)
        }
        if (v == 0) {
            const int rank = base + __builtin_amdgcn_mbcnt_hi((unsigned)(empties >> 32),
                                                              __builtin_amdgcn_mbcnt_lo((unsigned)empties, 0u));
            if (rank < k) legal[(size_t)b * ncells + rank] = fc + 1;
        }
        base += __popcll(empties);
    }
    if (lane == 0) {
        color[b] = ring.color[r];
        result[b] = 0;
        reward[b] = ring.reward[r];
        atomicMax(max_k, k);
    }
}

// azx_play's host hand-over: queue rows [0, n) widened on the device into the caller's layout (board
// i32[n][ncells], moves_prob f32[n][ncells], row stride ncells), so the host side is one copy per array.
__global__ __launch_bounds__(64) void k_rows_export(const uint8_t *qb, const float *qp, long long n, int ncells,
                                                    int32_t *board, float *prob) {
    const long long r = blockIdx.x;
    if (r >= n) return;
    for (int c = threadIdx.x; c < ncells; c += 64) {
        board[r * ncells + c] = (int32_t)qb[r * AZX_CELL_STRIDE + c];
        prob[r * ncells + c] = qp[r * AZX_CELL_STRIDE + c];
    }
}

// ---- fixed-size replay records (the multi-GPU exchange unit) -------------------------
// One record per row, AZX_RECORD_BYTES(ncells) bytes, 16-byte aligned fields:
//   0 uid i64 | 8 reward f32 | 12 color i16 | 14 k i16 | 16 moves_prob f32[ncells] | 16+4*ncells board u8[ncells] | pad
// Ranks pack their harvested rows (k_rows_pack), all-gather the records over xGMI (RCCL, device
// tensors) and every rank appends all records to its ring (k_records_put) -- no host in the middle.
__global__ __launch_bounds__(64) void k_rows_pack(ReplayRows src, const long long *uid, long long first, long long n,
                                                  int ncells, uint8_t *rec) {
    const long long i = blockIdx.x;
    if (i >= n) return;
    const long long r = first + i;
    const int lane = threadIdx.x;
    const size_t rb = AZX_RECORD_BYTES(ncells);
    uint8_t *o = rec + (size_t)i * rb;
    if (lane == 0) {
        *reinterpret_cast<long long *>(o) = uid ? uid[r] : -1ll;
        *reinterpret_cast<float *>(o + 8) = src.reward[r];
        *reinterpret_cast<int16_t *>(o + 12) = (int16_t)src.color[r];
        *reinterpret_cast<int16_t *>(o + 14) = (int16_t)src.k[r];
    }
    float *op = reinterpret_cast<float *>(o + 16);
    uint8_t *ob = o + 16 + 4 * (size_t)ncells;
    for (int c = lane; c < ncells; c += 64) {
        op[c] = src.prob[r * AZX_CELL_STRIDE + c];
        ob[c] = src.board[r * AZX_CELL_STRIDE + c];
    }
    for (size_t c = 16 + 5 * (size_t)ncells + lane; c < rb; c += 64) o[c] = 0;
}

// records [skip, n) enter the ring at (write_idx + i) mod cap, like k_replay_put
__global__ __launch_bounds__(64) void k_records_put(const uint8_t *rec, ReplayRows ring, long long n, long long skip,
                                                    long long cap, long long write_idx, int ncells) {
    const long long i = skip + blockIdx.x;
    if (i >= n) return;
    const int lane = threadIdx.x;
    const long long dst = (write_idx + i) % cap;
    const uint8_t *o = rec + (size_t)i * AZX_RECORD_BYTES(ncells);
    if (lane == 0) {
        ring.reward[dst] = *reinterpret_cast<const float *>(o + 8);
        ring.color[dst] = (int32_t)*reinterpret_cast<const int16_t *>(o + 12);
        ring.k[dst] = (int32_t)*reinterpret_cast<const int16_t *>(o + 14);
    }
    const float *ip = reinterpret_cast<const float *>(o + 16);
    const uint8_t *ib = o + 16 + 4 * (size_t)ncells;
    for (int c = lane; c < AZX_CELL_STRIDE; c += 64) {
        ring.prob[dst * AZX_CELL_STRIDE + c] = c < ncells ? ip[c] : 0.0f;
        ring.board[dst * AZX_CELL_STRIDE + c] = c < ncells ? ib[c] : (uint8_t)0;
    }
}

void azx_launch_rows_export(const uint8_t *qb, const float *qp, long long n, int ncells, int32_t *board,
                            float *prob, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_rows_export, dim3((unsigned)n), dim3(64), 0, st, qb, qp, n, ncells, board, prob);
}

void azx_launch_rows_pack(const ReplayRows &src, const long long *uid, long long first, long long n, int ncells,
                          uint8_t *rec, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_rows_pack, dim3((unsigned)n), dim3(64), 0, st, src, uid, first, n, ncells, rec);
}

void azx_launch_records_put(const uint8_t *rec, const ReplayRows &ring, long long n, long long cap,
                            long long write_idx, int ncells, hipStream_t st) {
    const long long skip = n > cap ? n - cap : 0;
    if (n - skip <= 0) return;
    hipLaunchKernelGGL(k_records_put, dim3((unsigned)(n - skip)), dim3(64), 0, st, rec, ring, n, skip, cap,
                       write_idx, ncells);
}

void azx_launch_replay_put(const ReplayRows &src, const ReplayRows &ring, long long n, long long cap,
                           long long write_idx, hipStream_t st) {
    const long long skip = n > cap ? n - cap : 0;
    if (n - skip <= 0) return;
    hipLaunchKernelGGL(k_replay_put, dim3((unsigned)(n - skip)), dim3(64), 0, st, src, ring, n, skip, cap, write_idx);
}

void azx_launch_replay_collate(const ReplayRows &ring, const long long *idx, int B, int ncells,
                               long long *color, int32_t *legal, long long *result, int32_t *board,
                               float *prob, float *reward, int32_t *max_k, int mover_n, hipStream_t st) {
    if (B <= 0) return;
    hipLaunchKernelGGL(k_replay_collate, dim3(B), dim3(64), 0, st, ring, idx, B, ncells, color, legal,
                       result, board, prob, reward, max_k, mover_n);
}
