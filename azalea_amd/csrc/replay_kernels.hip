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
__global__ __launch_bounds__(64) void k_replay_collate(ReplayRows ring, const long long *idx, int B,
                                                       int ncells, long long *color, int32_t *legal,
                                                       long long *result, int32_t *board,
                                                       float *prob, float *reward, int32_t *max_k) {
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    if (b >= B) return;
    const long long r = idx[b];
    const uint8_t *rb = ring.board + r * AZX_CELL_STRIDE;
    const float *rp = ring.prob + r * AZX_CELL_STRIDE;
    const int k = ring.k[r];
    int base = 0;
#pragma unroll
    for (int s = 0; s < AZX_CELL_STRIDE / 64; ++s) {
        const int c = s * 64 + lane;
        const bool on = c < ncells;
        const int v = on ? (int)rb[c] : 3;
        const unsigned long long empties = __ballot(v == 0);
        if (on) {
            board[(size_t)b * ncells + c] = v;
            prob[(size_t)b * ncells + c] = c < k ? rp[c] : 0.0f;
            if (c >= k) legal[(size_t)b * ncells + c] = 0;          // padding (prep.py:70-86)
        }
        if (v == 0) {
            const int rank = base + __builtin_amdgcn_mbcnt_hi((unsigned)(empties >> 32),
                                                              __builtin_amdgcn_mbcnt_lo((unsigned)empties, 0u));
            if (rank < k) legal[(size_t)b * ncells + rank] = c + 1;
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

void azx_launch_replay_put(const ReplayRows &src, const ReplayRows &ring, long long n, long long cap,
                           long long write_idx, hipStream_t st) {
    const long long skip = n > cap ? n - cap : 0;
    if (n - skip <= 0) return;
    hipLaunchKernelGGL(k_replay_put, dim3((unsigned)(n - skip)), dim3(64), 0, st, src, ring, n, skip, cap, write_idx);
}

void azx_launch_replay_collate(const ReplayRows &ring, const long long *idx, int B, int ncells,
                               long long *color, int32_t *legal, long long *result, int32_t *board,
                               float *prob, float *reward, int32_t *max_k, hipStream_t st) {
    if (B <= 0) return;
    hipLaunchKernelGGL(k_replay_collate, dim3(B), dim3(64), 0, st, ring, idx, B, ncells, color, legal,
                       result, board, prob, reward, max_k);
}
