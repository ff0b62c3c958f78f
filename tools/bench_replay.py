"""Measurement for the device replay ring: rows/s and HBM GB/s of the FIFO put of
harvested games and of the minibatch collate, against the 8 TB/s HBM peak.  One JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from azalea_amd import engine as eng
from azalea_amd.device_replay import DeviceReplayBuffer

CELLS, STRIDE = 121, 192
ROW_RING = STRIDE + 4 * STRIDE + 12                      # board u8 + prob f32 + color/k/reward
ROW_OUT = CELLS * 12 + 20                                # legal i32 + board i32 + prob f32 + color/result i64 + reward


def main():
    E = eng.Engine(board_size=11, n_games=4096, simulations=20, search_batch_size=10,
                   evaluator=eng.EVAL_UNIFORM, noise_scale=0.25, seed=7)
    cap = 1 << 20
    buf = DeviceReplayBuffer(E, cap)
    t0 = time.perf_counter()
    rows, st = E.replay_fill(300000)
    fill_s = time.perf_counter() - t0
    out = {"ring_rows": rows, "fill_seconds_incl_selfplay": fill_s}
    for B in (1024, 65536):
        idx = np.random.RandomState(1).randint(0, rows, B)
        for _ in range(3):
            buf.sample(idx)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            buf.sample(idx)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        gbs = B * (ROW_RING + ROW_OUT) / dt / 1e9
        out["collate_B%d" % B] = {"ms": 1e3 * dt, "rows_per_s": B / dt, "algorithmic_GBps": gbs,
                                  "hbm_frac": gbs / 8000.0}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
