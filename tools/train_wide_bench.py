#!/usr/bin/env python3
"""The training step at a wide shape (default 19x256 on 11x11, batch 128): hand-written (NativeTrainStep) against the
stock kernels captured as a HIP graph (GraphedTrainStep) and the eager step, device time by events.
    python3 tools/train_wide_bench.py [--blocks 19] [--chans 256] [--board 11] [--batch 128] [--steps 20] [--modes native,hip_graph,eager]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from azalea_amd.network import HexNetwork


def batch_of(n, B, dev, seed=0):
    rng = np.random.RandomState(seed)
    cells = n * n
    board = rng.randint(0, 3, (B, n, n)).astype(np.int32)
    board[rng.rand(B, n, n) < 0.4] = 0
    lm = np.zeros((B, cells), np.int32)
    mp = np.zeros((B, cells), np.float32)
    for i in range(B):
        e = np.flatnonzero(board[i].ravel() == 0) + 1
        lm[i, :len(e)] = e
        mp[i, :len(e)] = rng.dirichlet(np.full(len(e), 0.3))
    return dict(board=torch.tensor(board, device=dev), legal_moves=torch.tensor(lm, device=dev),
                moves_prob=torch.tensor(mp, device=dev), reward=torch.tensor(rng.choice([-1.0, 1.0], B).astype(np.float32), device=dev))


def timed(fn, steps, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=19)
    ap.add_argument("--chans", type=int, default=256)
    ap.add_argument("--board", type=int, default=11)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--modes", default="native,hip_graph,eager")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    n, B, C, L = a.board, a.batch, a.chans, 2 * a.blocks
    flop = 3.0 * L * B * n * n * C * C * 9 * 2            # the tower's three passes (SURVEY 8(d) counting)
    out = {"shape": "%dx%d on %dx%d, batch %d" % (a.blocks, C, n, n, B), "tower_flop_per_step": flop}
    for mode in a.modes.split(","):
        torch.manual_seed(0)
        net = HexNetwork(board_size=n, num_blocks=a.blocks, base_chans=C).to(dev)
        opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
        batch = batch_of(n, B, dev)
        if mode == "native":
            from azalea_amd.native_train import NativeTrainStep
            step = NativeTrainStep(net, opt, B, dev)
            step.step(batch)
            ms = timed(step._run, a.steps, 3)
            loss = step.loss.cpu().numpy().tolist()
            step.close()
        elif mode == "hip_graph":
            from azalea_amd.policy_trainer import GraphedTrainStep
            step = GraphedTrainStep(net, opt, B, dev)
            for _ in range(5):
                step.step(batch)
            ms = timed(step._run, a.steps, 2)
            loss = step.loss.cpu().numpy().tolist()
        else:
            from azalea_amd.policy_trainer import supervised_step
            ms = timed(lambda: supervised_step(net, dict(batch), train=True, optimizer=opt, device=dev), a.steps, 3)
            loss = None
        out[mode] = {"ms_per_step": ms, "steps_per_sec": 1e3 / ms, "algorithmic_tflops": flop / (ms * 1e-3) / 1e12,
                     "frac_of_f16_mfma_peak": flop / (ms * 1e-3) / 2.5e15, "loss": loss}
    if "native" in out and "hip_graph" in out:
        out["speedup_native_vs_hip_graph"] = out["hip_graph"]["ms_per_step"] / out["native"]["ms_per_step"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
