#!/usr/bin/env python3
"""Run the hand-written training step (NativeTrainStep, 6x64 / 11x11 / batch 128 by default) a number of times on
random data -- the program to put behind `rocprofv3 --kernel-trace --stats --` for per-kernel times.
    python3 tools/train_native_run.py [--steps 50] [--batch 128] [--blocks 6] [--chans 64] [--board 11]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from azalea_amd.native_train import NativeTrainStep
from azalea_amd.network import HexNetwork


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--blocks", type=int, default=6)
    ap.add_argument("--chans", type=int, default=64)
    ap.add_argument("--board", type=int, default=11)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = HexNetwork(board_size=a.board, num_blocks=a.blocks, base_chans=a.chans).to(dev)
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    step = NativeTrainStep(net, opt, a.batch, dev)
    rng = np.random.RandomState(0)
    n, B, cells = a.board, a.batch, a.board * a.board
    board = rng.randint(0, 3, (B, n, n)).astype(np.int32)
    board[rng.rand(B, n, n) < 0.4] = 0
    lm = np.zeros((B, cells), np.int32)
    mp = np.zeros((B, cells), np.float32)
    for i in range(B):
        e = np.flatnonzero(board[i].ravel() == 0) + 1
        lm[i, :len(e)] = e
        mp[i, :len(e)] = rng.dirichlet(np.full(len(e), 0.3))
    batch = dict(board=torch.tensor(board, device=dev), legal_moves=torch.tensor(lm, device=dev),
                 moves_prob=torch.tensor(mp, device=dev), reward=torch.tensor(rng.choice([-1.0, 1.0], B).astype(np.float32), device=dev))
    step.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step._run()
    t_host = time.perf_counter() - t0          # the host has queued everything
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("native step: %.3f ms (%d steps; the host queued a step in %.3f ms), loss %s"
          % (1e3 * dt / a.steps, a.steps, 1e3 * t_host / a.steps, step.loss.cpu().numpy()))
    # what queueing ONE step costs the host when nothing is pending (no back-pressure from a full queue)
    costs = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step._run()
        costs.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    print("host cost of queueing one step on an idle device: %.3f ms (median of 5)" % (1e3 * sorted(costs)[2]))


if __name__ == "__main__":
    main()
