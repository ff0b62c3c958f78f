#!/usr/bin/env python3
"""SURVEY 8(f).4 measured: the reference's training step (policy_trainer.py:123-142: forward + loss of
network.py:92-102, backward, SGD with momentum and weight decay) on stock PyTorch-ROCm at the reference's
hyper-parameters (config/hex11_train_config.yml: 6x64 on 11x11, batch 128, 10x oversampling), fed from the HBM replay
ring (engine self-play -> azx_replay_fill -> azx_replay_collate -> device tensors).  Reports steps/s, the share of a
step spent collating, and the fresh rows per second the trainer consumes (batch / oversampling per step) beside what
one GPU's self-play produces -- i.e. who waits for whom when both share a GPU.
    python tools/bench_train_step.py [--batch 128 1024 4096] [--steps 200]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from torch import optim

from azalea_amd import engine as eng
from azalea_amd.device_replay import DeviceReplayBuffer
from azalea_amd.network import HexNetwork
from azalea_amd.policy_trainer import supervised_step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[128, 1024, 4096])
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=100000)
    ap.add_argument("--oversampling", type=float, default=10.0)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    # the ring is filled by uniform-prior self-play (the row format does not depend on the evaluator)
    E = eng.Engine(board_size=11, n_games=2048, simulations=50, search_batch_size=10, evaluator=eng.EVAL_UNIFORM,
                   noise_scale=0.25)
    buf = DeviceReplayBuffer(E, args.rows, shared=False)
    t0 = time.perf_counter()
    rows, st = E.replay_fill(args.rows)
    fill_s = time.perf_counter() - t0
    torch.manual_seed(0)
    out = {"ring_rows": len(buf), "fill_rows_per_sec": rows / fill_s, "lines": []}
    for B in args.batch:
        net = HexNetwork(board_size=11, num_blocks=6, base_chans=64).to(dev)
        opt = optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
        order = buf.epoch_indices()
        need = (args.steps + args.warmup) * B
        order = np.resize(order, need)
        collate = step = 0.0
        loss = 0.0
        for i in range(args.steps + args.warmup):
            if i == args.warmup:
                torch.cuda.synchronize()
                collate = step = 0.0
                t_all = time.perf_counter()
            a = time.perf_counter()
            batch = buf.sample(order[i * B:(i + 1) * B])
            torch.cuda.synchronize()
            b = time.perf_counter()
            _, loss = supervised_step(net, batch, train=True, optimizer=opt, device=dev)   # loss.item() syncs
            c = time.perf_counter()
            collate += b - a
            step += c - b
        total = time.perf_counter() - t_all
        out["lines"].append({"batch": B, "steps_per_sec": args.steps / total, "positions_per_sec": args.steps * B / total,
                             "ms_per_step": 1e3 * total / args.steps, "collate_share": collate / total,
                             "fresh_rows_per_sec_consumed": args.steps * B / args.oversampling / total,
                             "last_loss": float(loss)})
    # the same step without its three host syncs (run()'s two .item() calls and supervised_step's): the losses stay
    # on the device and are read once at the end, as a trainer that logs every log_interval steps would
    import torch.nn.functional as F
    for B in args.batch:
        net = HexNetwork(board_size=11, num_blocks=6, base_chans=64).to(dev).train()
        opt = optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
        order = np.resize(buf.epoch_indices(), (args.steps + args.warmup) * B)
        acc = torch.zeros((), dtype=torch.float32, device=dev)
        for i in range(args.steps + args.warmup):
            if i == args.warmup:
                torch.cuda.synchronize()
                t_all = time.perf_counter()
            batch = buf.sample(order[i * B:(i + 1) * B])
            opt.zero_grad()
            o = net.forward(batch["board"], batch["legal_moves"])
            loss = F.mse_loss(o["value"], batch["reward"]) - (batch["moves_prob"] * o["moves_logprob"]).sum() / B
            loss.backward()
            opt.step()
            acc += loss.detach()
        torch.cuda.synchronize()
        total = time.perf_counter() - t_all
        out["lines"].append({"batch": B, "mode": "no per-step host sync", "steps_per_sec": args.steps / total,
                             "positions_per_sec": args.steps * B / total, "ms_per_step": 1e3 * total / args.steps,
                             "fresh_rows_per_sec_consumed": args.steps * B / args.oversampling / total,
                             "mean_loss": float(acc.item()) / (args.steps + args.warmup)})
    # the step captured as a HIP graph (policy_trainer.GraphedTrainStep): one graph launch per step
    from azalea_amd.policy_trainer import GraphedTrainStep
    for B in args.batch:
        net = HexNetwork(board_size=11, num_blocks=6, base_chans=64).to(dev).train()
        opt = optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
        gs = GraphedTrainStep(net, opt, B, dev)
        order = np.resize(buf.epoch_indices(), (args.steps + args.warmup) * B)
        for i in range(args.steps + args.warmup):
            if i == args.warmup:
                torch.cuda.synchronize()
                t_all = time.perf_counter()
            l3, _ = gs.step_from_ring(buf, order[i * B:(i + 1) * B])
        torch.cuda.synchronize()
        total = time.perf_counter() - t_all
        out["lines"].append({"batch": B, "mode": "hip graph fed from the ring (GraphedTrainStep.step_from_ring)", "steps_per_sec": args.steps / total,
                             "positions_per_sec": args.steps * B / total, "ms_per_step": 1e3 * total / args.steps,
                             "fresh_rows_per_sec_consumed": args.steps * B / args.oversampling / total,
                             "last_loss": float(l3[0].item())})
    E.close()
    out["note"] = ("stock PyTorch-ROCm fp32 (MIOpen / rocBLAS), one stream, loss.item() per step as the reference; "
                   "self-play on one MI355X produces ~1.2e4 rows/s (bench.py 'api' leg)")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
