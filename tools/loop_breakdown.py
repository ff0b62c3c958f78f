#!/usr/bin/env python3
"""Diagnostic: where the trainer's loop on one GPU (tools/bench_train_loop.py, "inline" / native) spends its wall time --
the steps between two refills (device-synchronised just before a refill), the refilling consume() call, and inside it
the engine's own play time (PlayStats.seconds).     python3 tools/loop_breakdown.py [--steps 2000] [--sims 400]"""
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

from azalea_amd import AzaleaAgent, HexGame, Player, Policy
from azalea_amd.device_replay import DeviceReplayBuffer
from azalea_amd.native_train import NativeTrainStep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--sims", type=int, default=400)
    ap.add_argument("--fill", type=int, default=60000)
    args = ap.parse_args()
    dev = "cuda:0"
    torch.manual_seed(0)
    policy = Policy()
    policy.initialize(dict(device=dev, network="HexNetwork", board_size=11, num_blocks=6, base_chans=64,
                           simulations=args.sims, search_batch_size=10, exploration_coef=0.5, exploration_depth=15,
                           exploration_noise_alpha=0.03, exploration_noise_scale=0.25, exploration_temperature=1.0, seed=1))
    policy.net.to(dev).train()
    policy.settings.update(move_sampling=True, move_exploration=True)
    agent = AzaleaAgent(lambda: HexGame(11), policy=policy, device=dev)
    player = Player(None, [agent], n_games=args.games, gather=False)
    E = player.device_engine()
    buf = DeviceReplayBuffer(E, 100000, shared=False)
    from azalea_amd import engine as eng
    player.prepare_device_engine(E)
    E.reset(moves=eng.random_prefixes(11, np.arange(args.games), 92, 1))
    buf.consume(args.fill, player)
    buf.fresh_counter = 0
    opt = optim.SGD(policy.net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    B = 128
    gs = NativeTrainStep(policy.net, opt, B, torch.device(dev))
    order = np.random.RandomState(0).randint(0, len(buf), (args.steps + 20, B))
    t_steps = t_refill = t_engine = t_prepare = 0.0
    refills = rows = 0
    for i in range(20):
        gs.step_from_ring(buf, order[i] % len(buf))
    torch.cuda.synchronize()
    t0 = mark = time.perf_counter()
    for i in range(20, args.steps + 20):
        gs.step_from_ring(buf, order[i] % len(buf))
        if buf.fresh_counter - B / 10.0 < B / 10.0:          # the consume below will refill (replay_buffer.py:121-132): close the steps' interval first
            torch.cuda.synchronize()
            a = time.perf_counter()
            t_steps += a - mark
            p0 = time.perf_counter()
            player.prepare_device_engine(E)
            torch.cuda.synchronize()
            t_prepare += time.perf_counter() - p0
            m = buf.consume(B / 10.0, player)
            torch.cuda.synchronize()
            mark = time.perf_counter()
            t_refill += mark - a
            t_engine += m["seconds_per_game"]
            refills += 1
            rows += int(m["moves_per_game"])
        else:
            buf.consume(B / 10.0, player)
    torch.cuda.synchronize()
    end = time.perf_counter()
    t_steps += end - mark
    player.stop()
    print(json.dumps({"steps": args.steps, "seconds": end - t0, "steps_per_sec": args.steps / (end - t0), "refills": refills,
                      "rows": rows, "steps_seconds": t_steps, "ms_per_step_between_refills": 1e3 * t_steps / args.steps,
                      "refill_seconds": t_refill, "ms_per_refill": 1e3 * t_refill / max(1, refills),
                      "of_which_weight_refresh_ms": 1e3 * t_prepare / max(1, refills),
                      "of_which_engine_play_ms": 1e3 * t_engine / max(1, refills)}))


if __name__ == "__main__":
    main()
