#!/usr/bin/env python3
"""SURVEY 8(f).3 measured: the reference's round-robin tournament (evaluation.py:17-80; config/hex11_eval_config.yml:
three checkpoints, eval_rounds 10) on the engine, one game at a time through play_game (`evaluate`) and with every
game resident on the GPU at once (`evaluate_batched`: one engine per agent, only the slots to move are searched).
Three random-weight 6x64 networks on 11x11; same games and tallies either way (tests/test_gpu_evaluation.py).
    python tools/bench_evaluation.py [--sims 400] [--rounds 10] [--big-rounds 100]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from azalea_amd import evaluation
from azalea_amd.azalea_agent import AzaleaAgent
from azalea_amd.game.hex import HexGame
from azalea_amd.policy import Policy


def agents(sims):
    out = []
    for seed in (1, 2, 3):
        torch.manual_seed(seed)
        p = Policy()
        p.initialize(dict(device="cuda:0", network="HexNetwork", board_size=11, num_blocks=6, base_chans=64,
                          simulations=sims, search_batch_size=10, exploration_coef=0.5, exploration_depth=15,
                          exploration_noise_alpha=0.03, exploration_noise_scale=0.25, exploration_temperature=1.0))
        p.settings["move_sampling"] = True
        p.settings["move_exploration"] = False
        out.append(AzaleaAgent(lambda: HexGame(11), policy=p, device="cuda:0"))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sims", type=int, default=400)
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--big-rounds", type=int, default=100)
    args = ap.parse_args()
    ag = agents(args.sims)
    games = 3 * args.rounds
    t0 = time.perf_counter()
    seq = evaluation.evaluate(ag, args.rounds)
    t1 = time.perf_counter()
    bat = evaluation.evaluate_batched(ag, args.rounds)
    t2 = time.perf_counter()
    same = all(list(map(int, seq[p])) == list(map(int, bat[p])) for p in seq)
    big = evaluation.evaluate_batched(ag, args.big_rounds)
    t3 = time.perf_counter()
    print(json.dumps({"board": 11, "net": "6x64 random-init x3", "sims": args.sims, "rounds": args.rounds, "games": games,
                      "sequential_seconds": t1 - t0, "sequential_games_per_sec": games / (t1 - t0),
                      "batched_seconds": t2 - t1, "batched_games_per_sec": games / (t2 - t1), "same_tallies": same,
                      "big_rounds": args.big_rounds, "big_games": 3 * args.big_rounds, "big_batched_seconds": t3 - t2,
                      "big_batched_games_per_sec": 3 * args.big_rounds / (t3 - t2),
                      "tallies": {"%d-%d" % p: list(map(int, v)) for p, v in big.items()}}))


if __name__ == "__main__":
    main()
