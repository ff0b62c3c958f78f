#!/usr/bin/env python3
"""BASELINE configs[0] on the engine: ONE 11x11 self-play game, 40 sims/move (50 select_leaf calls), 6x64 resnet,
through the reference's own call path -- play_game (play_game.py:18-78) over AzaleaAgent / Policy.choose_action
(policy.py:132-168) in PARITY mode: one engine slot, the numpy RandomState on the host (410... here 50 Dirichlet
rows per move and the multinomial move draw are the reference's numpy calls), search + network forward on the GPU.
evaluation.evaluate (sequential) and an interactive `azalea-play` run exactly this path.

Prints one JSON object: games/s, ms/move, the share of a move spent inside the engine's search call, beside the
reference's own figures for this config (0.142-0.175 games/s, 570-670 sims/s, measured in
the survey container on its CPU path -- not on this box).

    python tools/bench_single_game.py [--games 5] [--sims 40] [--board 11]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=5)
    ap.add_argument("--sims", type=int, default=40)
    ap.add_argument("--board", type=int, default=11)
    ap.add_argument("--blocks", type=int, default=6)
    ap.add_argument("--chans", type=int, default=64)
    args = ap.parse_args()
    import torch
    from azalea_amd import AzaleaAgent, HexGame, Policy
    from azalea_amd.play_game import play_game

    n = args.board
    cfg = dict(device="cuda", network="HexNetwork", board_size=n, num_blocks=args.blocks, base_chans=args.chans,
               simulations=args.sims, search_batch_size=10, exploration_coef=0.5, exploration_depth=15,
               exploration_noise_alpha=0.03, exploration_noise_scale=0.25, exploration_temperature=1.0, seed=1)
    torch.manual_seed(0)
    policy = Policy()
    policy.initialize(cfg)
    policy.settings.update(move_sampling=True, move_exploration=True)
    agent = AzaleaAgent(lambda: HexGame(n), policy=policy, device="cuda")
    agent.seed(1)
    play_game([agent], collect_data=True)                     # warm-up: engine creation, weight packing, first launches

    # time the engine's share: wrap the search call of the policy's engine
    eng = policy._engine
    acc = {"search": 0.0, "calls": 0}
    real_search = eng.search

    def timed_search(*a, **k):
        t0 = time.perf_counter()
        r = real_search(*a, **k)
        acc["search"] += time.perf_counter() - t0
        acc["calls"] += 1
        return r
    eng.search = timed_search
    plies, per_game = 0, []
    t_all = time.perf_counter()
    for g in range(args.games):
        t0 = time.perf_counter()
        _, frame, gm = play_game([agent], collect_data=True)
        per_game.append(time.perf_counter() - t0)
        plies += len(frame)
    wall = time.perf_counter() - t_all
    sel = (args.sims // 10 + 1) * 10
    out = {"config": "BASELINE configs[0] on the engine: 1 game at a time, %dx%d, %d sims/move (%d select_leaf calls), %dx%d "
                     "resnet, Policy/AzaleaAgent/play_game in parity mode (host RandomState)" % (n, n, args.sims, sel, args.blocks, args.chans),
           "games": args.games, "plies": plies, "seconds": wall,
           "games_per_sec": args.games / wall, "sims_per_sec": plies * sel / wall, "ms_per_move": 1e3 * wall / plies,
           "engine_search_ms_per_move": 1e3 * acc["search"] / max(1, acc["calls"]),
           "engine_search_share": acc["search"] / wall,
           "launches_per_move": "%d tree launches + %d network launch pairs (k_tower + k_heads) of <= 10 boards" % (sel // 10 + 2, sel // 10 + 1),
           "kernels": eng.kernel_info(),
           "reference_cpu_path": {"games_per_sec": [0.142, 0.175], "sims_per_sec": [570, 670],
                                  "what": "BASELINE.md section 2, config 1: the reference on its CPU path in the survey container "
                                          "(one 77-ply game, 5.7 s of play + 1.2-3.3 s of SearchTree allocation)"},
           "per_game_seconds": per_game}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
