#!/usr/bin/env python3
"""G11: whole self-play games of the REFERENCE in the configurations of tests/test_gpu_game_distribution.py.

Run in the build container only (imports the reference; it does not exist on the GPU box):

    PYTHONPATH=tools/refshim:/root/reference python tests/golden/make_game_stats.py

For the first seeds of each configuration this plays `azalea.play_game.play_game([AzaleaAgent(Policy(stub net))])`
exactly as tests/golden/make_golden.py's G5 does (uniform priors, value = fnv1a hash of the board) and records, per
game, the move list, the winner and the per-ply search metrics the reference reports.  tests/test_oracle_golden.py
holds tests/oracle_games.py (the oracle sampler the GPU distribution test compares the engine with) to these games
bit for bit, so the distribution test's reference sample IS the reference's algorithm under numpy's RandomState.
"""
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

CONFIGS = {   # tag: (n, sims, batch, c, depth, alpha, eps, temp, games)
    "7": (7, 60, 10, 0.5, 6, 0.3, 0.25, 1.0, 96),
    "11": (11, 100, 10, 0.5, 6, 0.3, 0.25, 1.0, 12),
    # the headline's own search hyper-parameters (config/hex11_train_config.yml:19-36: alpha 0.03, depth 15), where a
    # Dirichlet row over ~100 children is one or two spikes and the rest underflows
    "11h": (11, 100, 10, 0.5, 15, 0.03, 0.25, 1.0, 16),
}


def one(args):
    tag, seed = args
    import make_golden as mg      # imports azalea
    n, sims, bs, c, depth, alpha, eps, temp, _ = CONFIGS[tag]
    policy = mg.make_policy("uniformhash", n, sims, bs, c, depth, alpha, eps, temp)
    policy.settings["move_sampling"] = True        # self-play settings (policy_trainer.py:68-69)
    policy.settings["move_exploration"] = True
    agent = mg.AzaleaAgent(lambda n=n: mg.HexGame(n), policy=policy, device="cpu")
    agent.seed(seed)
    per_ply = []
    choose = policy.choose_action

    def recording(game):
        move, info = choose(game)
        m = info["metrics"]
        per_ply.append((move, m["search_root_width"], m["search_root_visits"], m["search_value"],
                        int((info["moves_prob"] > 0).sum()), info["prob"]))
        return move, info
    policy.choose_action = recording
    result, frame, metrics = mg.play_game([agent], collect_data=True)
    cells = n * n
    moves = np.zeros(cells, np.int16)
    width, support = np.zeros(cells, np.int16), np.zeros(cells, np.int16)
    visits, sval, aprob = (np.full(cells, np.nan, np.float32) for _ in range(3))
    for i, (mv, w, v, sv, sup, ap) in enumerate(per_ply):
        moves[i], width[i], visits[i], sval[i], support[i], aprob[i] = mv, w, np.float32(v), np.float32(sv), sup, np.float32(ap)
    return dict(length=len(per_ply), first_wins=int(result == 3), moves=moves, width=width, mean_visits=visits,
                search_value=sval, support=support, action_prob=aprob)


def main():
    out = {}
    with mp.get_context("fork").Pool(max(1, (os.cpu_count() or 2) - 1)) as pool:
        for tag, cfg in CONFIGS.items():
            games = pool.map(one, [(tag, s) for s in range(cfg[-1])])
            out["cfg_" + tag] = np.array(cfg, np.float64)
            out["length_" + tag] = np.array([g["length"] for g in games], np.int16)
            out["first_wins_" + tag] = np.array([g["first_wins"] for g in games], np.int8)
            for k in ("moves", "width", "mean_visits", "search_value", "support", "action_prob"):
                out[k + "_" + tag] = np.stack([g[k] for g in games])
            print(tag, "games", len(games), "mean length", out["length_" + tag].mean())
    path = os.path.join(HERE, "g11_game_summaries.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
