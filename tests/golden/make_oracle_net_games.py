#!/usr/bin/env python3
"""G13: the ORACLE's whole self-play games with its fp32 6x64 network at the headline's hyper-parameters.

    python tests/golden/make_oracle_net_games.py [--procs P]

TEST INFRASTRUCTURE (imports oracle/ through tests/oracle_games.py; not the reference -- runs anywhere gcc does).  The
reference side of tests/test_gpu_game_distribution.py::test_headline_kernels_at_the_headline_hyper_parameters: 2 x 256
games of 11x11 Hex, 60 -> 70 selects per move, Dirichlet alpha 0.03 / eps 0.25 / exploration_depth 15 / c_puct 0.5 / batch
10 (config/hex11_train_config.yml:19-36), leaves evaluated by `oracle.Net` on G3's seeded 6x64 weights (tests/golden/
g3_forward_11_6x64.npz: the reference's forward of these weights is held to 1e-5 by tests/test_oracle_golden.py), under
numpy's RandomState exactly as mcts.py:126-131 / policy.py:142-160 use it.  ~20 core-minutes, which is why the result is
a fixture; the columns are kept at the plies the test reads.  tests/test_oracle_golden.py replays the first games of the
fixture live, so the file cannot drift from the sampler.
"""
import argparse
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_games as og   # noqa: E402

N, SIMS, GAMES, BLOCKS, CHANS = 11, 60, 256, 6, 64
CFG = dict(batch=10, c=0.5, depth=15, alpha=0.03, eps=0.25, temp=1.0)
PLIES = list(range(0, 21)) + [24, 28, 30, 32, 36, 40, 45, 50, 55, 60, 70, 80]
SEED0 = {"a": 0, "b": 100000}
KEEP_MOVES = 4          # move lists of the first games of each half (the live replay check)


def weights_file():
    z = np.load(os.path.join(HERE, "g3_forward_11_6x64.npz"))
    state = {k[2:]: z[k] for k in z.files if k.startswith("w:") and z[k].dtype.kind == "f"}
    path = os.path.join(tempfile.mkdtemp(), "weights.npz")
    np.savez(path, **state)
    return path


def config(wpath):
    return dict(n=N, sims=SIMS, weights=wpath, blocks=BLOCKS, chans=CHANS, noise_until=None, **CFG)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=0)
    a = ap.parse_args()
    cfg = config(weights_file())
    out = {"n": N, "sims": SIMS, "games": GAMES, "cells": N * N, "plies": np.array(PLIES, np.int32)}
    out.update({"cfg_" + k: float(v) for k, v in CFG.items()})
    for half, s0 in SEED0.items():
        smp = og.sample(cfg, range(s0, s0 + GAMES), a.procs or None)
        out["length_" + half], out["first_wins_" + half] = smp["length"], smp["first_wins"]
        out["moves_" + half] = smp["moves"][:KEEP_MOVES]
        for c in og.COLUMNS:
            out[c + "_" + half] = smp[c][:, PLIES]
        print(half, "mean length %.2f, first player wins %.3f" % (smp["length"].mean(), smp["first_wins"].mean()))
    path = os.path.join(HERE, "g13_oracle_net_games_11h.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
