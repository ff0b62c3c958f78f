"""Diagnostic soak (tools only): whole games in several engine configurations, every harvested game
checked for legality (one new stone per row, normalised move distributions, winner on the last row)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from azalea_amd import engine as eng

def check(rows, n):
    uid = rows["game_uid"]; P = len(uid)
    b = rows["board"].reshape(P, -1)
    starts = np.flatnonzero(np.r_[True, uid[1:] != uid[:-1]]); ends = np.r_[starts[1:], P]
    assert len(np.unique(uid)) == len(starts)
    assert np.array_equal((b == 0).sum(1), rows["nlegal"])
    assert np.abs(rows["moves_prob"].sum(1) - 1).max() < 1e-5
    for s, e in zip(starts[:400], ends[:400]):
        assert (b[s] == 0).all() and ((b[s + 1:e] != b[s:e - 1]).sum(1) == 1).all() and rows["reward"][e - 1] == 1.0
    return len(starts)

for name, kw, target in (
    ("11x11 uniform fast, 400 sims", dict(board_size=11, n_games=4096, simulations=400, evaluator=eng.EVAL_UNIFORM), 600000),
    ("13x13 uniform fast, 100 sims", dict(board_size=13, n_games=2048, simulations=100, evaluator=eng.EVAL_UNIFORM), 400000),
    ("7x7 uniform-hash generic, 60 sims", dict(board_size=7, n_games=512, simulations=60, evaluator=eng.EVAL_UNIFORM_HASH), 100000),
    ("11x11 tight arena (compaction), 200 sims", dict(board_size=11, n_games=512, simulations=200, evaluator=eng.EVAL_UNIFORM, nodes_per_game=60000), 100000),
):
    E = eng.Engine(search_batch_size=10, **kw)
    t = time.time(); rows, st = E.play(target)
    g = check(rows, kw["board_size"])
    print("%-42s positions %8d games %6d errors %d  %.1fs" % (name, len(rows["reward"]), g, st["game_errors"], time.time() - t))
    E.close()
print("soak_modes ok")
