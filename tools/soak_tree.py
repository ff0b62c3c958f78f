import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from azalea_amd import engine as eng
E = eng.Engine(board_size=11, n_games=4096, simulations=400, search_batch_size=10, evaluator=eng.EVAL_UNIFORM)
t = time.time(); tot = 0; games = 0; errs = 0
for it in range(12):
    rows, st = E.play(250000)
    tot += len(rows["reward"]); games += st["games"]; errs += st["game_errors"]
    uid = rows["game_uid"]; b = rows["board"].reshape(len(uid), -1)
    starts = np.flatnonzero(np.r_[True, uid[1:] != uid[:-1]]); ends = np.r_[starts[1:], len(uid)]
    assert len(np.unique(uid)) == len(starts)
    k = (b == 0).sum(1)
    assert np.array_equal(k, rows["nlegal"]) and np.abs(rows["moves_prob"].sum(1) - 1).max() < 1e-5
    for s, e in zip(starts[:200], ends[:200]):
        assert (b[s] == 0).all() and ((b[s+1:e] != b[s:e-1]).sum(1) == 1).all() and rows["reward"][e-1] == 1.0
    assert st["selects"] == st["plies"] * 410
print("soak ok: positions", tot, "games", games, "errors", errs, "seconds %.1f" % (time.time() - t), "mean game length %.1f" % (tot / games))
E.close()
