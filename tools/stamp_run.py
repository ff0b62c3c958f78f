"""Diagnostic: per-region cycle shares of k_mcts from a -DAZX_STAMP build (tools only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from azalea_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libazx_stamp.so")
from azalea_amd import engine as eng
ns = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
E = eng.Engine(board_size=11, n_games=G, simulations=400, search_batch_size=10, evaluator=eng.EVAL_UNIFORM, noise_scale=ns)
E.play_steps(20)
a = E.debug_counters().astype(np.float64)
st = E.play_steps(40)
b = E.debug_counters().astype(np.float64)
d = b - a
names = ["root level score", "deeper level load+score", "hex step", "leaf record + VL", "undo + dedup", "eval/expand/backup"]
tot = d[10:16].sum()
print("selects", d[0], "ms/launch", 1e3 * st["mcts_seconds"] / st["mcts_launches"])
for n, v in zip(names, d[10:16]):
    print("%-26s %6.1f%%  %8.0f cycles/sim" % (n, 100 * v / tot, v / d[0]))
print("stamped cycles/sim", tot / d[0])
