"""Diagnostic: spread of the per-game search time summed over the moves of one persistent k_play launch
(-DAZX_STAMP=2 build): how far the slowest game lags the average, i.e. the launch's idle tail."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from azalea_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libazx_stamp2.so")
from azalea_amd import engine as eng
E = eng.Engine(board_size=11, n_games=4096, simulations=400, search_batch_size=10, evaluator=eng.EVAL_UNIFORM, noise_scale=0.25)
E.play_steps(20)
a = E.debug_counters_raw()[:, 13].astype(np.float64)
st = E.play_steps(130)
b = E.debug_counters_raw()[:, 13].astype(np.float64)
d = b - a
print("launch ms %.2f; per-game summed search ticks: mean %.4g  min %.4g  max %.4g  max/mean %.3f  p99/mean %.3f" % (
    1e3 * st["mcts_seconds"], d.mean(), d.min(), d.max(), d.max() / d.mean(), np.percentile(d, 99) / d.mean()))
