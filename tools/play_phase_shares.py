"""Diagnostic (tools only; needs a library built with -DAZX_STAMP_PLAY): share of a k_play wave's cycles spent in the
search, the move draw and the game step (incl. harvest / restart / deferred-compaction flagging)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from azalea_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[1])
from azalea_amd import engine as eng
import ctypes as C
G = 4096
E = eng.Engine(board_size=11, n_games=G, simulations=400, search_batch_size=10, evaluator=eng.EVAL_UNIFORM, seed=1)
idx = np.arange(G, dtype=np.int64)
E.reset(moves=eng.random_prefixes(11, idx, 92, 7))
E.play_steps(242)
raw0 = np.zeros((G, 16), np.uint64); _lib.check(E.L.azx_debug_counters_raw(E.h, raw0.ctypes.data_as(C.POINTER(C.c_uint64)), G))
st = E.play_steps(130)
raw1 = np.zeros((G, 16), np.uint64); _lib.check(E.L.azx_debug_counters_raw(E.h, raw1.ctypes.data_as(C.POINTER(C.c_uint64)), G))
d = (raw1 - raw0)[:, 10:13].astype(np.float64)
tot = d.sum(1)
print("ms per move %.4f" % (1e3 * st["mcts_seconds"] / 130))
print("per-wave cycle shares (mean over games): search %.4f  choose %.4f  advance %.4f" % tuple((d / tot[:, None]).mean(0)))
print("cycles per move: search %.0f choose %.0f advance %.0f; advance max over games %.0f" % (d[:, 0].mean() / 130, d[:, 1].mean() / 130, d[:, 2].mean() / 130, d[:, 2].max() / 130))
