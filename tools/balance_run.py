"""Diagnostic: per-wave lifetime of one k_mcts launch and its spread over SIMDs (-DAZX_STAMP=2 build)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from azalea_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libazx_stamp2.so")
from azalea_amd import engine as eng
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
E = eng.Engine(board_size=11, n_games=G, simulations=400, search_batch_size=10, evaluator=eng.EVAL_UNIFORM, noise_scale=0.25)
E.play_steps(int(sys.argv[2]) if len(sys.argv) > 2 else 60)
st = E.play_steps(1)
raw = E.debug_counters_raw()
life = raw[:, 10].astype(np.float64); hw = raw[:, 11]; t0 = raw[:, 12].astype(np.float64)
span = (t0 + life).max() - t0.min()
print("launch ms %.3f  span(memtime ticks) %.0f  wave life mean %.0f min %.0f max %.0f  mean/span %.3f" % (
    1e3 * st["mcts_seconds"], span, life.mean(), life.min(), life.max(), life.mean() / span))
print("start spread ticks", t0.max() - t0.min())
hwid = (hw & 0xffffffff).astype(np.int64); xcc = (hw >> 32).astype(np.int64) & 0xf
simd = (hwid >> 4) & 3; cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
u, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
sums = np.bincount(inv, weights=life)
print("SIMDs used", len(u), "waves/SIMD min/mean/max", cnt.min(), cnt.mean(), cnt.max())
print("per-SIMD sum of lifetimes: mean %.0f max %.0f  max/mean %.3f" % (sums.mean(), sums.max(), sums.max() / sums.mean()))
print("per-SIMD max lifetime: mean %.0f" % np.array([life[inv == i].max() for i in range(len(u))]).mean())
order = np.argsort(t0)
print("blockIdx of first 16 started:", order[:16])
print("key of blockIdx 0..15:", key[:16])
print("lifetime quantiles", np.percentile(life, [1, 10, 50, 90, 99]))
gm = E.get_games() if hasattr(E, "get_games") else None
if gm is not None:
    ply = np.asarray(gm["ply"] if isinstance(gm, dict) else gm[2]).astype(np.float64)
    print("corr(life, ply) %.3f" % np.corrcoef(life, ply)[0, 1])
    for lo, hi in ((0, 10), (10, 30), (30, 50), (50, 70), (70, 121)):
        m = (ply >= lo) & (ply < hi)
        if m.any(): print("ply %3d-%3d: n %5d mean life %.0f" % (lo, hi, m.sum(), life[m].mean()))
