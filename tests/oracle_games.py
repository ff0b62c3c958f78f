"""Whole self-play games by the CPU oracle under numpy's RandomState, many seeds, summarised per ply.

TEST INFRASTRUCTURE (imports oracle/): the reference side of tests/test_gpu_game_distribution.py.  Every game is
`oracle.play_game` -- play_game.py:44-67 over Policy.choose_action (policy.py:132-168): `rng.dirichlet` once per
select_leaf (mcts.py:126-131), `rng.multinomial(1, probs)` for the move (policy.py:160), the temperature gated by
`exploration_depth` and the noise not (policy.py:142-149) -- held bit for bit to the reference's recorded games by
tests/test_oracle_golden.py (G5) and, for exactly the configurations used here, by the G11 fixture
(tests/golden/g11_game_summaries.npz, written by tests/golden/make_game_stats.py from the reference itself).

Run as a script in its own process (the GPU tests start it with subprocess so that the fork pool below never
inherits an initialised HIP runtime):

    python tests/oracle_games.py --n 7 --sims 60 --games 4096 --seed0 0 --out /tmp/a.npz
"""
import argparse
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# per-ply columns every sampler (oracle here, engine in the GPU test) fills, [games, cells], NaN beyond the game
COLUMNS = ("entropy", "width", "mean_visits", "search_value", "support", "action_prob")


def prior_table(n):
    """f32(1/k): the uniform prior the reference's stub network hands a position with k legal moves."""
    return (np.float32(1.0) / np.arange(0, n * n + 1).clip(1).astype(np.float32)).astype(np.float32)


def entropy(p):
    p = np.asarray(p, np.float64)
    p = p[p > 0]
    return float(-(p * np.log(p)).sum())


def summarise_oracle_game(n, result, rows):
    cells = n * n
    out = {c: np.full(cells, np.nan, np.float32) for c in COLUMNS}
    moves = np.zeros(cells, np.int16)
    for i, r in enumerate(rows):
        nv = r["child_visits"].astype(np.float64)
        out["entropy"][i] = entropy(nv / nv.sum())                      # root-child-visit entropy
        out["width"][i] = (nv > 0).sum()                                # search_tree.py:109
        out["mean_visits"][i] = np.float32(r["child_visits"].mean())    # search_tree.py:110
        out["search_value"][i] = r["search_value"]                      # mcts.py:291
        mp_ = r["moves_prob"]
        out["support"][i] = (mp_ > 0).sum()
        out["action_prob"][i] = mp_[r["move_id"]]
        moves[i] = r["move"]
    return dict(length=len(rows), first_wins=int(result == 3), moves=moves, **out)


_NETS = {}


def _evaluator(cfg):
    """The stub network of the reference's golden games (uniform priors, board-hash value) or, with cfg["weights"], the
    oracle's fp32 HexNetwork on a saved state_dict (blocked AVX-512 convolutions: held to the parity forward at 1e-5)."""
    import ctypes as C
    from oracle import oracle as orc
    n = cfg["n"]
    if not cfg.get("weights"):
        return orc.UniformEval(hash_value=True, prior_by_k=prior_table(n))
    key = cfg["weights"]
    if key not in _NETS:                      # one per worker process
        z = np.load(key)
        net = orc.Net(n, int(cfg["blocks"]), int(cfg["chans"]), {k: z[k] for k in z.files})
        net.fn = C.cast(orc.lib().oeval_net_fast, C.c_void_p)
        _NETS[key] = net
    return _NETS[key]


def _one(args):
    cfg, seed = args
    from oracle import oracle as orc
    n = cfg["n"]
    ev = _evaluator(cfg)
    result, rows, _ = orc.play_game(
        n, ev, simulations=cfg["sims"], batch_size=cfg["batch"], c_puct=cfg["c"],
        exploration_depth=cfg["depth"], noise_alpha=cfg["alpha"], noise_scale=cfg["eps"],
        temperature=cfg["temp"], seed=seed, tree_cap=cfg.get("tree_cap", 1 << 22),
        noise_until=cfg.get("noise_until"))
    return summarise_oracle_game(n, result, rows)


def stack(games):
    out = dict(length=np.array([g["length"] for g in games], np.int32),
               first_wins=np.array([g["first_wins"] for g in games], np.int8))
    for c in COLUMNS + ("moves",):
        out[c] = np.stack([g[c] for g in games])
    return out


def sample(cfg, seeds, procs=None):
    procs = procs or max(1, min(32, (os.cpu_count() or 2) - 1))
    work = [(cfg, int(s)) for s in seeds]
    if procs == 1:
        return stack([_one(w) for w in work])
    with mp.get_context("fork").Pool(procs) as pool:
        return stack(pool.map(_one, work, chunksize=max(1, len(work) // (procs * 8))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, required=True)
    ap.add_argument("--sims", type=int, required=True)
    ap.add_argument("--batch", type=int, default=10)
    ap.add_argument("--c", type=float, default=0.5)
    ap.add_argument("--depth", type=int, default=6)
    ap.add_argument("--alpha", type=float, default=0.3)
    ap.add_argument("--eps", type=float, default=0.25)
    ap.add_argument("--temp", type=float, default=1.0)
    ap.add_argument("--noise-until", type=int, default=None)
    ap.add_argument("--games", type=int, required=True)
    ap.add_argument("--seed0", type=int, default=0)
    ap.add_argument("--procs", type=int, default=0)
    ap.add_argument("--out", required=True)
    ap.add_argument("--weights", default=None, help="npz of a HexNetwork state_dict: play with the oracle's network")
    ap.add_argument("--blocks", type=int, default=0)
    ap.add_argument("--chans", type=int, default=0)
    a = ap.parse_args()
    cfg = dict(n=a.n, sims=a.sims, batch=a.batch, c=a.c, depth=a.depth, alpha=a.alpha, eps=a.eps, temp=a.temp,
               noise_until=a.noise_until, weights=a.weights, blocks=a.blocks, chans=a.chans)
    res = sample(cfg, range(a.seed0, a.seed0 + a.games), a.procs or None)
    np.savez_compressed(a.out, **res)


if __name__ == "__main__":
    main()
