"""Elo scores from tournament tallies (Bradley-Terry maximum likelihood) -- what the reference feeds `evaluate`'s
outcome dict into (ranking.py:46-58; compare_cli.py:76).

    P(i beats j) = 1 / (1 + 10 ** ((elo_j - elo_i) / 400)),   player 0 pinned at 0 Elo (ranking.py:13-14, 37-38).

The negative log-likelihood and its gradient are written out (the reference builds them with autograd, pair by pair):
with s = elo * ln10 / 400 and d = s_i - s_j,  -log P(i beats j) = softplus(-d), and d/dd of a pair's term is
losses * sigmoid(d) - wins * sigmoid(-d).  Minimised with L-BFGS-B from zero, as the reference does.
"""
from typing import Dict, Sequence, Tuple

import numpy as np
from scipy.optimize import minimize

_K = np.log(10.0) / 400.0


class RankingError(Exception):
    """The optimiser did not converge (ranking.py:42-43, 57-58)."""


def _tables(num_players: int, outcomes: Dict[Tuple[int, int], Sequence[int]]):
    i = np.array([p[0] for p in outcomes], np.int64)
    j = np.array([p[1] for p in outcomes], np.int64)
    t = np.array([[int(x) for x in outcomes[p]] for p in outcomes], np.float64).reshape(-1, 3)
    if len(t) and t[:, 1].any():
        raise AssertionError("draws not supported")          # ranking.py:31
    if len(i) and (min(i.min(), j.min()) < 0 or max(i.max(), j.max()) >= num_players):
        raise IndexError("a pair names a player outside 0..%d" % (num_players - 1))
    return i, j, t[:, 0], t[:, 2]


def neg_log_likelihood(elo: np.ndarray, i, j, wins, losses):
    """(loss, d loss / d elo) with player 0 held at 0."""
    elo = np.array(elo, np.float64)
    elo[0] = 0.0
    s = elo * _K
    d = s[i] - s[j]
    loss = float(np.sum(wins * np.logaddexp(0.0, -d) + losses * np.logaddexp(0.0, d)))
    sig = 0.5 * (1.0 + np.tanh(0.5 * d))                      # sigmoid(d), stable at both ends
    gd = losses * sig - wins * (1.0 - sig)
    g = np.zeros_like(s)
    np.add.at(g, i, gd)
    np.add.at(g, j, -gd)
    g[0] = 0.0
    # the reference hands scipy the gradient with respect to s (natural-log units), not elo (ranking.py:17-18, 33-38):
    # kept, because the optimiser's steps -- and therefore where its tolerance stops it -- depend on that scale
    return loss, g


def compute_ranking(num_players: int, outcomes: Dict[Tuple[int, int], Sequence[int]]) -> np.ndarray:
    """Elo per player from {(p1, p2): (p1 wins, draws, p1 losses)} -- `evaluate` / `evaluate_batched`'s result."""
    i, j, wins, losses = _tables(num_players, outcomes)
    res = minimize(neg_log_likelihood, np.zeros(num_players), args=(i, j, wins, losses), method="L-BFGS-B", jac=True)
    if not res.success:
        raise RankingError("did not converge")
    return res.x


def expected_score(elo_a: float, elo_b: float) -> float:
    """P(a beats b) under the model."""
    return float(1.0 / (1.0 + 10.0 ** ((elo_b - elo_a) / 400.0)))
