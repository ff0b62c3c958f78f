"""Two-sample comparison of whole-game self-play statistics (tests/test_gpu_game_distribution.py).

A sample is the dict tests/oracle_games.py builds: `length[G]`, `first_wins[G]` and per-ply columns `[G, cells]`
(NaN beyond a game's last ply).  `compare` returns {test name: p-value}; the caller asserts every p > P_MIN.
"""
import numpy as np
from scipy import stats

P_MIN = 1e-3


def chi2_two_sample(x, y, min_expected=8.0):
    """Chi-square homogeneity test of two samples of a discrete variable; neighbouring values are pooled until
    every pooled bin expects >= min_expected in the smaller sample."""
    x, y = np.asarray(x), np.asarray(y)
    vals = np.unique(np.concatenate([x, y]))
    cx = np.array([(x == v).sum() for v in vals], np.float64)
    cy = np.array([(y == v).sum() for v in vals], np.float64)
    scale = min(len(x), len(y)) / float(len(x) + len(y))
    bx, by, ax, ay = [], [], 0.0, 0.0
    for a, b in zip(cx, cy):
        ax += a
        ay += b
        if (ax + ay) * scale >= min_expected:
            bx.append(ax)
            by.append(ay)
            ax = ay = 0.0
    if ax + ay > 0:
        if bx:
            bx[-1] += ax
            by[-1] += ay
        else:
            bx.append(ax)
            by.append(ay)
    if len(bx) < 2:
        return 1.0
    return float(stats.chi2_contingency(np.array([bx, by]))[1])


def proportion_test(kx, nx, ky, ny):
    p = (kx + ky) / float(nx + ny)
    se = np.sqrt(p * (1 - p) * (1.0 / nx + 1.0 / ny))
    if se == 0:
        return 1.0
    z = (kx / nx - ky / ny) / se
    return float(2 * stats.norm.sf(abs(z)))


def column(sample, name, ply):
    v = sample[name][:, ply]
    return v[~np.isnan(v)]


def compare(a, b, depth, plies, min_games=200):
    """p-values of: game-length histogram, first-player win rate, and per selected ply the root-child-visit
    entropy and the search value (KS), the root width, the mean child visits (carried subtree + this move's
    simulations) and the support size of the recorded moves_prob (chi-square)."""
    out = {"length": chi2_two_sample(a["length"], b["length"]),
           "first_player_wins": proportion_test(int(a["first_wins"].sum()), len(a["first_wins"]),
                                                int(b["first_wins"].sum()), len(b["first_wins"]))}
    for ply in plies:
        if min(len(column(a, "width", ply)), len(column(b, "width", ply))) < min_games:
            continue
        xa, xb = column(a, "entropy", ply), column(b, "entropy", ply)
        if min(len(xa), len(xb)) >= min_games:     # the engine's rows carry the visit distribution below the depth only
            # (on a 1e-4 grid, like action_prob below: the entropy of a visit distribution takes discrete values, the
            # engine's comes from float32 probabilities and the oracle's from the counts -- the same atom a few ulp apart)
            out["entropy@%d" % ply] = float(stats.ks_2samp(np.round(xa, 4), np.round(xb, 4)).pvalue)
        out["search_value@%d" % ply] = float(stats.ks_2samp(column(a, "search_value", ply),
                                                            column(b, "search_value", ply)).pvalue)
        out["width@%d" % ply] = chi2_two_sample(column(a, "width", ply), column(b, "width", ply))
        # mean child visits: a multiple of 1/k; compare on the visit total
        k = a["entropy"].shape[1] - ply
        out["visits@%d" % ply] = chi2_two_sample(np.round(column(a, "mean_visits", ply) * k),
                                                 np.round(column(b, "mean_visits", ply) * k))
        if ply >= depth:
            out["support@%d" % ply] = chi2_two_sample(column(a, "support", ply), column(b, "support", ply))
        else:
            # visits / total: atoms at multiples of 1/total.  The engine reports exp(f32 log p), a few ulp off the
            # oracle's f64 quotient, and a KS test reads a systematic last-digit offset on an atom as a shift of the
            # whole atom's mass -- compare on a 1e-4 grid
            out["action_prob@%d" % ply] = float(stats.ks_2samp(np.round(column(a, "action_prob", ply), 4),
                                                               np.round(column(b, "action_prob", ply), 4)).pvalue)
    return out


def worst(pvals):
    name = min(pvals, key=pvals.get)
    return name, pvals[name]
