"""Elo from tournament tallies (`evaluate`'s outcomes feed ranking.py:46-58) against the reference's own
scores on five tournaments (golden G12, tests/golden/make_ranking.py)."""
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_elo_matches_the_reference_g12():
    from azalea_amd import ranking
    cases = json.load(open(os.path.join(GOLDEN, "g12_ranking.json")))
    assert len(cases) == 5
    for c in cases:
        outcomes = {tuple(k): tuple(v) for k, v in c["outcomes"]}
        elo = ranking.compute_ranking(c["players"], outcomes)
        assert elo[0] == 0.0
        np.testing.assert_allclose(elo, c["elo"], atol=1e-3)        # Elo points


def test_elo_model_and_errors():
    from azalea_amd import ranking
    elo = ranking.compute_ranking(2, {(0, 1): (63, 0, 137)})
    assert abs(ranking.expected_score(elo[1], elo[0]) - 137 / 200) < 1e-5      # one pair: the MLE reproduces the tally
    with pytest.raises(AssertionError):
        ranking.compute_ranking(2, {(0, 1): (3, 1, 2)})                         # ranking.py:31
    with pytest.raises(IndexError):
        ranking.compute_ranking(2, {(0, 2): (3, 0, 2)})
    # the gradient handed to the optimiser is the derivative of the loss it is handed
    i, j, w, l = ranking._tables(4, {(0, 1): (3, 0, 7), (1, 2): (6, 0, 4), (3, 0): (5, 0, 5), (2, 3): (1, 0, 9)})
    x = np.array([0.0, 30.0, -80.0, 120.0])
    _, g = ranking.neg_log_likelihood(x, i, j, w, l)
    for k in range(1, 4):
        e = np.zeros(4)
        e[k] = 1e-3
        num = (ranking.neg_log_likelihood(x + e, i, j, w, l)[0] - ranking.neg_log_likelihood(x - e, i, j, w, l)[0]) / 2e-3
        assert abs(num / ranking._K - g[k]) < 1e-6 * max(1.0, abs(g[k]))
