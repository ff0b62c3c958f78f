"""Round-robin tournament between policies (azalea/evaluation.py:17-80), SURVEY 8(f).3.

`evaluate` keeps the reference's schedule: pairs (i, j), i < j in `gen_pairs` order, one game per
pair and round, task seed `10000 * round + index`, first player by a coin flip of
`RandomState(seed)`, both agents re-seeded from the same stream, outcome = order * (result - 2)
(+1: the pair's first agent won).  Games run in this process, one after the other, through
`play_game` -- every agent's search runs on its own engine.

`evaluate_batched` plays all games of a round concurrently on the GPU when every agent is a
`Policy` with a device network: per agent one engine holds a slot for each of its games, a ply
searches only the slots whose turn it is (`azx_set_active`), and every game keeps the private
RandomStates the sequential schedule would give it, so the outcomes are the same games, move for
move.
"""
import logging
from collections import defaultdict
from typing import Dict, List, Tuple

import numpy as np

from .play_game import play_game

Pair = Tuple[int, int]
OutcomeCounts = List[int]


def gen_pairs(num_players: int) -> List[Pair]:
    """Round-robin pair ordering (evaluation.py:39-44)."""
    return [(i, j) for j in range(num_players) for i in range(j)]


def _tally(outcomes, pair, res, game, num_games):
    outcomes[pair][0] += (res > 0)    # first player of the pair wins
    outcomes[pair][1] += (res == 0)   # draws
    outcomes[pair][2] += (res < 0)    # second player wins
    winrate = outcomes[pair][0] / sum(outcomes[pair])
    logging.info("game %d/%d: pair %s: outcomes %s (wins %.2f)", game, num_games, pair, outcomes[pair], winrate)


def worker(pair: Pair, agents, seed: int) -> Tuple[Pair, int]:
    """One game of a pair (evaluation.py:60-80)."""
    rng = np.random.RandomState(seed)
    order = rng.choice([-1, 1])
    for a in agents:
        a.seed(rng.randint(1 << 32))
    result, _, _ = play_game(agents[::order])
    return pair, order * (result - 2)


def evaluate(agents, num_rounds: int, num_workers=None) -> Dict[Pair, OutcomeCounts]:
    """Round robin, `num_rounds` games per pair (evaluation.py:17-36).  `num_workers` is accepted
    for signature compatibility; there is no process pool."""
    outcomes: Dict[Pair, OutcomeCounts] = defaultdict(lambda: [0, 0, 0])
    pairs = gen_pairs(len(agents))
    num_games = num_rounds * len(pairs)
    game = 1
    for r in range(num_rounds):
        for s, pair in enumerate(pairs):
            _, res = worker(pair, (agents[pair[0]], agents[pair[1]]), 10000 * r + s)
            _tally(outcomes, pair, res, game, num_games)
            game += 1
    return outcomes
