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


# ---------------------------------------------------------------------------------------------
# all games of the tournament at once, on the GPU
# ---------------------------------------------------------------------------------------------
def _device_policy(agent):
    from .policy import Policy
    pol = getattr(agent, "policy", None)
    if not isinstance(pol, Policy) or not pol._uses_device_net():
        raise TypeError("evaluate_batched needs agents whose Policy holds a HexNetwork")
    return pol


def evaluate_batched(agents, num_rounds: int, *, game_max_length: int = 300) -> Dict[Pair, OutcomeCounts]:
    """Same tournament as `evaluate` (same seeds, coin flips, per-game RandomStates, so the same
    games and tallies), with every game resident on the GPU at once: agent a's engine holds one
    slot per game a plays in and searches, each ply, the slots whose turn it is."""
    import torch
    from . import engine as _eng
    from .policy import SearchTreeFull, as_distribution

    pols = [_device_policy(a) for a in agents]
    n = agents[0].game.board_size
    pairs = gen_pairs(len(agents))
    # ---- the schedule of evaluation.py:24-57, with the worker's RNG draws (evaluation.py:69-75) ----
    games = []
    for r in range(num_rounds):
        for s, pair in enumerate(pairs):
            rng = np.random.RandomState(10000 * r + s)
            order = rng.choice([-1, 1])
            rngs = {}
            for a in pair:
                g_rng = np.random.RandomState()
                g_rng.seed(rng.randint(1 << 32) + 1)          # AzaleaAgent.seed -> policy.seed(s + 1)
                rngs[a] = g_rng
            first, second = (pair if order == 1 else pair[::-1])
            games.append(dict(pair=pair, order=order, rngs=rngs, players=(first, second),
                              game=agents[0].game.__class__(n), result=0, ply=0, slot={}))
    # ---- one engine per agent, one slot per game it plays in ----
    engines = []
    for a, pol in enumerate(pols):
        mine = [g for g in games if a in g["pair"]]
        for i, g in enumerate(mine):
            g["slot"][a] = i
        dev = pol.net.device
        eng = _eng.Engine(board_size=n, n_games=max(1, len(mine)), simulations=pol.simulations,
                          search_batch_size=pol.search_batch_size, exploration_coef=pol.exploration_coef,
                          exploration_depth=pol.exploration_depth, noise_alpha=pol.exploration_noise_alpha,
                          noise_scale=pol.exploration_noise_scale, temperature=pol.exploration_temperature,
                          evaluator=_eng.EVAL_RESNET, num_blocks=pol.num_blocks, base_chans=pol.base_chans,
                          device=(dev.index or 0) if dev.type == "cuda" else 0)
        pol.net.eval()
        sd = {k: v for k, v in pol.net.state_dict().items() if v.dtype == torch.float32}
        if dev.type == "cuda":
            eng.set_weights({k: (v.contiguous().data_ptr(), v.numel()) for k, v in sd.items()}, on_device=True)
        else:
            eng.set_weights({k: v.detach().cpu().numpy() for k, v in sd.items()})
        engines.append((eng, mine))
    try:
        for ply in range(game_max_length):
            alive = [g for g in games if not g["result"]]
            if not alive:
                break
            chosen = {}                                    # id(game) -> (move, move_id)
            for a, (eng, mine) in enumerate(engines):
                todo = [g for g in mine if not g["result"] and g["players"][g["ply"] % 2] == a]
                if not todo:
                    continue
                pol = pols[a]
                # Policy.choose_action's schedule (policy.py:132-149)
                temperature = noise_scale = 0.0
                if pol.settings["move_sampling"]:
                    temperature = pol.exploration_temperature
                    if pol.settings["move_exploration"]:
                        noise_scale = pol.exploration_noise_scale
                mask = np.zeros(eng.G, np.int32)
                noise = None
                if noise_scale:
                    noise = np.zeros((eng.G, eng.selects_per_search, n * n), np.float64)
                for g in todo:
                    slot = g["slot"][a]
                    mask[slot] = 1
                    if noise_scale:
                        k = len(g["game"].state.legal_moves)
                        alpha = np.full(k, pol.exploration_noise_alpha)
                        for j in range(eng.selects_per_search):
                            noise[slot, j, :k] = g["rngs"][a].dirichlet(alpha)
                eng.set_active(mask)
                eng.search(noise=noise, noise_scale=noise_scale)
                status = eng.get_status()
                root = eng.get_root()
                for g in todo:
                    slot = g["slot"][a]
                    if status[slot]:
                        raise SearchTreeFull("too many nodes")
                    legal = g["game"].state.legal_moves
                    k = len(legal)
                    assert int(root["k"][slot]) == k
                    t = 0.0 if g["ply"] >= pol.exploration_depth else temperature
                    probs = as_distribution(root["child_visits"][slot, :k], t)
                    move_id = int(np.argmax(g["rngs"][a].multinomial(1, probs)))
                    chosen[id(g)] = (int(legal[move_id]), move_id)
            # every agent follows every move of its games (Policy.execute_action, policy.py:170-176)
            for a, (eng, mine) in enumerate(engines):
                ids = np.full(eng.G, -1, np.int32)
                mask = np.zeros(eng.G, np.int32)
                for g in mine:
                    if not g["result"]:
                        ids[g["slot"][a]] = chosen[id(g)][1]
                        mask[g["slot"][a]] = 1
                if mask.any():
                    eng.set_active(mask)
                    eng.advance(ids)
            for g in alive:
                g["game"].step(chosen[id(g)][0])
                g["ply"] += 1
                g["result"] = int(g["game"].state.result)
        outcomes: Dict[Pair, OutcomeCounts] = defaultdict(lambda: [0, 0, 0])
        for i, g in enumerate(games):
            result = g["result"]
            if not result:
                logging.warning("game didn't terminate in %d moves", game_max_length)
                result = 2
            _tally(outcomes, g["pair"], g["order"] * (result - 2), i + 1, len(games))
        return outcomes
    finally:
        for eng, _ in engines:
            eng.close()
