"""One game between agents with optional data collection (azalea/play_game.py:18-139)."""
import logging
import time
from collections import defaultdict
from typing import Dict, Sequence, Tuple

import numpy as np

from .replay_buffer import ReplayDataFrame


class _Wrapped:
    """Delegating wrapper: everything but choose_action goes to the wrapped agent."""

    def __init__(self, agent, after_choice):
        self._agent = agent
        self._after = after_choice

    def __getattr__(self, name):
        return getattr(self._agent, name)

    def choose_action(self) -> int:
        move = self._agent.choose_action()
        self._after(self._agent, move)
        return move


def play_game(agents: Sequence, *, game_max_length: int = 300, print_moves: bool = False,
              collect_data: bool = False) -> Tuple[int, ReplayDataFrame, Dict]:
    """May raise SearchTreeFull.  Returns (result, replay rows, per-game metrics)."""
    for a in agents:
        a.reset()
    data = ReplayDataFrame()
    metrics: Dict[str, float] = defaultdict(int)

    def collect(agent, move):                      # play_game.py:92-94: PRE-move state
        data.state.append(agent.game.state)
        data.moves_prob.append(agent.info["moves_prob"].astype(np.float32))

    def track(agent, move):                        # play_game.py:111-114
        for name, v in agent.info["metrics"].items():
            metrics[name] += v
        metrics["action_logprob"] += np.log(agent.info["prob"])

    def show(agent, move):
        color = ["white", "black"][agent.game.state.color]
        print("Move %d (%s): %s %.2f" % (agent.ply + 1, color, move, agent.info["prob"]))

    if collect_data:
        agents = [_Wrapped(a, collect) for a in agents]
    agents = [_Wrapped(a, track) for a in agents]
    if print_moves:
        agents = [_Wrapped(a, show) for a in agents]

    start = time.time()
    result, ply = 0, -1
    for ply in range(game_max_length):
        move = agents[0].choose_action()
        results = [a.execute_action(move) for a in agents]
        result = results[0]
        assert all(r == result for r in results), "conflicting game states"
        if result:
            break
        agents = agents[::-1]
    length = ply + 1
    if not result:
        logging.warning("game didn't terminate in %d moves", game_max_length)
        result = 2
    reward = np.full(length, result - 2.0, dtype=np.float32)      # play_game.py:64-65
    reward[1::2] *= -1
    if collect_data:
        data.reward = list(reward)
    for name in metrics:
        metrics[name] /= max(1, length)
    metrics.update(games=1, reward=float(reward[-1]), moves_per_game=length,
                   seconds_per_game=time.time() - start, game_error=0)
    return result, data, metrics
