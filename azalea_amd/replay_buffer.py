"""Replay data contract consumed by the trainer (azalea/replay_buffer.py:11-149).

ReplayDataFrame is a struct of three parallel lists (state, moves_prob, reward) that behaves as a
torch Dataset; ReplayBuffer adds the wrap-around FIFO write and the fresh-example accounting
that decides when self-play must refill it.
"""
from dataclasses import dataclass
from typing import Dict, List

import numpy as np
from torch.utils.data import Dataset

_COLUMNS = ("state", "moves_prob", "reward")


@dataclass
class ReplayRecord:
    state: object            # GameState before the move
    moves_prob: np.ndarray   # float32 [k], MCTS visit distribution
    reward: np.float32       # outcome from the mover's perspective


class ReplayDataFrame(Dataset):
    def __init__(self, state=None, moves_prob=None, reward=None):
        self.state: List[object] = [] if state is None else state
        self.moves_prob: List[np.ndarray] = [] if moves_prob is None else moves_prob
        self.reward: List[np.float32] = [] if reward is None else reward

    def __len__(self) -> int:
        return len(self.state)

    def __eq__(self, other):
        return isinstance(other, ReplayDataFrame) and all(
            getattr(self, c) == getattr(other, c) for c in _COLUMNS)

    def __getitem__(self, idx):
        cols = [getattr(self, c)[idx] for c in _COLUMNS]
        if isinstance(idx, slice):
            return ReplayDataFrame(*cols)
        if isinstance(idx, (int, np.integer)):
            return ReplayRecord(*cols)
        raise TypeError(idx)

    def __setitem__(self, idx, rows) -> None:
        before = len(self)
        for c in _COLUMNS:
            getattr(self, c)[idx] = getattr(rows, c)
        assert len(self) == before

    def append(self, rows: "ReplayDataFrame") -> None:
        for c in _COLUMNS:
            getattr(self, c).extend(getattr(rows, c))


class ReplayBuffer(ReplayDataFrame):
    """Fixed-size FIFO over a data frame (replay_buffer.py:107-149)."""

    def __init__(self, contents: ReplayDataFrame):
        super().__init__(contents.state, contents.moves_prob, contents.reward)
        self.write_idx = 0
        self.fresh_counter = 0

    def consume(self, num_examples, player) -> Dict[str, float]:
        """Account for `num_examples` used by training; refill from self-play when the stock of
        fresh examples is short (replay_buffer.py:121-132)."""
        self.fresh_counter -= num_examples
        refill = max(0, num_examples - self.fresh_counter)
        if not refill:
            return {}
        replays, metrics = player.read(refill)
        self.put(replays)
        return metrics

    def put(self, new_data: ReplayDataFrame) -> None:
        """Overwrite the oldest rows, wrapping at the end (replay_buffer.py:134-149)."""
        size, start, count = len(self), 0, len(new_data)
        while start < count:
            room = size - self.write_idx
            take = min(room, count - start)
            self[self.write_idx:self.write_idx + take] = new_data[start:start + take]
            self.write_idx = (self.write_idx + take) % size
            self.fresh_counter += take
            start += take

    def state_dict(self) -> Dict:
        return {"state": self.state, "moves_prob": self.moves_prob, "reward": self.reward,
                "write_idx": self.write_idx, "fresh_counter": self.fresh_counter}

    def load_state_dict(self, state: Dict) -> None:
        self.state, self.moves_prob, self.reward = state["state"], state["moves_prob"], state["reward"]
        self.write_idx, self.fresh_counter = state["write_idx"], state["fresh_counter"]
