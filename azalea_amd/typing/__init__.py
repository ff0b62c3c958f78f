"""Structural types of the agent/game seam (azalea/typing/agent.py, searchable_env.py)."""
from enum import IntEnum
from typing import Any, Mapping, Optional

import numpy as np
from typing_extensions import Protocol


class GameResult(IntEnum):
    """First player's view (typing/agent.py:34-41)."""
    ONGOING = 0
    LOSS = 1
    DRAW = 2
    WIN = 3


class GameState(Protocol):
    color: int
    legal_moves: np.ndarray
    result: int
    board: np.ndarray


class SearchableEnv(Protocol):
    def reset(self, *args, **kwargs) -> None: ...
    def seed(self, seed: Optional[int]) -> None: ...
    def step(self, action: int) -> None: ...
    @property
    def state(self) -> GameState: ...
    def snapshot(self) -> None: ...
    def restore(self) -> None: ...


class Agent(Protocol):
    def reset(self, *args, **kwargs) -> None: ...
    def seed(self, seed: Optional[int]) -> None: ...
    @property
    def settings(self) -> Mapping[str, Any]: ...
    def choose_action(self) -> int: ...
    def execute_action(self, action: int) -> GameResult: ...


__all__ = ["Agent", "SearchableEnv", "GameState", "GameResult"]
