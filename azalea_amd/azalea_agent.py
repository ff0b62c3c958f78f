"""AzaleaAgent: one game object + one policy (azalea/azalea_agent.py:11-64)."""
from typing import Callable, Dict, Optional

import torch

from .policy import Policy
from .random_policy import RandomPolicy


class AzaleaAgent:
    def __init__(self, game_factory: Callable, *, path: str = None, policy=None, device=None):
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        if path is not None and policy is not None:
            raise ValueError("cannot give both path and policy")
        if path is not None:
            policy = Policy.load(path, device=device)
        elif policy is None:
            policy = RandomPolicy()          # no policy => random mover (azalea_agent.py:25-26)
        self.game = game_factory()
        self.policy = policy
        self.info = None
        self.seed()

    def reset(self) -> None:
        self.game.reset()
        self.policy.reset()
        self.info = None

    def seed(self, seed: Optional[int] = None) -> None:
        """game gets `seed`, policy `seed + 1` (azalea_agent.py:41-44)."""
        self.game.seed(seed)
        self.policy.seed(None if seed is None else seed + 1)

    @property
    def ply(self) -> int:
        return self.policy.ply

    @property
    def settings(self) -> Dict:
        return self.policy.settings

    def choose_action(self) -> int:
        move, self.info = self.policy.choose_action(self.game)
        return move

    def execute_action(self, move: int) -> int:
        self.policy.execute_action(move, self.game.state.legal_moves)
        self.game.step(move)
        return self.game.state.result
