"""Uniform-random mover with Policy's choose_action contract (azalea/random_policy.py); seeds the
replay buffer before training (policy_trainer.py:145-158)."""
from typing import Dict, Optional

import numpy as np


class RandomPolicy:
    def __init__(self):
        self.rng = np.random.RandomState()
        self.settings: Dict = {}
        self.ply = 0
        self.seed()

    def reset(self):
        self.ply = 0

    def seed(self, seed: Optional[int] = None) -> None:
        self.rng.seed(seed)

    def state_dict(self) -> Dict:
        return {}

    def load_state_dict(self, state: Dict) -> None:
        pass

    def choose_action(self, game):
        st = game.state
        assert not st.result
        moves = st.legal_moves
        pick = self.rng.randint(len(moves))
        probs = np.full(len(moves), 1.0, np.float32) / len(moves)
        info = dict(move_id=pick, moves=moves, moves_prob=probs, prob=probs[pick], metrics={})
        return moves[pick], info

    def execute_action(self, move, moves):
        self.ply += 1

    def tree_metrics(self):
        return {}
