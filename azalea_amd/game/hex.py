"""Host-side Hex game object with the reference's SearchableEnv surface (azalea/game/hex.py:19-134,
azalea/typing/searchable_env.py:8-38).

This is API bookkeeping for AzaleaAgent/Policy (``agent.game.state``, ``game.step``): the search
itself -- move generation, win detection, tree walks -- runs on the GPU in libazx_hip.so, where
each engine slot keeps its own copy of the position.  Win detection here is an incremental
union-find over stones (the reference flood-fills from the last move, hex.py:204-231; both report
the same winner because only the last mover's group can newly connect its two edges).
"""
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np


@dataclass
class HexGameState:
    color: int               # 0 = first player (X) to move, 1 = second (hex.py:13)
    legal_moves: np.ndarray  # int32, ascending flat index + 1; empty once the game is over
    result: int              # 0 ongoing, 1 second player won, 3 first player won
    board: np.ndarray        # int32 [N, N]: 0 empty, 1 X, 2 O


class HexGame:
    def __init__(self, board_size: int = 11) -> None:
        self.board_size = board_size
        self._game_snapshot = None
        self.reset()

    # ---- rules ---------------------------------------------------------------------------
    def reset(self) -> None:
        n = self.board_size
        self._board = np.zeros((n, n), np.int32)
        self._color = 1          # 1 = X moves first (hex.py:148)
        self._winner = 0
        self._moves = []         # move history (lets the engine slot be re-synchronised)
        # union-find over cells plus four virtual edge nodes: X-top, X-bottom, O-left, O-right
        self._parent = list(range(n * n + 4))
        self._game_snapshot = None

    def _find(self, a: int) -> int:
        p = self._parent
        while p[a] != a:
            p[a] = p[p[a]]
            a = p[a]
        return a

    def _union(self, a: int, b: int) -> None:
        ra, rb = self._find(a), self._find(b)
        if ra != rb:
            self._parent[ra] = rb

    def step(self, move: int) -> None:
        n = self.board_size
        tile = int(move) - 1
        if tile < 0 or tile >= n * n or self._board.flat[tile] != 0 or self._winner != 0:
            raise AssertionError("illegal move")
        color = self._color
        self._board.flat[tile] = color
        self._color = 3 - color
        self._moves.append(int(move))
        r, c = divmod(tile, n)
        base = n * n
        if color == 1:
            if r == 0:
                self._union(tile, base)
            if r == n - 1:
                self._union(tile, base + 1)
        else:
            if c == 0:
                self._union(tile, base + 2)
            if c == n - 1:
                self._union(tile, base + 3)
        for dr, dc in ((-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0)):   # hex.py:190-195
            rr, cc = r + dr, c + dc
            if 0 <= rr < n and 0 <= cc < n and self._board[rr, cc] == color:
                self._union(tile, rr * n + cc)
        lo, hi = (base, base + 1) if color == 1 else (base + 2, base + 3)
        if self._find(lo) == self._find(hi):
            self._winner = color

    def _legal_moves(self) -> np.ndarray:
        if self._winner:
            return np.empty(0, np.int32)
        return (np.flatnonzero(self._board.ravel() == 0) + 1).astype(np.int32)

    def _result(self) -> int:
        if self._winner:
            return 1 if self._winner == 2 else 3
        return 0

    # ---- SearchableEnv surface -----------------------------------------------------------
    def seed(self, seed: Optional[int] = None) -> None:
        pass

    @property
    def state(self) -> HexGameState:
        return HexGameState(self._color - 1, self._legal_moves(), self._result(), self._board.copy())

    @property
    def move_history(self):
        return list(self._moves)

    def __getstate__(self):
        return (self._board.copy(), self._color, self._winner, list(self._moves))

    def __setstate__(self, state):
        board, color, winner, moves = state
        self.board_size = board.shape[0]
        self.reset()
        for m in moves:
            self.step(m)
        assert np.array_equal(self._board, board) and self._color == color and self._winner == winner

    def snapshot(self) -> None:
        self._game_snapshot = self.__getstate__()

    def restore(self) -> None:
        assert self._game_snapshot
        self.__setstate__(self._game_snapshot)

    # ---- perspective helpers (hex.py:72-134) --------------------------------------------
    @staticmethod
    def flip_player_board(board: np.ndarray) -> np.ndarray:
        """Opponent's view: swap colours, mirror along the anti-diagonal."""
        board = np.asarray(board)
        if board.ndim == 2:
            return HexGame.flip_player_board(board[None])
        swapped = np.where(board > 0, 3 - board, 0)
        # out[b, i, j] = in[b, N-1-j, N-1-i]
        return np.ascontiguousarray(swapped[:, ::-1, ::-1].transpose(0, 2, 1))

    @staticmethod
    def flip_player_board_moves(board: np.ndarray, moves: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        board, moves = np.asarray(board), np.asarray(moves)
        if board.ndim == 2:
            return HexGame.flip_player_board_moves(board[None], moves[None])
        n = board.shape[-1]
        fboard = HexGame.flip_player_board(board)
        t = moves - 1
        r, c = t // n, t % n
        fmoves = np.where(moves > 0, (n - 1 - c) * n + (n - 1 - r) + 1, 0).astype(moves.dtype)
        return fboard, fmoves

    @staticmethod
    def random_reflect(board, moves=None, rng=None):
        """Identity, as in the reference snapshot (hex.py:124-134): consumes no randomness."""
        if moves is not None:
            return board, moves
        return board
