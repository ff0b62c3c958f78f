"""Replays the Network.run calls recorded in golden G5r (tests/golden/make_golden.py RunRecorder): the
reference's own network outputs, row by row, handed to whatever search asks for them -- and every
request is checked against the inputs the reference's search produced at that point (flipped boards
and flipped, order-preserving legal-move lists: hex.py:72-122 through mcts.py:178-181)."""
import numpy as np


class RunTape:
    def __init__(self, z):
        self.board, self.moves, self.off = z["run_board"], z["run_moves"], z["run_off"]
        self.value, self.logprob = z["run_value"], z["run_logprob"]
        self.calls = z["run_calls"]
        self.row = 0
        self.call = 0

    def __len__(self):
        return len(self.value)

    def next_call(self, boards, legal_moves):
        """boards [B,n,n], legal_moves [B,K] of one Network.run call -> (value[B], logprob[B,K])."""
        B, K = legal_moves.shape
        assert self.call < len(self.calls) and B == int(self.calls[self.call]), (self.call, B)
        value = np.zeros(B, np.float32)
        logprob = np.full((B, K), -99.0, np.float32)
        for i in range(B):
            r = self.row + i
            a, b = int(self.off[r]), int(self.off[r + 1])
            k = b - a
            assert np.array_equal(np.asarray(boards[i], np.int8), self.board[r]), ("board", r)
            assert np.array_equal(np.asarray(legal_moves[i, :k], np.int16), self.moves[a:b]), ("moves", r)
            assert not np.asarray(legal_moves[i, k:]).any(), ("padding", r)
            value[i] = self.value[r]
            logprob[i, :k] = self.logprob[a:b]
        self.row += B
        self.call += 1
        return value, logprob

    def inputs(self, rows, n):
        """(boards i32[R,n,n], legal_moves i32[R,n*n] zero padded) of the given tape rows."""
        boards = self.board[rows].astype(np.int32)
        lm = np.zeros((len(rows), n * n), np.int32)
        for j, r in enumerate(rows):
            a, b = int(self.off[r]), int(self.off[r + 1])
            lm[j, :b - a] = self.moves[a:b]
        return boards, lm
