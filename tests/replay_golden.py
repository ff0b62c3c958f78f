"""Helpers shared by the replay-buffer tests: rebuild frames from golden G7 (tests/golden/g7_replay.npz)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KEYS = ("color", "legal_moves", "result", "board", "moves_prob", "reward")


def load_g7():
    return np.load(os.path.join(GOLDEN, "g7_replay.npz"))


def source_frame(z):
    """The recorded 11x11 game the reference's buffers were filled from, as a ReplayDataFrame."""
    from azalea_amd.game.hex import HexGameState
    from azalea_amd.replay_buffer import ReplayDataFrame
    f = ReplayDataFrame()
    for i in range(len(z["src_reward"])):
        lm = z["src_legal_moves"][i]
        k = int((lm > 0).sum())
        f.state.append(HexGameState(int(z["src_color"][i]), lm[:k].astype(np.int32), int(z["src_result"][i]),
                                    z["src_board"][i].astype(np.int32)))
        f.moves_prob.append(z["src_moves_prob"][i, :k].astype(np.float32))
        f.reward.append(np.float32(z["src_reward"][i]))
    return f


def widen(batch, cells):
    """collated batch (numpy dict) with legal_moves / moves_prob zero-padded to `cells` columns"""
    out = dict(batch)
    for name in ("legal_moves", "moves_prob"):
        a = np.asarray(batch[name])
        w = np.zeros((a.shape[0], cells), a.dtype)
        w[:, :a.shape[1]] = a
        out[name] = w
    return out


def assert_batch_equal(got, z, prefix, cells=121, exact_width=None):
    for k in KEYS:
        g = np.asarray(got[k])
        w = z[prefix + k]
        if exact_width is not None and k in ("legal_moves", "moves_prob"):
            assert g.shape[1] == exact_width, (k, g.shape, exact_width)
        if k in ("legal_moves", "moves_prob") and g.shape[1] != w.shape[1]:
            g = widen({"legal_moves": g, "moves_prob": g}, cells)[k]
        assert g.dtype == w.dtype, (k, g.dtype, w.dtype)
        assert np.array_equal(g, w), k
