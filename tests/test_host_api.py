"""Host-side mirror of the reference interface (CPU only): game object, replay data contract,
collate, FIFO buffer, C-ABI library loads and exports every declared symbol."""
import os
import re
import zlib

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_declared_symbol():
    """include/azx.h is the boundary: every function it declares must resolve in the .so and be
    bound in azalea_amd/_lib.py (no compute calls here -- there is no GPU)."""
    from azalea_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "azx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(azx_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"azx_engine", "azx_config", "azx_play_stats"}
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.azx_version() >= 1


def test_ctypes_structs_match_the_header(tmp_path):
    """The two structs that cross the C ABI by pointer (azx_config in, azx_play_stats out) must have
    the layout include/azx.h gives them: a C program compiled against the header prints size and
    field offsets, the ctypes mirrors in azalea_amd/_lib.py must agree field by field."""
    import ctypes as C
    import subprocess
    from azalea_amd import _lib
    pairs = (("azx_config", _lib.Config), ("azx_play_stats", _lib.PlayStats))
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "azx.h"', 'int main(void) {']
    for cname, cls in pairs:
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ['return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for cname, cls in pairs:
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got["%s.%s" % (cname, fname)]) == getattr(cls, fname).offset, (cname, fname)


def test_engine_fails_loudly_without_gpu():
    from azalea_amd import _lib, engine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.AzxError):
        engine.Engine(board_size=5)


@pytest.mark.parametrize("n", [3, 5, 11, 13])
def test_host_hexgame_vs_golden(n):
    from azalea_amd import HexGame
    z = np.load(os.path.join(GOLDEN, "g1_movegen.npz"))
    mv, res, ln = z["moves_%d" % n], z["result_%d" % n], z["length_%d" % n]
    crc, fin = z["legalcrc_%d" % n], z["final_%d" % n]
    for g in range(min(80, len(mv))):
        h = HexGame(n)
        for p in range(ln[g]):
            st = h.state
            assert st.color == p % 2 and st.legal_moves.dtype == np.int32
            assert zlib.crc32(st.legal_moves.tobytes()) == crc[g, p]
            h.step(mv[g, p])
            assert h.state.result == res[g, p]
        assert np.array_equal(h.state.board, fin[g]) and len(h.state.legal_moves) == 0
        with pytest.raises(AssertionError):
            h.step(1)


def test_host_hexgame_snapshot_restore_and_flip():
    from azalea_amd import HexGame
    g = HexGame(5)
    for m in (3, 7, 12):
        g.step(m)
    g.snapshot()
    before = g.state
    g.step(20)
    g.restore()
    assert np.array_equal(g.state.board, before.board) and g.state.color == before.color
    z = np.load(os.path.join(GOLDEN, "g2_flip.npz"))
    for n in (5, 11, 13):
        fb, fm = HexGame.flip_player_board_moves(z["board_%d" % n], z["moves_%d" % n])
        assert np.array_equal(fb, z["fboard_%d" % n]) and np.array_equal(fm, z["fmoves_%d" % n])
    b, m = np.zeros((3, 3), np.int32), np.array([1, 2], np.int32)
    assert HexGame.random_reflect(b, m)[0] is b          # identity, hex.py:124-134


def _frame_from_g5(tag):
    from azalea_amd.game.hex import HexGameState
    from azalea_amd.replay_buffer import ReplayDataFrame
    z = np.load(os.path.join(GOLDEN, "g5_game_%s.npz" % tag))
    f = ReplayDataFrame()
    for i in range(len(z["board"])):
        k = int(z["nlegal"][i])
        f.state.append(HexGameState(int(z["color"][i]), z["legal_moves"][i, :k].astype(np.int32), 0,
                                    z["board"][i].astype(np.int32)))
        f.moves_prob.append(z["moves_prob"][i, :k].astype(np.float32))
        f.reward.append(np.float32(z["reward"][i]))
    return f


def test_g6_collate_matches_reference():
    """prep.torch_batch_replays output of the reference on the same rows (keys, order, dtypes)."""
    from azalea_amd.prep import torch_batch_replays
    z = np.load(os.path.join(GOLDEN, "g6_collate.npz"))
    frame = _frame_from_g5("a_7_s20")
    tb = torch_batch_replays([frame[int(i)] for i in z["idx"]])
    want = [k[4:] for k in z.files if k.startswith("out_")]
    assert list(tb.keys()) == want
    for k in want:
        got = tb[k].numpy()
        assert str(got.dtype) == str(z["dtype_" + k]), k
        assert np.array_equal(got, z["out_" + k]), k


def test_replay_frame_and_fifo_buffer():
    from azalea_amd.replay_buffer import ReplayBuffer, ReplayDataFrame, ReplayRecord
    frame = _frame_from_g5("c_5_s30_greedy")
    n = len(frame)
    assert isinstance(frame[0], ReplayRecord) and isinstance(frame[2:5], ReplayDataFrame)
    assert len(frame[2:5]) == 3
    buf = ReplayBuffer(frame[:10])

    class FakePlayer:
        def __init__(self):
            self.asked = []

        def read(self, size):
            self.asked.append(size)
            return frame[10:10 + 7], {"games": 1}
    pl = FakePlayer()
    assert buf.consume(4, pl) == {"games": 1}      # fresh_counter -4 -> refill 8 (replay_buffer.py:125-126)
    assert pl.asked == [8] and buf.write_idx == 7 and buf.fresh_counter == 3
    assert buf.consume(1, pl) == {} and buf.fresh_counter == 2
    buf.put(frame[10:16])                            # wraps: 3 at the end, 3 at the start
    assert buf.write_idx == 3 and len(buf) == 10
    assert buf.state[9] is frame.state[12] and buf.state[0] is frame.state[13]
    with pytest.raises(TypeError):
        frame["x"]
    assert n == len(frame)


def test_pad_like_reference_tests():
    """test/test_prep.py of the reference: zero padding and dtype for ragged rows."""
    from azalea_amd.prep import pad
    assert pad([1, 2, 3]).tolist() == [1, 2, 3]
    x = pad([np.array([1, 2], np.int32), np.array([3], np.int32)])
    assert x.tolist() == [[1, 2], [3, 0]] and x.dtype == np.int32
    y = pad([np.array([1.5], np.float32), np.array([2.5, 3.5], np.float32)])
    assert y.dtype == np.float32 and y.tolist() == [[1.5, 0.0], [2.5, 3.5]]
    z = pad([np.ones((2, 3), np.float32), np.ones((1, 4), np.float32)])
    assert z.shape == (2, 2, 4) and z[1, 1].sum() == 0
    w = pad([np.ones((1, 2), np.int32)], size=(3, 3))
    assert w.shape == (1, 3, 3)


def test_as_distribution_matches_golden_probs():
    from azalea_amd.policy import as_distribution
    z = np.load(os.path.join(GOLDEN, "g4_search_c_11_p30_s400_hp.npz"))
    r = int(z["s0_root_id"])
    fc, k = int(z["s0_first_child"][r]), int(z["s0_num_children"][r])
    nv = z["s0_num_visits"][fc:fc + k]
    assert np.array_equal(as_distribution(nv, 1.0), z["s0_probs"])
    p0 = as_distribution(np.array([1, 3, 3, 0], np.float32), 0.0)
    assert np.allclose(p0, [0.0, 0.5, 0.5, 0.0], atol=1e-15) and p0[1] == p0[2]   # ties stay (search_tree.py:338-339)


def test_random_player_whole_games_and_metrics():
    """Player.read with the random mover (replay-buffer seeding, policy_trainer.py:145-158)."""
    from azalea_amd import AzaleaAgent, HexGame, Player
    from azalea_amd.prep import torch_batch_replays
    agent = AzaleaAgent(lambda: HexGame(5))
    agent.seed(1)
    pl = Player(None, [agent])
    frame, m = pl.read(60)
    pl.stop()
    assert len(frame) >= 60 and m["games"] >= 3 and m["moves_per_game"] == len(frame)
    assert all(r in (1.0, -1.0) for r in frame.reward)
    batch = torch_batch_replays([frame[i] for i in range(8)])
    assert set(batch) == {"color", "legal_moves", "result", "board", "moves_prob", "reward"}


def test_policy_state_dict_roundtrip(tmp_path):
    from azalea_amd import Policy
    cfg = dict(device="cpu", network="HexNetwork", board_size=5, num_blocks=1, base_chans=8,
               simulations=20, search_batch_size=10, exploration_coef=0.5, exploration_depth=3,
               exploration_noise_alpha=0.03, exploration_noise_scale=0.25,
               exploration_temperature=1.0, seed=7)
    p = Policy()
    with pytest.raises(RuntimeError):
        p.net
    p.initialize(cfg)
    path = str(tmp_path / "x.policy.pth")
    torch.save({"policy": p.state_dict()}, path)      # policy_trainer.py:168-174 schema
    q = Policy.load(path, device="cpu")
    assert q.simulations == 20 and q.board_size == 5
    for a, b in zip(p.net.state_dict().values(), q.net.state_dict().values()):
        assert torch.equal(a, b)
    assert q.rng.randint(1 << 30) == p.rng.randint(1 << 30)


def test_player_read_survives_a_skipped_game(monkeypatch):
    """parallel_player.py:71-76: a game that raises SearchTreeFull is skipped and reading goes on
    (the reference's batch_examples just sees an empty frame); the skip shows up in game_error."""
    from azalea_amd import AzaleaAgent, HexGame, Player
    from azalea_amd import parallel_player as pp
    from azalea_amd.policy import SearchTreeFull
    real, calls = pp.play_game, []

    def flaky(agents, **kw):
        calls.append(1)
        if len(calls) in (1, 3):
            raise SearchTreeFull("too many nodes")
        return real(agents, **kw)
    monkeypatch.setattr(pp, "play_game", flaky)
    agent = AzaleaAgent(lambda: HexGame(4))
    agent.seed(3)
    pl = Player(None, [agent])
    frame, metrics = pl.read(20)
    assert len(frame) >= 20 and metrics["games"] >= 2 and metrics["game_error"] == 2
    frame, metrics = pl.read(5)
    assert len(frame) >= 5 and metrics.get("game_error", 0) == 0
    pl.stop()

    def always(agents, **kw):
        raise SearchTreeFull("too many nodes")
    monkeypatch.setattr(pp, "play_game", always)
    monkeypatch.setattr(pp.Player, "MAX_BARREN_PRODUCTIONS", 5)
    with pytest.raises(RuntimeError):
        Player(None, [agent]).read(5)


def test_rows_to_frame_whole_table_passes():
    """Engine rows -> ReplayDataFrame (the struct-of-lists contract of replay_buffer.py:11-38): dtypes,
    ascending legal moves, moves_prob cut to k, np.float32 rewards; a mismatching nlegal is refused."""
    from azalea_amd.parallel_player import rows_to_frame
    rng = np.random.RandomState(2)
    P, n = 37, 5
    board = rng.randint(0, 3, (P, n, n)).astype(np.int32)
    board[3] = 0
    board[4] = 1                                     # no legal move at all
    k = (board.reshape(P, -1) == 0).sum(1).astype(np.int32)
    prob = rng.rand(P, n * n).astype(np.float32)
    rows = dict(board=board, color=rng.randint(0, 2, P).astype(np.int32), nlegal=k, moves_prob=prob,
                reward=rng.choice([-1.0, 1.0], P).astype(np.float32), game_uid=np.arange(P))
    f = rows_to_frame(rows)
    assert len(f) == P
    for i in range(P):
        st = f.state[i]
        want = (np.flatnonzero(board[i].ravel() == 0) + 1).astype(np.int32)
        assert st.legal_moves.dtype == np.int32 and np.array_equal(st.legal_moves, want)
        assert st.board.dtype == np.int32 and np.array_equal(st.board, board[i]) and st.result == 0
        assert st.color == rows["color"][i] and isinstance(st.color, int)
        assert f.moves_prob[i].dtype == np.float32 and np.array_equal(f.moves_prob[i], prob[i, :k[i]])
        assert type(f.reward[i]) is np.float32 and f.reward[i] == rows["reward"][i]
    assert len(rows_to_frame({key: v[:0] for key, v in rows.items()})) == 0
    rows["nlegal"] = k + 1
    with pytest.raises(AssertionError):
        rows_to_frame(rows)


def test_record_layout_matches_the_header():
    """AZX_RECORD_BYTES and the field offsets of the multi-GPU replay record (include/azx.h) against the
    numpy view the host side packs with."""
    from azalea_amd import _lib
    from azalea_amd import distributed as azd
    hdr = open(os.path.join(ROOT, "include", "azx.h")).read()
    assert "((16 + 5 * (cells) + 15) / 16 * 16)" in hdr
    for cells in (9, 25, 121, 169):
        dt = azd.record_dtype(cells)
        assert dt.itemsize == _lib.record_bytes(cells) == (16 + 5 * cells + 15) // 16 * 16
        assert [dt.fields[f][1] for f in ("game_uid", "reward", "color", "nlegal", "moves_prob", "board")] == \
            [0, 8, 12, 14, 16, 16 + 4 * cells]
