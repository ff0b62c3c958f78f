"""Pins the CPU oracle (oracle/) against the golden vectors generated from the reference
(tests/golden/make_golden.py).  CPU only.  Bit-exact for rules/trees, <=1e-4 for the network."""
import glob
import os
import zlib

import numpy as np
import pytest

from oracle import oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


# ------------------------------------------------------------------------------- G1 / G2
@pytest.mark.parametrize("n", [3, 5, 11, 13])
def test_g1_movegen_playouts(n):
    z = np.load(os.path.join(GOLDEN, "g1_movegen.npz"))
    moves, res, nl = z["moves_%d" % n], z["result_%d" % n], z["nlegal_%d" % n]
    crc, length, final = z["legalcrc_%d" % n], z["length_%d" % n], z["final_%d" % n]
    for g in range(len(moves)):
        h = orc.Hex(n)
        for p in range(length[g]):
            lm = h.legal_moves()
            assert len(lm) == nl[g, p]
            assert zlib.crc32(lm.astype(np.int32).tobytes()) == crc[g, p]
            assert h.color == p % 2
            h.step(moves[g, p])
            assert h.result == res[g, p]
        assert h.result in (1, 3)
        assert len(h.legal_moves()) == 0          # hex.py:152-153
        assert np.array_equal(h.board, final[g])


def test_g1_illegal_moves_rejected():
    h = orc.Hex(5)
    h.step(3)
    with pytest.raises(ValueError):
        h.step(3)
    with pytest.raises(ValueError):
        h.step(0)


@pytest.mark.parametrize("n", [5, 11, 13])
def test_g2_flip(n):
    z = np.load(os.path.join(GOLDEN, "g2_flip.npz"))
    fb, fm = orc.flip_board_moves(z["board_%d" % n], z["moves_%d" % n])
    assert np.array_equal(fb, z["fboard_%d" % n])
    assert np.array_equal(fm, z["fmoves_%d" % n])
    # involution
    b2, m2 = orc.flip_board_moves(fb, fm)
    assert np.array_equal(b2, z["board_%d" % n]) and np.array_equal(m2, z["moves_%d" % n])


# ------------------------------------------------------------------------------- G3
@pytest.mark.parametrize("tag", ["5_1x8", "13_2x32", "11_6x64"])
def test_g3_forward(tag):
    z = np.load(os.path.join(GOLDEN, "g3_forward_%s.npz" % tag))
    n, blocks, chans = [int(x) for x in z["cfg"]]
    state = {k[2:]: z[k] for k in z.files if k.startswith("w:")}
    net = orc.Net(n, blocks, chans, state)
    value, logprob = net.forward(z["board"], z["legal_moves"])
    legal = z["legal_moves"] > 0
    assert np.abs(value - z["value"]).max() <= 1e-4
    assert np.abs(logprob - z["moves_logprob"])[legal].max() <= 1e-4
    # padded entries sit ~99 below the row's log-sum-exp (network.py:150-151)
    if (~legal).any():
        assert np.abs(logprob - z["moves_logprob"])[~legal].max() <= 1e-3


# ------------------------------------------------------------------------------- G4
G4 = sorted(glob.glob(os.path.join(GOLDEN, "g4_search_*.npz")))


def _replay_prefix(z):
    n = int(z["cfg_n"])
    g = orc.Hex(n)
    for m in z["prefix_moves"]:
        g.step(int(m))
    return g


def _check_tree(tree, z, pre):
    d = tree.dump()
    assert d["num_nodes"] == int(z[pre + "num_nodes"])
    assert d["root_id"] == int(z[pre + "root_id"])
    for name in ("parent", "first_child", "num_children"):
        assert np.array_equal(d[name], z[pre + name]), name
    for name in ("num_visits", "total_value", "prior_prob"):
        assert np.array_equal(bits(d[name]), bits(z[pre + name])), name


@pytest.mark.parametrize("path", G4, ids=[os.path.basename(p)[10:-4] for p in G4])
def test_g4_search_tape(path):
    """Feed the oracle the exact (value, prior) stream the reference's evaluate_batch produced:
    tree topology, visit counts, total values and priors must match bit for bit."""
    z = np.load(path)
    game = _replay_prefix(z)
    tree = orc.Tree(1 << 18)
    follow = int(z["follow"])
    for step in range(follow + 1):
        pre = "s%d_" % step
        assert np.array_equal(game.board, z[pre + "root_board"])
        assert np.array_equal(game.legal_moves(), z[pre + "legal_moves"])
        ev = orc.TapeEval(z[pre + "tape_value"], z[pre + "tape_nch"], z[pre + "tape_prior"],
                          z[pre + "tape_off"])
        eps = float(z["cfg_eps"])
        noise = z[pre + "noise"] if eps else None
        st = orc.search(tree, game, ev, int(z["cfg_sims"]), int(z["cfg_batch"]),
                        float(z["cfg_c"]), eps, noise)
        assert st.status == 0 and not ev.mismatch
        # every non-terminal tape row consumed
        assert st.n_eval == int((z[pre + "tape_nch"] > 0).sum())
        _check_tree(tree, z, pre)
        assert abs(st.search_value - float(z[pre + "search_value"])) <= 1e-6
        nv, tv, pp, rv, rt = tree.root_stats()
        assert np.array_equal(orc.as_distribution(nv, 1.0), z[pre + "probs"])
        assert np.float32(rt) / np.float32(rv) == z[pre + "value"]
        if step < follow:
            mid = int(z[pre + "move_id"])
            lm = game.legal_moves()
            tree.move(mid)
            game.step(int(lm[mid]))


@pytest.mark.parametrize("path", [p for p in G4 if "_u0" in p or "_uh" in p],
                         ids=lambda p: os.path.basename(p)[10:-4])
def test_g4_search_uniform_builtin(path):
    """Same trees from the oracle's built-in stub evaluator (uniform priors by-k table +
    fnv1a value hash): pins ofnv1a and the value formula too."""
    z = np.load(path)
    n = int(z["cfg_n"])
    table = np.zeros(n * n + 1, np.float32)
    nch, off, pr = z["s0_tape_nch"], z["s0_tape_off"], z["s0_tape_prior"]
    for i, k in enumerate(nch):
        if k:
            table[k] = pr[off[i]]
    game = _replay_prefix(z)
    tree = orc.Tree(1 << 18)
    ev = orc.UniformEval(hash_value=str(z["mode"]) == "uniformhash", prior_by_k=table)
    st = orc.search(tree, game, ev, int(z["cfg_sims"]), int(z["cfg_batch"]), float(z["cfg_c"]))
    assert st.status == 0
    _check_tree(tree, z, "s0_")


def test_tree_full_reports_status():
    game = orc.Hex(11)
    tree = orc.Tree(500)
    st = orc.search(tree, game, orc.UniformEval(), 40, 10, 0.5)
    assert st.status == -1


# ------------------------------------------------------------------------------- G5
def _stub_eval(mode):
    """Python restatement of make_golden.StubNet feeding the oracle (same torch/numpy calls
    for log_softmax/exp as the reference run used, so priors are bit-identical)."""
    import torch

    def fn(boards, lm):
        B, K = lm.shape
        value = np.zeros(B, np.float32)
        logit = np.zeros((B, K), np.float32)
        for i in range(B):
            h = orc.fnv1a(boards[i].ravel())
            if mode != "uniform0":
                value[i] = np.float32((h & 0xFFFF) / 32768.0 - 1.0)
            if mode == "hashprior":
                for j in range(K):
                    t = int(lm[i, j])
                    if t:
                        x = (h ^ (t * 2654435761)) & 0xFFFFFFFF
                        x = (x * 2246822519) & 0xFFFFFFFF
                        logit[i, j] = np.float32(((x >> 13) & 0xFF) / 64.0)
        lt = torch.tensor(logit)
        lt.masked_fill_(torch.tensor(lm == 0), -99)
        lp = torch.log_softmax(lt, dim=1).numpy()
        return value, np.exp(lp)
    return orc.CallbackEval(fn)


G5 = sorted(glob.glob(os.path.join(GOLDEN, "g5_game_*.npz")))


@pytest.mark.parametrize("path", G5, ids=[os.path.basename(p)[8:-4] for p in G5])
def test_g5_full_game_trace(path):
    z = np.load(path)
    n = int(z["cfg_n"])
    result, rows, reward = orc.play_game(
        n, _stub_eval(str(z["mode"])), simulations=int(z["cfg_sims"]),
        batch_size=int(z["cfg_batch"]), c_puct=float(z["cfg_c"]),
        exploration_depth=int(z["cfg_depth"]), noise_alpha=float(z["cfg_alpha"]),
        noise_scale=float(z["cfg_eps"]), temperature=float(z["cfg_temp"]),
        seed=int(z["cfg_seed"]), move_sampling=bool(z["cfg_sampling"]),
        move_exploration=bool(z["cfg_explore"]))
    assert result == int(z["result"])
    assert len(rows) == len(z["board"])
    for i, r in enumerate(rows):
        k = int(z["nlegal"][i])
        assert np.array_equal(r["board"], z["board"][i]), i
        assert r["color"] == z["color"][i]
        assert np.array_equal(r["legal_moves"], z["legal_moves"][i, :k])
        assert np.array_equal(bits(r["moves_prob"]), bits(z["moves_prob"][i, :k])), i
    assert np.array_equal(reward, z["reward"])
    metrics = dict(zip([str(s) for s in z["metric_names"]], z["metric_values"]))
    assert metrics["moves_per_game"] == len(rows)
    assert abs(np.mean([r["num_nodes"] for r in rows]) - metrics["search_tree_nodes"]) < 1e-6
    assert abs(np.mean([r["search_value"] for r in rows]) - metrics["search_value"]) < 1e-6


# ------------------------------------------------------------------------------- G5 with the real network
def test_g5r_real_net_game_from_the_reference_tape():
    """BASELINE configs[0] in small (one 11x11 game, 40 sims, 6x64 net): the oracle, handed the
    reference's own Network.run outputs in call order, must ask for exactly the inputs the reference's
    search produced (flipped boards / move lists) and replay the recorded game bit for bit."""
    from run_tape import RunTape
    z = np.load(os.path.join(GOLDEN, "g5r_game_11_6x64.npz"))
    tape = RunTape(z)

    def fn(boards, lm):
        value, logprob = tape.next_call(boards, lm)
        return value, np.exp(logprob)          # mcts.py:206
    result, rows, reward = orc.play_game(
        int(z["cfg_n"]), orc.CallbackEval(fn), simulations=int(z["cfg_sims"]), batch_size=int(z["cfg_batch"]),
        c_puct=float(z["cfg_c"]), exploration_depth=int(z["cfg_depth"]), noise_alpha=float(z["cfg_alpha"]),
        noise_scale=float(z["cfg_eps"]), temperature=float(z["cfg_temp"]), seed=int(z["cfg_seed"]))
    assert tape.row == len(tape) and tape.call == len(tape.calls)
    assert result == int(z["result"]) and len(rows) == len(z["board"])
    for i, r in enumerate(rows):
        k = int(z["nlegal"][i])
        assert np.array_equal(r["board"], z["board"][i]) and r["color"] == z["color"][i]
        assert np.array_equal(r["legal_moves"], z["legal_moves"][i, :k])
        assert np.array_equal(bits(r["moves_prob"]), bits(z["moves_prob"][i, :k])), i
    assert np.array_equal(reward, z["reward"])


def test_g5r_oracle_network_on_the_game_s_own_leaves():
    """The oracle's fp32 network on positions the reference's search really evaluated (every 23rd
    Network.run row of the recorded game, both colours, early to late) within 1e-4 of the reference."""
    from run_tape import RunTape
    z = np.load(os.path.join(GOLDEN, "g5r_game_11_6x64.npz"))
    w = np.load(os.path.join(GOLDEN, str(z["cfg_net"])))
    tape = RunTape(z)
    n = int(z["cfg_n"])
    rows = np.arange(0, len(tape), 23)
    boards, lm = tape.inputs(rows, n)
    net = orc.Net(n, 6, 64, {k[2:]: w[k] for k in w.files if k.startswith("w:")})
    v, lp = net.forward(boards, lm)
    assert np.abs(v - tape.value[rows]).max() <= 1e-4
    for j, r in enumerate(rows):
        a, b = int(tape.off[r]), int(tape.off[r + 1])
        assert np.abs(lp[j, :b - a] - tape.logprob[a:b]).max() <= 1e-4


def test_timed_network_path_matches_the_parity_path():
    """bench.py's cpu_baseline times net_fast.c (blocked convolutions, FMA); it must be the same network as the
    pinned parity path net.c: value and legal log-probabilities within 1e-5 on 11x11 / 6x64, 13x13 / 2x256 (the
    configs[4] width), a 7x7 and an odd-sized net whose channel count is not a multiple of the register block."""
    import torch
    from azalea_amd.network import HexNetwork
    for n, blocks, chans, B in ((11, 6, 64, 6), (13, 2, 256, 2), (7, 1, 64, 3), (5, 2, 8, 3), (9, 2, 40, 2)):
        torch.manual_seed(n)
        net = HexNetwork(board_size=n, num_blocks=blocks, base_chans=chans).eval()
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.6, 1.4)
        N = orc.Net(n, blocks, chans, {k: v.detach().numpy() for k, v in net.state_dict().items()})
        rng = np.random.RandomState(n)
        boards = rng.randint(0, 3, (B, n, n)).astype(np.int32)
        lm = np.zeros((B, n * n), np.int32)
        for i in range(B):
            e = np.flatnonzero(boards[i].ravel() == 0) + 1
            lm[i, :len(e)] = e
        v, lp = N.forward(boards, lm)
        vf, lpf = N.forward(boards, lm, fast=True)
        assert np.abs(v - vf).max() <= 1e-5 and np.abs(lp - lpf)[lm > 0].max() <= 1e-5, (n, blocks, chans)


@pytest.mark.parametrize("tag", ["7", "11", "11h"])
def test_oracle_game_sampler_replays_the_reference_games_g11(tag):
    """tests/oracle_games.py -- the reference side of the GPU whole-game distribution test -- plays, seed for seed, the
    games the REFERENCE played in that test's configurations (G11: tests/golden/make_game_stats.py): the same moves,
    winner, and per-ply root width / mean visits / search value / moves_prob support / probability of the move
    drawn, bit for bit."""
    import oracle_games as og
    z = np.load(os.path.join(GOLDEN, "g11_game_summaries.npz"))
    n, sims, bs, c, depth, alpha, eps, temp, games = z["cfg_" + tag]
    cfg = dict(n=int(n), sims=int(sims), batch=int(bs), c=float(c), depth=int(depth), alpha=float(alpha),
               eps=float(eps), temp=float(temp))
    got = og.sample(cfg, range(int(games)), procs=1 if tag == "7" else 4)
    assert np.array_equal(got["length"], z["length_" + tag])
    assert np.array_equal(got["first_wins"], z["first_wins_" + tag])
    assert np.array_equal(got["moves"], z["moves_" + tag])
    for g in range(int(games)):
        L = int(got["length"][g])
        for k in ("width", "support"):
            assert np.array_equal(got[k][g, :L], z[k + "_" + tag][g, :L].astype(np.float32)), (k, g)
        for k in ("mean_visits", "search_value", "action_prob"):
            assert np.array_equal(got[k][g, :L].view(np.uint32), z[k + "_" + tag][g, :L].view(np.uint32)), (k, g)


def test_oracle_network_games_fixture_g13_is_what_the_sampler_plays():
    """G13 (tests/golden/make_oracle_net_games.py): the oracle's 2 x 256 whole games with its fp32 6x64 network at the
    headline's hyper-parameters, the reference side of the GPU test of the headline's kernels.  The fixture is ORACLE
    output, recorded because it costs ~20 core-minutes; the first game of each half is replayed here live -- same moves,
    same per-ply columns at the plies the fixture keeps -- so the file cannot drift from tests/oracle_games.py (which G11
    holds to the reference's own games at these hyper-parameters, uniform evaluator)."""
    import sys
    sys.path.insert(0, GOLDEN)
    try:
        import make_oracle_net_games as mk
    finally:
        sys.path.pop(0)
    import oracle_games as og
    z = np.load(os.path.join(GOLDEN, "g13_oracle_net_games_11h.npz"))
    assert (int(z["n"]), int(z["sims"]), int(z["games"])) == (mk.N, mk.SIMS, mk.GAMES)
    assert all(float(z["cfg_" + k]) == float(v) for k, v in mk.CFG.items()) and z["plies"].tolist() == mk.PLIES
    cfg = mk.config(mk.weights_file())
    got = og.sample(cfg, [mk.SEED0["a"], mk.SEED0["b"]], procs=2)
    for i, half in enumerate(("a", "b")):
        L = int(got["length"][i])
        assert L == int(z["length_" + half][0]) and int(got["first_wins"][i]) == int(z["first_wins_" + half][0])
        assert np.array_equal(got["moves"][i], z["moves_" + half][0])
        for c in og.COLUMNS:
            want, have = z[c + "_" + half][0], got[c][i][mk.PLIES]
            assert np.array_equal(np.isnan(want), np.isnan(have)), (c, half)
            assert np.array_equal(want[~np.isnan(want)].view(np.uint32), have[~np.isnan(have)].view(np.uint32)), (c, half)
