"""Whole-game distribution parity of THROUGHPUT MODE (device RNG, k_play / k_choose / k_advance: the mode every
reported number runs in) against the reference's algorithm under its own RNG.

Parity mode replays the reference's games bit for bit because numpy's RandomState stays on the host (G5).  Throughput
mode draws its Dirichlet rows and its moves on the device (Gamma by CDF inversion, Philox), so it cannot be compared
game by game -- it is compared as a DISTRIBUTION: the engine plays >= 4096 games of 7x7 Hex (and 512 of 11x11) with
the uniform-prior / board-hash evaluator, noise eps 0.25 / alpha 0.3, T = 1, exploration_depth 6, and the CPU oracle
plays as many under numpy's `RandomState.dirichlet` + `multinomial` exactly as mcts.py:126-131 and policy.py:142-160
do (tests/oracle_games.py; held bit for bit to the REFERENCE's own games of these configurations by G11).  Two-sample
tests (tests/game_stats.py), each at p > 1e-3:

  * game-length histogram (chi-square), first-player win rate (two-proportion z);
  * per ply: root-child-visit entropy and mcts.py:291's search value (Kolmogorov-Smirnov), root width, total child
    visits incl. the carried subtree (search_tree.py:109-110), support of the recorded moves_prob (chi-square);
  * recorded moves_prob rows are visits / sum below `exploration_depth` and uniform over the visit maxima from it on
    (search_tree.py:327-344), while the noise still acts there (policy.py:142-149: T is gated by depth, noise is not).

Thresholds: the oracle against ITSELF on disjoint seeds gives min p = 0.02 over the 77 tests at 7x7 (0.024 over 67
at 11x11), re-measured by this test on the box; the engine is compared with both oracle runs pooled (68 / 60 tests:
measured min p 1.5e-3 / 0.056, 3 / 0 of them below 0.05); an oracle that (wrongly) gates the noise by depth as well
fails the same tests at p < 1e-100 (power check below).  The engine's games are a deterministic function of its seed,
so the verdict does not flicker from run to run.

Round 6 adds the configuration every reported number runs on (11x11, Dirichlet alpha 0.03, eps 0.25, exploration_depth 15,
c_puct 0.5, batch 10: config/hex11_train_config.yml:19-36) -- with the uniform-hash evaluator (512 games, 100 -> 110
selects) and on the headline's own kernels (the 6x64 device network, 256 games of 60 -> 70 selects, against the oracle's
fp32 forward: fixture G13).  G11's "11h" set pins the oracle sampler to the REFERENCE's games at exactly these
hyper-parameters.  Measured: 85 tests each, min p 0.013 / 0.0048, 4 / 4 below 0.05; the oracle against itself (92 tests)
min p 6.6e-4 -- inside the family-wise 5 % bound the calibration is held to.

What this test found when it was first run (round 5): the per-ply `search_value` metric of throughput mode was the
raw sum of the leaf values, not divided by num_batches * batch_size as mcts.py:291 does (KS p = 0 at every ply;
everything else agreed) -- fixed in k_choose.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import game_stats as gs          # noqa: E402
import oracle_games as og        # noqa: E402

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {
    # tag: (n, sims, games, plies tested, min games still running at a tested ply)
    "7": (7, 60, 4096, [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 20, 24, 28], 200),
    "11": (11, 100, 512, [0, 1, 2, 3, 4, 5, 6, 8, 12, 20, 30, 40, 50], 150),
}
CFG = dict(batch=10, c=0.5, depth=6, alpha=0.3, eps=0.25, temp=1.0)
# the configuration every reported number runs on (config/hex11_train_config.yml:19-36, bench.py make_engine): alpha 0.03 is
# where a Dirichlet row over ~100 children is one or two spikes above a float-underflow tail -- the regime the device's
# Gamma-by-CDF-inversion sampler (log2 domain) was built for
CFG_HEADLINE = dict(batch=10, c=0.5, depth=15, alpha=0.03, eps=0.25, temp=1.0)
HEADLINE_PLIES = [0, 1, 2, 3, 4, 5, 6, 8, 10, 12, 14, 15, 16, 18, 20, 30, 40, 50]


def oracle_sample(tmp_path, n, sims, games, seed0, extra=(), CFG=CFG):
    out = str(tmp_path / ("oracle_%d_%d_%d%s.npz" % (n, games, seed0, "_w" if "--noise-until" in extra else "")))
    cmd = [sys.executable, os.path.join(HERE, "oracle_games.py"), "--n", str(n), "--sims", str(sims),
           "--games", str(games), "--seed0", str(seed0), "--out", out]
    for k, v in CFG.items():
        cmd += ["--" + k, str(v)]
    subprocess.check_call(cmd + list(extra))
    return dict(np.load(out))


def engine_sample(n, sims, games, seed, net=None, CFG=CFG):
    """The first `games` games (uids 0..games-1: the first generation of the pool, so no length bias from taking
    whichever games finish first) of a throughput-mode engine, summarised like tests/oracle_games.py.  `net`:
    (blocks, chans, state_dict as numpy) -> the device network evaluates the leaves (the headline's kernels: k_mcts
    phases + tower + heads + k_choose + k_advance); else the uniform-prior / board-hash evaluator (k_play)."""
    from azalea_amd import engine as eng
    cells = n * n
    kw = dict(evaluator=eng.EVAL_UNIFORM_HASH) if net is None else dict(evaluator=eng.EVAL_RESNET, num_blocks=net[0], base_chans=net[1])
    E = eng.Engine(board_size=n, n_games=games, simulations=sims, search_batch_size=CFG["batch"],
                   exploration_coef=CFG["c"], exploration_depth=CFG["depth"], noise_alpha=CFG["alpha"],
                   noise_scale=CFG["eps"], temperature=CFG["temp"], seed=seed, **kw)
    if net is None:
        E.set_prior_table(og.prior_table(n))
    else:
        E.set_weights(net[2])
    out = dict(length=np.zeros(games, np.int32), first_wins=np.zeros(games, np.int8),
               **{c: np.full((games, cells), np.nan, np.float32) for c in og.COLUMNS})
    seen = np.zeros(games, bool)
    onehot_rows = checked = 0
    for _ in range(64):
        rows, st = E.play(games * 4)
        m = E.play_row_metrics()
        assert len(m) == len(rows["reward"]) and st["game_errors"] == 0
        uid = rows["game_uid"]
        starts = np.flatnonzero(m[:, 3] > 0.5)
        ends = np.r_[starts[1:], len(uid)]
        for s, e in zip(starts, ends):
            u = int(uid[s])
            assert (uid[s:e] == u).all()
            if u >= games:
                continue
            assert not seen[u]
            seen[u] = True
            L = e - s
            out["length"][u] = L
            out["first_wins"][u] = rows["reward"][s] > 0           # row 0 is the first player's: +1 = they won
            prob = rows["moves_prob"][s:e].astype(np.float64)
            nl = rows["nlegal"][s:e]
            assert np.array_equal(nl, cells - np.arange(L))
            for i in range(L):
                p = prob[i, :nl[i]]
                assert abs(p.sum() - 1.0) < 1e-5 and (prob[i, nl[i]:] == 0).all()
                sup = int((p > 0).sum())
                out["support"][u, i] = sup
                if i < CFG["depth"]:
                    out["entropy"][u, i] = og.entropy(p)               # T = 1: moves_prob = visits / sum
                else:                                                   # T = 0: uniform over the maxima
                    assert np.allclose(p[p > 0], 1.0 / sup, rtol=1e-6)
                    onehot_rows += sup == 1
                    checked += 1
            out["width"][u, :L] = m[s:e, 1]
            out["mean_visits"][u, :L] = m[s:e, 4]
            out["search_value"][u, :L] = m[s:e, 0]
            out["action_prob"][u, :L] = np.exp(m[s:e, 2].astype(np.float64))
            # below the depth the visit distribution is on the row itself: the metrics must agree with it
            d = min(L, CFG["depth"])
            assert np.array_equal(out["support"][u, :d], out["width"][u, :d])
        if seen.all():
            break
    E.close()
    assert seen.all(), "%d first-generation games never harvested" % (~seen).sum()
    assert checked > 0 and onehot_rows > 0        # (how often the maximum is unique is the support@ply comparison's business)
    return out


def compare_with_oracle(tag, tmp_path, n, sims, games, plies, min_games, cfg, seed, net=None, extra=(), oracle=None):
    """Engine sample against two pooled oracle samples (disjoint seeds), with the oracle-vs-oracle comparison as the
    calibration of the thresholds.  `oracle`: (a, b) already loaded (tests/golden/g13_*: the oracle's network games
    take minutes on the host, so they are recorded once by tests/golden/make_oracle_net_games.py)."""
    from scipy import stats as sps
    a, b = oracle if oracle is not None else (oracle_sample(tmp_path, n, sims, games, 0, extra, cfg),
                                               oracle_sample(tmp_path, n, sims, games, 100000, extra, cfg))
    same = gs.compare(a, b, cfg["depth"], plies, min_games)
    e = engine_sample(n, sims, games, seed=seed, net=net, CFG=cfg)
    if os.environ.get("AZX_DIST_DUMP"):      # keep the three samples for offline analysis
        for name, smp in (("engine", e), ("oracle_a", a), ("oracle_b", b)):
            np.savez_compressed(os.path.join(os.environ["AZX_DIST_DUMP"], "dist_%s_%s.npz" % (tag, name)), **smp)
    ref = {k: np.concatenate([a[k], b[k]]) for k in e}       # the reference sample: both oracle runs, 2 x the engine's
    res = gs.compare(e, ref, cfg["depth"], plies, min_games)
    low, low_same = sum(v < 0.05 for v in res.values()), sum(v < 0.05 for v in same.values())
    print("[%s] oracle vs oracle: %d tests, worst %s p=%.3g, %d below 0.05" % ((tag, len(same)) + gs.worst(same) + (low_same,)))
    print("[%s] engine vs oracle: %d tests, worst %s p=%.3g, %d below 0.05" % ((tag, len(res)) + gs.worst(res) + (low,)))
    print("[%s] lengths: engine %.2f oracle %.2f / %.2f; first player wins: %.4f vs %.4f / %.4f" % (
        tag, e["length"].mean(), a["length"].mean(), b["length"].mean(),
        e["first_wins"].mean(), a["first_wins"].mean(), b["first_wins"].mean()))
    # every statistic the oracle's games have is tested on the engine's (entropy: below the depth, where the rows carry it)
    assert len(res) >= len(same) - sum(p >= cfg["depth"] for p in plies) and len(res) >= 50
    # calibration: the oracle against itself.  Over ~90 tests the smallest of that many uniform p-values is below 1e-3
    # nine times in a hundred (the 11h pair of seeds: entropy@3 at 6.6e-4, the engine's own worst 0.013), so it is held to
    # the family-wise 5 % bound; the engine is held to the fixed, stricter-per-test P_MIN
    assert gs.worst(same)[1] > min(gs.P_MIN, 0.05 / len(same)), gs.worst(same)
    bad = {k: v for k, v in res.items() if v <= gs.P_MIN}
    assert not bad, bad
    # ... and no drift too small for any single test: the count of p < 0.05 stays binomial (99.9 % quantile)
    assert low <= sps.binom.ppf(0.999, len(res), 0.05), (low, len(res))
    return res, same


@pytest.mark.parametrize("tag", ["7", "11"])
def test_throughput_mode_plays_the_reference_game_distribution(tag, tmp_path):
    n, sims, games, plies, min_games = CASES[tag]
    compare_with_oracle(tag, tmp_path, n, sims, games, plies, min_games, CFG, seed=20261003)


def test_throughput_mode_at_the_headline_hyper_parameters(tmp_path):
    """The configuration every reported number runs on: 11x11, Dirichlet alpha 0.03, eps 0.25,
    exploration_depth 15, c_puct 0.5, search batch 10 (config/hex11_train_config.yml:19-36) -- 512 games of 100 -> 110
    selects with the uniform-prior / board-hash evaluator in k_play, against the oracle under numpy's
    `RandomState.dirichlet` (mcts.py:126-131), whose sampler is pinned to the REFERENCE's own games at exactly these
    hyper-parameters by G11's "11h" set (tests/test_oracle_golden.py)."""
    compare_with_oracle("11h", tmp_path, 11, 100, 512, HEADLINE_PLIES, 150, CFG_HEADLINE, seed=20261005)


def golden_oracle_games(name):
    """(a, b): the two oracle samples of tests/golden/<name> expanded to the [games, cells] columns gs.compare reads."""
    z = np.load(os.path.join(HERE, "golden", name))
    cells, plies = int(z["cells"]), z["plies"]
    out = []
    for half in ("a", "b"):
        smp = dict(length=z["length_" + half], first_wins=z["first_wins_" + half])
        for c in og.COLUMNS:
            full = np.full((len(smp["length"]), cells), np.nan, np.float32)
            full[:, plies] = z[c + "_" + half]
            smp[c] = full
        out.append(smp)
    return tuple(out), z


def test_headline_kernels_at_the_headline_hyper_parameters(tmp_path):
    """The headline itself, whole games: 11x11, the 6x64 DEVICE network (k_tower_f16x3_s16 + k_heads_mfma on G3's seeded
    weights with non-trivial BatchNorm statistics), alpha 0.03 / eps 0.25 / depth 15 / c 0.5 / batch 10, 256 games of
    60 -> 70 selects in k_mcts's phases + k_choose + k_advance -- against the oracle playing with ITS fp32 forward of the
    same weights under numpy's RNG.  The oracle's 2 x 256 games cost ~20 core-minutes, so they are a committed fixture
    (tests/golden/g13_oracle_net_games_11h.npz, written by tests/golden/make_oracle_net_games.py from
    tests/oracle_games.py; tests/test_oracle_golden.py replays its first games live), regenerated here when absent."""
    n, sims, games, blocks, chans = 11, 60, 256, 6, 64
    z = np.load(os.path.join(HERE, "golden", "g3_forward_11_6x64.npz"))
    state = {k[2:]: z[k] for k in z.files if k.startswith("w:") and z[k].dtype.kind == "f"}
    path = os.path.join(HERE, "golden", "g13_oracle_net_games_11h.npz")
    oracle = None
    extra = ()
    if os.path.exists(path):
        oracle, meta = golden_oracle_games("g13_oracle_net_games_11h.npz")
        assert (int(meta["n"]), int(meta["sims"]), int(meta["games"])) == (n, sims, games)
        assert set(HEADLINE_PLIES) <= set(meta["plies"].tolist())
        assert all(float(meta["cfg_" + k]) == float(v) for k, v in CFG_HEADLINE.items())
    else:
        wpath = str(tmp_path / "weights.npz")
        np.savez(wpath, **state)
        extra = ["--weights", wpath, "--blocks", str(blocks), "--chans", str(chans)]
    compare_with_oracle("net11h", tmp_path, n, sims, games, HEADLINE_PLIES, 80, CFG_HEADLINE, seed=20261006,
                        net=(blocks, chans, state), extra=extra, oracle=oracle)


def test_distribution_test_has_the_power_to_see_a_wrong_noise_gate(tmp_path):
    """The same comparison rejects an implementation that gates the Dirichlet noise by exploration_depth as well
    (the obvious misreading of policy.py:142-149)."""
    n, sims, games, plies, min_games = CASES["7"]
    games = 1024
    a = oracle_sample(tmp_path, n, sims, games, 0)
    w = oracle_sample(tmp_path, n, sims, games, 300000, extra=["--noise-until", str(CFG["depth"])])
    res = gs.compare(a, w, CFG["depth"], plies, min_games)
    assert res["width@6"] < 1e-50 and res["width@8"] < 1e-50
    assert all(res["width@%d" % p] > gs.P_MIN for p in range(CFG["depth"]))    # and only from the depth on


def test_throughput_mode_with_the_device_network_plays_the_reference_game_distribution(tmp_path):
    """The same comparison on the HEADLINE's kernels: leaves evaluated by the device network (a seeded 1x64 HexNetwork on
    7x7 with non-trivial BatchNorm statistics: the split-f16 tower + heads), the search in k_mcts's phases, the move
    draw and the step in k_choose / k_advance -- against the oracle playing with ITS fp32 forward of the same weights
    under numpy's RNG (the two forwards agree to 1e-4 by the net parity tests; a prior differing in the sixth digit
    moves a visit now and then, not a distribution).  1 024 games, 40 -> 50 selects, exploration_depth 6."""
    import torch
    from azalea_amd.network import HexNetwork
    n, sims, games, blocks, chans = 7, 40, 1024, 1, 64
    plies, min_games = [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 20, 24], 150
    torch.manual_seed(3)
    net = HexNetwork(board_size=n, num_blocks=blocks, base_chans=chans).eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.6, 1.4)
    state = {k: v.detach().numpy() for k, v in net.state_dict().items() if v.dtype.is_floating_point}
    wpath = str(tmp_path / "weights.npz")
    np.savez(wpath, **state)
    extra = ["--weights", wpath, "--blocks", str(blocks), "--chans", str(chans)]
    compare_with_oracle("net7", tmp_path, n, sims, games, plies, min_games, CFG, seed=20261004,
                        net=(blocks, chans, state), extra=extra)
