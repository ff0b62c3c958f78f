"""GPU parity cases added in round 2: the real-network self-play game,
the perspective flip checked directly, the device move draw against as_distribution + multinomial,
SearchTreeFull in both modes, the 19x256 tower against the C oracle, the global game index, the
parked-slot path, start prefixes in play mode, and the device record exchange.  All through the C ABI."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-4


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def eng():
    from azalea_amd import engine
    return engine


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


# ---- G5 with the real network (BASELINE configs[0] in small) ---------------------------------------
class TapeNet:
    """Duck-typed Network whose run() hands back the reference's recorded outputs (RunTape checks that
    every call asks about exactly the inputs the reference's search produced)."""

    def __init__(self, tape):
        import torch
        self.tape, self.device = tape, torch.device("cpu")

    def eval(self):
        return self

    def run(self, batch, compute_loss=False):
        import torch
        v, lp = self.tape.next_call(batch["board"].cpu().numpy(), batch["legal_moves"].cpu().numpy())
        return dict(value=torch.tensor(v), moves_logprob=torch.tensor(lp))


def test_g5r_real_net_game_through_policy_and_play_game():
    """One whole reference self-play game with the 6x64 network (11x11, 40 sims, noise + sampling on):
    Policy / AzaleaAgent / play_game over the HIP search, fed the reference's own Network.run outputs,
    reproduce the trace bit for bit -- and every batch of leaves the device selects (boards flipped to
    the first player's view, flipped move lists) equals the reference's, call by call."""
    from run_tape import RunTape
    from azalea_amd import AzaleaAgent, HexGame, Policy
    from azalea_amd.play_game import play_game
    z = np.load(os.path.join(GOLDEN, "g5r_game_11_6x64.npz"))
    tape = RunTape(z)
    n = int(z["cfg_n"])
    p = Policy()
    p.net = TapeNet(tape)
    p.network_type, p.board_size, p.num_blocks, p.base_chans = "stub", n, 0, 0
    p.simulations, p.search_batch_size = int(z["cfg_sims"]), int(z["cfg_batch"])
    p.exploration_coef, p.exploration_depth = float(z["cfg_c"]), int(z["cfg_depth"])
    p.exploration_noise_alpha, p.exploration_noise_scale = float(z["cfg_alpha"]), float(z["cfg_eps"])
    p.exploration_temperature = float(z["cfg_temp"])
    p.settings.update(move_sampling=True, move_exploration=True)
    agent = AzaleaAgent(lambda: HexGame(n), policy=p, device="cpu")
    agent.seed(int(z["cfg_seed"]))
    result, frame, metrics = play_game([agent], collect_data=True)
    assert tape.row == len(tape) and tape.call == len(tape.calls)       # every recorded evaluation was consumed
    assert result == int(z["result"]) and len(frame) == len(z["board"])
    for i in range(len(frame)):
        k = int(z["nlegal"][i])
        st = frame.state[i]
        assert np.array_equal(st.board, z["board"][i]) and st.color == z["color"][i]
        assert np.array_equal(st.legal_moves, z["legal_moves"][i, :k])
        assert np.array_equal(bits(frame.moves_prob[i]), bits(z["moves_prob"][i, :k])), i
    assert np.array_equal(np.array(frame.reward, np.float32), z["reward"])
    want = dict(zip([str(s) for s in z["metric_names"]], z["metric_values"]))
    for name, v in want.items():
        assert abs(float(metrics[name]) - v) <= 1e-5 * max(1.0, abs(v)), name


def test_g5r_device_network_on_every_leaf_of_the_game(eng):
    """The MFMA tower + heads on ALL positions the reference's search evaluated in that game (4k+ leaves,
    both colours, opening to endgame): value and legal log-probabilities within 1e-4 of the reference."""
    from run_tape import RunTape
    z = np.load(os.path.join(GOLDEN, "g5r_game_11_6x64.npz"))
    w = np.load(os.path.join(GOLDEN, str(z["cfg_net"])))
    tape = RunTape(z)
    n = int(z["cfg_n"])
    rows = np.arange(len(tape))
    boards, lm = tape.inputs(rows, n)
    E = eng.Engine(board_size=n, n_games=8, simulations=10, search_batch_size=10,
                   evaluator=eng.EVAL_RESNET, num_blocks=6, base_chans=64)
    E.set_weights({k[2:]: w[k] for k in w.files if k.startswith("w:")})
    value, logprob = E.forward(boards, lm)
    E.close()
    assert np.abs(value - tape.value).max() <= TOL
    worst = 0.0
    for r in rows:
        a, b = int(tape.off[r]), int(tape.off[r + 1])
        worst = max(worst, float(np.abs(logprob[r, :b - a] - tape.logprob[a:b]).max()))
    assert worst <= TOL, worst


# ---- H6: flip_player_board(_moves) checked directly ------------------------------------------------
@pytest.mark.parametrize("tag", ["c_11_p30_s400_hp", "g_13_empty_s40_hp", "f_5_p12_s100_uh", "i_11_p40_s60_noise_b7"])
def test_h6_leaf_boards_and_moves_are_the_reference_flip(eng, tag):
    """azx_get_leaves hands the evaluator boards in the first player's view and move lists flipped with
    their order preserved (hex.py:72-122, mcts.py:178-181).  Compared leaf by leaf with the states the
    reference's evaluate_batch saw (golden G4 tape_board / tape_color), flipped by the host mirror of the
    reference's rule -- which golden G2 pins."""
    from azalea_amd.game.hex import HexGame
    z = np.load(os.path.join(GOLDEN, "g4_search_%s.npz" % tag))
    n = int(z["cfg_n"])
    tb, tc, nch = z["s0_tape_board"], z["s0_tape_color"], z["s0_tape_nch"]
    value, prior, off = z["s0_tape_value"], z["s0_tape_prior"], z["s0_tape_off"]
    E = eng.Engine(board_size=n, n_games=1, simulations=int(z["cfg_sims"]), search_batch_size=int(z["cfg_batch"]),
                   exploration_coef=float(z["cfg_c"]), evaluator=eng.EVAL_EXTERNAL, nodes_per_game=1 << 17,
                   flags=eng.FLAG_NO_COMPACT)
    E.reset(moves=[list(z["prefix_moves"])])
    eps = float(z["cfg_eps"])
    noise = z["s0_noise"][None, :, :] if eps else None
    pos, seen, flipped = 0, 0, 0
    pending = E.search_begin(noise, eps)
    while True:
        if pending:
            boards, lm, slot, k = E.get_leaves()
            v = np.zeros(len(k), np.float32)
            p = np.zeros((len(k), n * n), np.float32)
            for i in range(len(k)):
                while nch[pos] == 0:                    # terminal states never reach the evaluator
                    pos += 1
                state_board = tb[pos].astype(np.int32)
                legal = (np.flatnonzero(state_board.ravel() == 0) + 1).astype(np.int32)
                assert len(legal) == k[i] == nch[pos]
                want_b, want_m = state_board[None], legal[None]
                if tc[pos] == 1:
                    want_b, want_m = HexGame.flip_player_board_moves(want_b, want_m)
                    flipped += 1
                assert np.array_equal(boards[i], want_b[0]), ("board", pos)
                assert np.array_equal(lm[i, :k[i]], want_m[0]), ("moves", pos)
                assert not lm[i, k[i]:].any()
                v[i] = value[pos]
                p[i, :k[i]] = prior[off[pos]:off[pos + 1]]
                pos += 1
                seen += 1
            E.put_evals(v, p)
        pending, done = E.search_step()
        if done:
            break
    E.close()
    assert seen > 20 and flipped > 5


# ---- the device move draw (k_choose) ------------------------------------------------------------------
def _as_distribution(counts, temperature):
    """search_tree.py:327-344 with the same numpy calls."""
    counts = np.asarray(counts, np.float32)
    with np.errstate(divide="ignore"):
        log_pi = np.log(counts.clip(min=1))
    log_pi[counts == 0] = -np.inf
    if temperature:
        log_pi = log_pi / temperature
    else:
        log_pi[log_pi < log_pi.max()] = -np.inf
    log_pi = log_pi.astype(np.float64)
    return np.exp(log_pi - np.logaddexp.reduce(log_pi))


@pytest.mark.parametrize("temperature", [1.0, 0.5, 0.0])
def test_device_move_draw_matches_as_distribution_and_multinomial(eng, temperature):
    """Throughput mode replaces as_distribution + rng.multinomial (search_tree.py:327-344, policy.py:160)
    by k_choose.  (1) The moves_prob row it records equals as_distribution(root child visits, T) to 1e-6
    for T = 1, 0.5 and 0 (ties stay: uniform over the most-visited children).  (2) Its draws follow that
    distribution: chi-square of ~20k draws (independent per-game streams on identical trees)."""
    from scipy import stats
    n, G, sims = 7, 4096, 60
    prefix = [3, 17, 25, 30, 9]
    E = eng.Engine(board_size=n, n_games=G, simulations=sims, search_batch_size=10, exploration_coef=0.5,
                   exploration_depth=999, temperature=temperature, noise_scale=0.0,
                   evaluator=eng.EVAL_UNIFORM_HASH, nodes_per_game=8192, seed=31)
    table = (np.float32(1.0) / np.arange(0, n * n + 1).clip(1).astype(np.float32)).astype(np.float32)
    E.set_prior_table(table)
    draws = []
    probs = None
    for rnd in range(5):
        E.reset(moves=[prefix] * G)           # fresh uids: every round is a new set of per-game RNG streams
        E.search()
        root = E.get_root()
        k = int(root["k"][0])
        visits = root["child_visits"][:, :k]
        assert (visits == visits[0]).all()    # no noise, same position: identical trees
        want = _as_distribution(visits[0], temperature)
        mid, prob = E.debug_choose()
        assert (mid >= 0).all() and (mid < k).all()
        assert np.abs(prob[:, :k] - want[None, :].astype(np.float32)).max() <= 1e-6
        assert (prob[:, k:] == 0).all()
        if temperature == 0.0:
            top = visits[0] == visits[0].max()
            assert top.sum() >= 1 and np.allclose(want[top], 1.0 / top.sum()) and (want[~top] == 0).all()
        assert (want[mid] > 0).all()          # never a zero-probability child
        draws.append(mid)
        probs = want
    E.close()
    draws = np.concatenate(draws)
    support = probs > 0
    obs = np.bincount(draws, minlength=len(probs))[support].astype(np.float64)
    exp = probs[support] * len(draws)
    if support.sum() > 1:
        chi2 = ((obs - exp) ** 2 / exp).sum()
        # p > 1e-4: a correct sampler fails once in ten thousand runs; the seeds are fixed anyway
        assert stats.chi2.sf(chi2, support.sum() - 1) > 1e-4, (chi2, support.sum())
        assert obs.min() > 0


# ---- SearchTreeFull -----------------------------------------------------------------------------------
def test_tree_full_sets_status_and_policy_raises(eng):
    """search_tree.py:258-259: a search that runs out of nodes reports SearchTreeFull (status 1) and
    never writes past its arena."""
    E = eng.Engine(board_size=11, n_games=2, simulations=40, search_batch_size=10,
                   evaluator=eng.EVAL_UNIFORM, nodes_per_game=500)
    E.search()
    assert (E.get_status() == 1).all()
    assert (E.get_root()["num_nodes"] <= 500).all()
    E.close()
    E = eng.Engine(board_size=11, n_games=2, simulations=40, search_batch_size=10,
                   evaluator=eng.EVAL_UNIFORM, nodes_per_game=1 << 16)
    E.search()
    assert (E.get_status() == 0).all()
    E.close()


def test_play_mode_skips_overflowing_games_and_returns_whole_ones(eng):
    """parallel_player.py:71-76 in throughput mode: a game whose tree overflows is dropped (counted in
    game_errors, its slot restarts) and only whole finished games are handed out."""
    n, G, sims = 7, 64, 30
    # a first search of a 7x7 game expands ~40 leaves with 47-48 children each below the root's 49: about
    # 1950 nodes, a little more or less from game to game (duplicate leaves, depth).  Arenas around that size
    # make SOME searches overflow -- the game is dropped, its slot restarts with a new game -- while others
    # get through; later searches need less (one child fewer per ply).
    rows = st = None
    for cap in (1990, 1975, 1960, 1945, 1930, 1915, 1900):
        E = eng.Engine(board_size=n, n_games=G, simulations=sims, search_batch_size=10, exploration_depth=4,
                       evaluator=eng.EVAL_UNIFORM, nodes_per_game=cap, seed=77)
        rows, st = E.play(600, max_plies=400)
        E.close()
        if st["game_errors"] > 0 and st["games"] > 0:
            break
    assert st["game_errors"] > 0 and st["games"] > 0, (cap, st["game_errors"], st["games"])
    uid = rows["game_uid"]
    starts = np.flatnonzero(np.r_[True, uid[1:] != uid[:-1]])
    ends = np.r_[starts[1:], len(uid)]
    assert len(starts) == st["games"] and len(np.unique(uid)) == len(starts)
    for s, e in zip(starts, ends):
        b = rows["board"][s:e].reshape(e - s, -1)
        assert (b[0] == 0).all() and ((b > 0).sum(1) == np.arange(e - s)).all()
        assert rows["reward"][e - 1] == 1.0


def test_player_read_counts_skipped_games(eng):
    """Player.read keeps reading past skipped games and reports them in metrics['game_error']."""
    import torch
    from azalea_amd import AzaleaAgent, HexGame, Player, Policy
    n, sims = 7, 30
    cfg = dict(device="cuda", network="HexNetwork", board_size=n, num_blocks=1, base_chans=64,
               simulations=sims, search_batch_size=10, exploration_coef=0.5, exploration_depth=4,
               exploration_noise_alpha=0.03, exploration_noise_scale=0.25, exploration_temperature=1.0, seed=3)
    torch.manual_seed(0)
    policy = Policy()
    policy.initialize(cfg)
    policy.settings.update(move_sampling=True, move_exploration=True)
    agent = AzaleaAgent(lambda: HexGame(n), policy=policy, device="cuda")
    player = Player(None, [agent], n_games=32)
    eng_ = player.device_engine()
    assert eng_.cfg.game_index_stride == 1 and eng_.cfg.game_index_offset == 0
    frame, metrics = player.read(100)
    assert len(frame) >= 100 and metrics["games"] >= 1 and metrics.get("game_error", 0) == 0
    player.stop()


# ---- 13x13 / 19x256 against the C oracle ---------------------------------------------------------------
def test_config5_network_forward_vs_c_oracle(eng, orc):
    """BASELINE configs[4]'s network (13x13, 19 blocks x 256 channels) on two positions against the CPU
    oracle's fp32 direct convolution (network.py:42-61 is shape-generic): <= 1e-4 (twelve positions, from the empty
    board to a nearly full one)."""
    import torch
    from azalea_amd.network import HexNetwork
    torch.manual_seed(13019256)
    net = HexNetwork(board_size=13, num_blocks=19, base_chans=256).eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.6, 1.4)
    state = {k: v.detach().numpy() for k, v in net.state_dict().items()}
    rng = np.random.RandomState(5)
    boards, moves = [], []
    stone_counts = (20, 87, 0, 5, 33, 48, 61, 74, 95, 110, 127, 140)
    for stones in stone_counts:
        h = orc.Hex(13)
        while int((h.board > 0).sum()) < stones:
            lm = h.legal_moves()
            h2 = h.copy()
            h2.step(int(lm[rng.randint(len(lm))]))
            if h2.result:
                break
            h = h2
        b, lm = h.board, h.legal_moves()
        if h.color == 1:
            b, lm = orc.flip_board_moves(b, lm)
        boards.append(b)
        moves.append(lm)
    P = len(boards)
    K = max(len(m) for m in moves)
    lm = np.zeros((P, K), np.int32)
    for i, m in enumerate(moves):
        lm[i, :len(m)] = m
    boards = np.array(boards, np.int32)
    E = eng.Engine(board_size=13, n_games=2, simulations=10, search_batch_size=10,
                   evaluator=eng.EVAL_RESNET, num_blocks=19, base_chans=256)
    E.set_weights(state)
    value, logprob = E.forward(boards, lm)
    E.close()
    N = orc.Net(13, 19, 256, state)
    legal = lm > 0
    # two positions through the pinned parity path (1 s each on the host), all twelve -- from the empty board to a
    # nearly full one -- through the oracle's blocked forward, which a CPU test holds to the parity path at 1e-5
    ov, olp = N.forward(boards[:2], lm[:2])
    assert np.abs(value[:2] - ov).max() <= TOL and np.abs(logprob[:2] - olp)[legal[:2]].max() <= TOL
    fv, flp = N.forward(boards, lm, fast=True)
    assert np.abs(value - fv).max() <= TOL and np.abs(logprob - flp)[legal].max() <= TOL


def test_config5_workload_properties(eng):
    """BASELINE configs[4] as a workload on one GPU (13x13, 19x256 resnet, 800 sims -> 810 selects; 64
    concurrent games, one move): exact simulation counts, the root-visit checksum of every tree, priors
    that sum to one."""
    import torch
    from azalea_amd.network import HexNetwork
    n, G, sims = 13, 64, 800
    torch.manual_seed(0)
    net = HexNetwork(board_size=n, num_blocks=19, base_chans=256).eval()
    E = eng.Engine(board_size=n, n_games=G, simulations=sims, search_batch_size=10, evaluator=eng.EVAL_RESNET,
                   num_blocks=19, base_chans=256, seed=5)
    E.set_weights({k: v.detach().numpy() for k, v in net.state_dict().items() if v.dtype == torch.float32})
    per_move = (sims // 10 + 1) * 10
    st = E.play_steps(1)
    assert st["plies"] == G and st["selects"] == G * per_move
    assert 0 < st["evals"] <= st["selects"] + G and st["net_launches"] >= sims // 10 + 1
    E.search()
    root = E.get_root()
    for g in range(G):
        k = int(root["k"][g])
        assert k == n * n - 1
        cv = root["child_visits"][g, :k]
        s, rv = float(cv.sum()), float(root["root_visits"][g])
        assert s in (rv, rv - 1.0) and rv >= per_move and np.all(cv == np.round(cv)) and cv.min() >= 0
        pr = root["child_prior"][g, :k]
        assert abs(float(pr.sum()) - 1.0) < 1e-4 and (pr > 0).all()
    E.close()


def test_config5_full_leaf_batch_sampled_against_the_oracle(eng, orc):
    """BASELINE configs[4]'s shape at the bench leg's size (512 concurrent 13x13 games, 19x256): one leaf batch of
    ~4.9 k positions through the wide tower as the search launches it (38 layer launches, the batch split over two
    streams), ten of its rows -- first, last, either side of the split, a seeded sample -- against the oracle's fp32
    forward: value and every legal prior within 1e-4."""
    import torch
    from azalea_amd.network import HexNetwork
    n, G = 13, 512
    torch.manual_seed(7)
    net = HexNetwork(board_size=n, num_blocks=19, base_chans=256).eval()
    state = {k: v.detach().numpy() for k, v in net.state_dict().items()}
    E = eng.Engine(board_size=n, n_games=G, simulations=800, search_batch_size=10, evaluator=eng.EVAL_RESNET,
                   num_blocks=19, base_chans=256)
    assert "k_conv_wide_f16x3_s16" in E.kernel_info() and "AZX_WIDE_STREAMS=2" in E.kernel_info()
    E.set_weights({k: v for k, v in state.items() if v.dtype == np.float32})
    E.reset(moves=eng.random_prefixes(n, np.arange(G), 100, 4321))
    assert E.search_begin() == G
    cnt, done = E.search_step()
    assert not done and 0.8 * G * 10 <= cnt <= G * 10
    boards, lm, slot, k = E.get_leaves()
    value, prior = E.get_evals()
    rng = np.random.RandomState(2)
    half = cnt // 2
    rows = np.unique(np.r_[0, cnt - 1, half - 1, half, half + 1, rng.randint(0, cnt, 5)])
    fv, flp = orc.Net(n, 19, 256, state).forward(boards[rows], lm[rows], fast=True)
    assert np.abs(value[rows] - fv).max() <= 1e-4
    for i, r in enumerate(rows):
        kk = int(k[r])
        assert kk == int((lm[r] > 0).sum()) and kk > 0
        assert np.abs(prior[r, :kk] - np.exp(flp[i, :kk])).max() <= 1e-4
    while not done:
        cnt, done = E.search_step()
    E.close()


# ---- multi-GPU sharding by global game index ---------------------------------------------
def _games_by_uid(rows):
    uid = rows["game_uid"]
    starts = np.flatnonzero(np.r_[True, uid[1:] != uid[:-1]])
    ends = np.r_[starts[1:], len(uid)]
    return {int(uid[s]): (rows["board"][s:e].copy(), bits(rows["moves_prob"][s:e]).copy(), rows["reward"][s:e].copy())
            for s, e in zip(starts, ends)}


def test_global_game_index_same_games_for_any_world_size(eng):
    """Seeds come from the global game index: one engine with 8 slots and two engines ('ranks' 0 and 1 of
    2) with 4 slots each play, from the same base seed, the SAME games -- uid by uid, row by row, bit
    for bit (device noise and device move draws included)."""
    common = dict(board_size=7, simulations=30, search_batch_size=10, exploration_coef=0.5, exploration_depth=5,
                  noise_alpha=0.03, noise_scale=0.25, temperature=1.0, evaluator=eng.EVAL_UNIFORM, seed=4242)
    one = eng.Engine(n_games=8, **common)
    rows, _ = one.play(300)
    ref = _games_by_uid(rows)
    one.close()
    got = {}
    for rank in range(2):
        E = eng.Engine(n_games=4, game_index_stride=2, game_index_offset=rank, **common)
        rows, _ = E.play(150)
        games = _games_by_uid(rows)
        assert all(u % 2 == rank for u in games)           # rank r owns the indices r, r+2, ...
        got.update(games)
        E.close()
    both = sorted(set(ref) & set(got))
    assert len(both) >= 8                                   # the first generation at least
    for u in both:
        for a, b in zip(ref[u], got[u]):
            assert np.array_equal(a, b), u


# ---- parked slots ---------------------------------------------------------------------------------------
def test_parked_slots_hand_their_games_over_on_the_next_read(eng):
    """A finished game that finds the harvest queue full parks its slot; the next read drains them first
    (nothing is lost, nothing stays parked).  The queue bound is forced small through azx_debug_set_queue_cap."""
    n, G = 5, 32
    E = eng.Engine(board_size=n, n_games=G, simulations=20, search_batch_size=10, exploration_depth=3,
                   evaluator=eng.EVAL_UNIFORM, seed=9)
    E.debug_set_queue_cap(40)
    rows1, st1 = E.play(200, max_plies=40)                  # ~every slot finishes a game; the queue holds 40 rows
    E.debug_set_queue_cap(0)
    assert 0 < len(rows1["reward"]) <= 40
    c = E.debug_counters()
    finished_so_far = int(c[6])
    rows2, st2 = E.play(1, max_plies=1)                      # parked games come out first
    games2 = _games_by_uid(rows2)
    assert len(games2) >= 3
    for u, (b, _, rw) in games2.items():
        assert (b[0] == 0).all() and rw[-1] == 1.0
    assert not set(_games_by_uid(rows1)) & set(games2)
    rows3, st3 = E.play(300)                                 # and the pool keeps playing with all its slots
    assert st3["games"] >= G // 2 and int(E.debug_counters()[6]) > finished_so_far
    E.close()


# ---- start prefixes in play mode ------------------------------------------------------------------------
def test_play_mode_from_start_prefixes(eng):
    """azx_reset with move prefixes, then play mode (bench.py's de-synchronised pool): a game's rows start
    at its prefix, colours and rewards count plies from the empty board, sum_game_length counts them too."""
    n, G = 7, 16
    idx = np.arange(G)
    prefixes = eng.random_prefixes(n, idx, 20, seed=5)
    assert len({len(p) for p in prefixes}) > 4
    again = eng.random_prefixes(n, idx[::-1], 20, seed=5)[::-1]
    assert prefixes == again                                  # game i's prefix depends on (seed, i) only
    E = eng.Engine(board_size=n, n_games=G, simulations=20, search_batch_size=10, exploration_depth=3,
                   evaluator=eng.EVAL_UNIFORM, seed=12)
    E.reset(moves=prefixes)
    gm = E.get_games()
    assert np.array_equal(gm["ply"], [len(p) for p in prefixes]) and (gm["result"] == 0).all()
    rows, st = E.play(150)
    E.close()
    uid = rows["game_uid"]
    starts = np.flatnonzero(np.r_[True, uid[1:] != uid[:-1]])
    ends = np.r_[starts[1:], len(uid)]
    total_len, from_prefix = 0, 0
    for s, e in zip(starts, ends):
        b = rows["board"][s:e].reshape(e - s, -1)
        stones = (b > 0).sum(1)
        p0 = int(stones[0])
        from_prefix += p0 > 0
        assert np.array_equal(stones, p0 + np.arange(e - s))
        assert np.array_equal(rows["color"][s:e], (p0 + np.arange(e - s)) % 2)
        rw = rows["reward"][s:e]
        assert rw[-1] == 1.0 and np.array_equal(rw, np.where((np.arange(e - s) % 2) == ((e - s - 1) % 2), 1.0, -1.0))
        total_len += p0 + (e - s)
    assert from_prefix >= 3 and st["sum_game_length"] == total_len


# ---- device record exchange (the multi-GPU replay path on one GPU) ---------------------------------------
def test_device_records_roundtrip_into_the_ring(eng):
    """azx_play_device -> azx_rows_pack -> (all-gather of device tensors) -> azx_replay_put_records: the
    records hold exactly the rows azx_play would hand to the host (host twin: distributed.pack_rows), and
    the ring they are appended to collates like a ring fed through azx_replay_put."""
    import torch
    from azalea_amd import distributed as azd
    n, G = 7, 32
    common = dict(board_size=n, n_games=G, simulations=20, search_batch_size=10, exploration_depth=3,
                  evaluator=eng.EVAL_UNIFORM, seed=21)
    A = eng.Engine(**common)
    rows, _ = A.play(300)
    B = eng.Engine(**common)                                  # same seed: same games
    nrows, st = B.play_device(300)
    assert nrows == len(rows["reward"])
    rec = torch.empty((nrows, B.record_bytes), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    B.rows_pack(0, nrows, rec.data_ptr())
    host = rec.cpu().numpy()
    back = azd.unpack_rows(host, n)
    # the device pack kernel and its numpy twin produce the same bytes for the same rows
    assert np.array_equal(host, azd.pack_rows(back, n * n))
    # same seed, same games -- but the ORDER games are harvested in follows the GPU's scheduling, so the two
    # engines' queues are compared game by game
    def by_uid(r):
        order = np.argsort(r["game_uid"], kind="stable")
        return {k: v[order] for k, v in r.items()}
    a, b = by_uid(rows), by_uid(back)
    for k in rows:
        assert np.array_equal(b[k].reshape(a[k].shape), a[k]), k
    rows = back                                               # B's own order from here on
    # two 'ranks' worth of records into B's ring vs the same rows put from the host into A's ring
    cap = 2 * nrows + 7
    A.replay_create(cap)
    B.replay_create(cap)
    for _ in range(2):
        B.replay_put_records(nrows, rec.data_ptr())
        A.replay_put(rows["board"], rows["color"], rows["nlegal"], rows["moves_prob"], rows["reward"])
    assert A.replay_state() == B.replay_state()
    from azalea_amd.device_replay import DeviceReplayBuffer
    idx = np.random.RandomState(0).randint(0, 2 * nrows, 64)
    outs = []
    for E in (A, B):
        buf = DeviceReplayBuffer.__new__(DeviceReplayBuffer)
        buf.engine, buf.capacity, buf.device = E, cap, torch.device("cuda", 0)
        outs.append({k: v.cpu().numpy() for k, v in buf.sample(idx).items()})
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k
    A.close()
    B.close()


# ---- per-game metrics -----------------------------------------------------------------------------------
def test_player_metrics_are_sums_of_per_game_means(eng):
    """play_game averages a game's search metrics over ITS plies and Player.read sums those means over the
    games it returns (play_game.py:41-43, :73-76; parallel_player.py:50-51).  The engine hands the per-ply
    values over with the rows (azx_play_row_metrics): their per-game means match the engine's own tallies
    and bound checks (root width <= legal moves, log-probabilities <= 0, one game-start mark per game)."""
    n, G = 7, 32
    E = eng.Engine(board_size=n, n_games=G, simulations=30, search_batch_size=10, exploration_depth=4,
                   evaluator=eng.EVAL_UNIFORM, seed=3)
    rows, st = E.play(400)
    m = E.play_row_metrics()
    P = len(rows["reward"])
    assert m.shape == (P, 8)
    uid = rows["game_uid"]
    starts = np.flatnonzero(np.r_[True, uid[1:] != uid[:-1]])
    assert np.array_equal(np.flatnonzero(m[:, 3] > 0.5), starts)
    assert (m[:, 1] >= 1).all() and (m[:, 1] <= rows["nlegal"]).all() and (m[:, 1] == np.round(m[:, 1])).all()
    assert (m[:, 2] <= 1e-6).all() and np.isfinite(m).all()
    # search_tree.py:109-112: children = legal moves; mean child visits * children = an integer number of
    # backed-up visits, at least one per root child that was visited; nodes ever allocated >= 1 + children and
    # never fewer than the ply before in the same game (moves re-root, nothing is reclaimed)
    k = rows["nlegal"].astype(np.float64)
    assert np.array_equal(m[:, 6], rows["nlegal"].astype(np.float32))
    tot = m[:, 4].astype(np.float64) * k
    assert np.allclose(tot, np.round(tot), atol=1e-3) and (np.round(tot) >= m[:, 1]).all()
    assert (m[:, 5] >= 1 + k).all() and (m[:, 5] == np.round(m[:, 5])).all()
    ends = np.r_[starts[1:], P]
    for s_, e_ in zip(starts, ends):
        assert (np.diff(m[s_:e_, 5]) > 0).all()
    # the rows of finished games are a subset of all plies the call played: sums cannot exceed the call's tallies
    assert m[:, 1].sum() <= st["sum_root_width"] + 1e-3
    sums = E.game_metric_sums()
    names = [name for name, _ in E.ROW_METRIC_COLUMNS]
    cols = [c for _, c in E.ROW_METRIC_COLUMNS]
    want = np.zeros(len(cols))
    for s_, e_ in zip(starts, ends):
        want += m[s_:e_][:, cols].astype(np.float64).mean(0)
    assert sorted(sums) == sorted(names)
    assert np.allclose([sums[nm] for nm in names], want)
    E.close()


def test_throughput_and_parity_mode_report_the_same_metric_keys(eng):
    """The dict Player.read sums in throughput mode (device rows + azx_play_row_metrics) has exactly the keys
    play_game produces in parity mode (play_game.py:69-76 over search_tree.py:109-112, mcts.py:291,
    policy.py:164), and on a fresh tree both modes count the same nodes: 1 + k + the expansions' children."""
    import torch
    from azalea_amd import AzaleaAgent, HexGame, Player, Policy
    from azalea_amd.play_game import play_game
    n, sims = 5, 20
    cfg = dict(device="cuda", network="HexNetwork", board_size=n, num_blocks=1, base_chans=64,
               simulations=sims, search_batch_size=10, exploration_coef=0.5, exploration_depth=4,
               exploration_noise_alpha=0.03, exploration_noise_scale=0.25, exploration_temperature=1.0, seed=3)
    torch.manual_seed(0)
    policy = Policy()
    policy.initialize(cfg)
    policy.settings.update(move_sampling=True, move_exploration=True)
    agent = AzaleaAgent(lambda: HexGame(n), policy=policy, device="cuda")
    agent.seed(11)
    _, _, parity = play_game([agent], collect_data=False)
    player = Player(None, [agent], n_games=16)
    frame, thru = player.read(60)
    player.stop()
    assert set(thru) == set(parity), (sorted(thru), sorted(parity))
    g = thru["games"]
    assert g >= 1
    # per-game means are of the same magnitude in both modes (same search settings, same network)
    for key in ("search_root_children", "search_root_visits", "search_tree_nodes", "search_root_width"):
        assert 0.3 * parity[key] < thru[key] / g < 3.0 * parity[key], (key, thru[key] / g, parity[key])


def test_engine_first_then_torch_cuda_in_one_process():
    """Loading the engine library and running a search BEFORE anything touched torch.cuda used to leave torch unable
    to bring the GPU up afterwards ("No HIP GPUs are available": two HIP runtimes, ours first).  _lib.lib() now
    initialises torch's runtime first; a fresh process that uses the engine and only then torch must work."""
    import subprocess
    import sys
    code = ("import numpy as np\n"
            "from azalea_amd import engine as e\n"
            "E = e.Engine(board_size=5, n_games=2, simulations=10, evaluator=e.EVAL_UNIFORM)\n"
            "E.search(); k = int(E.get_root()['k'][0]); E.close()\n"
            "import torch\n"
            "x = torch.arange(4, device='cuda', dtype=torch.float32)\n"
            "print('OK', k, float(x.sum()))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK 25 6.0" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
