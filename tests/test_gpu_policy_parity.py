"""The drop-in surface on the GPU: azalea_amd.Policy / AzaleaAgent / play_game / Player driving the
HIP engine must reproduce the reference's recorded self-play games (golden G5: same seeds, stub
networks) bit for bit -- boards, move distributions, rewards, search metrics."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


class StubNet:
    """Same duck-typed network as tests/golden/make_golden.py (uniform / hashed priors + value)."""
    device = torch.device("cpu")

    def __init__(self, mode):
        from oracle import oracle as orc
        self.mode, self.fnv = mode, orc.fnv1a

    def eval(self):
        return self

    def run(self, batch, compute_loss=False):
        board = batch["board"].numpy().astype(np.int32)
        lm = batch["legal_moves"].numpy()
        B, K = lm.shape
        value = np.zeros(B, np.float32)
        logit = np.zeros((B, K), np.float32)
        for i in range(B):
            h = self.fnv(board[i].ravel())
            if self.mode != "uniform0":
                value[i] = np.float32((h & 0xFFFF) / 32768.0 - 1.0)
            if self.mode == "hashprior":
                for j in range(K):
                    t = int(lm[i, j])
                    if t:
                        x = (h ^ (t * 2654435761)) & 0xFFFFFFFF
                        x = (x * 2246822519) & 0xFFFFFFFF
                        logit[i, j] = np.float32(((x >> 13) & 0xFF) / 64.0)
        lt = torch.tensor(logit)
        lt.masked_fill_(torch.tensor(lm == 0), -99)
        return dict(value=torch.tensor(value), moves_logprob=torch.log_softmax(lt, dim=1))


def make_policy(z):
    from azalea_amd import Policy
    p = Policy()
    p.net = StubNet(str(z["mode"]))
    p.network_type, p.board_size, p.num_blocks, p.base_chans = "stub", int(z["cfg_n"]), 0, 0
    p.simulations = int(z["cfg_sims"])
    p.search_batch_size = int(z["cfg_batch"])
    p.exploration_coef = float(z["cfg_c"])
    p.exploration_depth = int(z["cfg_depth"])
    p.exploration_noise_alpha = float(z["cfg_alpha"])
    p.exploration_noise_scale = float(z["cfg_eps"])
    p.exploration_temperature = float(z["cfg_temp"])
    p.settings["move_sampling"] = bool(z["cfg_sampling"])
    p.settings["move_exploration"] = bool(z["cfg_explore"])
    return p


G5 = sorted(glob.glob(os.path.join(GOLDEN, "g5_game_*.npz")))


@pytest.mark.parametrize("path", G5, ids=[os.path.basename(p)[8:-4] for p in G5])
def test_g5_policy_agent_play_game_trace(path):
    from azalea_amd import AzaleaAgent, HexGame
    from azalea_amd.play_game import play_game
    z = np.load(path)
    n = int(z["cfg_n"])
    agent = AzaleaAgent(lambda: HexGame(n), policy=make_policy(z), device="cpu")
    agent.seed(int(z["cfg_seed"]))
    result, frame, metrics = play_game([agent], collect_data=True)
    assert result == int(z["result"]) and len(frame) == len(z["board"])
    for i in range(len(frame)):
        k = int(z["nlegal"][i])
        st = frame.state[i]
        assert np.array_equal(st.board, z["board"][i]) and st.color == z["color"][i] and st.result == 0
        assert np.array_equal(st.legal_moves, z["legal_moves"][i, :k])
        assert frame.moves_prob[i].dtype == np.float32
        assert np.array_equal(bits(frame.moves_prob[i]), bits(z["moves_prob"][i, :k])), i
    assert np.array_equal(np.array(frame.reward, np.float32), z["reward"])
    want = dict(zip([str(s) for s in z["metric_names"]], z["metric_values"]))
    for name, v in want.items():
        assert abs(float(metrics[name]) - v) <= 1e-5 * max(1.0, abs(v)), name


def test_device_net_policy_is_deterministic_and_player_reads_whole_games():
    from azalea_amd import AzaleaAgent, HexGame, Player, Policy
    from azalea_amd.play_game import play_game
    from azalea_amd.prep import torch_batch_replays
    cfg = dict(device="cuda", network="HexNetwork", board_size=7, num_blocks=2, base_chans=64,
               simulations=30, search_batch_size=10, exploration_coef=0.5, exploration_depth=5,
               exploration_noise_alpha=0.03, exploration_noise_scale=0.25,
               exploration_temperature=1.0, seed=11)
    torch.manual_seed(0)
    policy = Policy()
    policy.initialize(cfg)
    policy.settings.update(move_sampling=True, move_exploration=True)
    agent = AzaleaAgent(lambda: HexGame(7), policy=policy, device="cuda")
    games = []
    for _ in range(2):
        agent.seed(99)
        r, frame, m = play_game([agent], collect_data=True)
        games.append((r, [s.board.copy() for s in frame.state], [p.copy() for p in frame.moves_prob]))
    assert games[0][0] == games[1][0] and len(games[0][1]) == len(games[1][1])
    for a, b in zip(games[0][2], games[1][2]):
        assert np.array_equal(bits(a), bits(b))
    # throughput mode behind Player.read: whole games, trainer-ready rows
    player = Player(None, [agent], n_games=64)
    frame, metrics = player.read(150)
    assert len(frame) >= 150 and metrics["games"] >= 1 and metrics["moves_per_game"] == len(frame)
    batch = torch_batch_replays([frame[i] for i in range(16)])
    out, loss = policy.net.run({k: v.to("cuda") for k, v in batch.items()}, compute_loss=True)
    assert np.isfinite(loss.item())
    frame2, _ = player.read(10)
    assert len(frame2) >= 10
    player.stop()


def test_trainer_style_loop_consumes_the_engine():
    """What azalea/policy_trainer.py:45-90 does with these classes: seed a ReplayBuffer from the
    random mover, DataLoader + torch_batch_replays collate, SGD step on the live network,
    replaybuf.consume(...) pulling fresh self-play (device network, throughput mode) from Player."""
    from torch import optim
    from torch.utils.data import DataLoader
    from azalea_amd import AzaleaAgent, HexGame, Player, Policy, ReplayBuffer
    from azalea_amd.prep import torch_batch_replays
    cfg = dict(device="cuda", network="HexNetwork", board_size=5, num_blocks=1, base_chans=64,
               simulations=20, search_batch_size=10, exploration_coef=0.5, exploration_depth=3,
               exploration_noise_alpha=0.03, exploration_noise_scale=0.25,
               exploration_temperature=1.0, seed=5)
    torch.manual_seed(1)
    policy = Policy()
    policy.initialize(cfg)
    game_factory = lambda: HexGame(5)   # noqa: E731
    seed_player = Player(None, [AzaleaAgent(game_factory)])
    examples, metrics = seed_player.read(200)
    seed_player.stop()
    assert metrics["games"] >= 1
    buf = ReplayBuffer(examples)
    size0 = len(buf)
    loader = DataLoader(buf, batch_size=32, shuffle=True, collate_fn=torch_batch_replays)
    opt = optim.SGD(policy.net.parameters(), lr=0.01, momentum=0.9)
    policy.net.train()
    policy.settings.update(move_sampling=True, move_exploration=True)
    player = Player(None, [AzaleaAgent(game_factory, policy=policy, device="cuda")], n_games=32)
    steps, fresh_games = 0, 0
    for batch in loader:
        batch = {k: v.to("cuda") for k, v in batch.items()}
        opt.zero_grad()
        out, loss = policy.net.run(batch, compute_loss=True)
        loss.backward()
        opt.step()
        m = buf.consume(32 / 4, player)            # oversampling 4, policy_trainer.py:90
        fresh_games += m.get("games", 0)
        assert np.isfinite(loss.item())
        steps += 1
        if steps == 6:
            break
    player.stop()
    assert fresh_games >= 1 and len(buf) == size0   # FIFO keeps its size; new rows overwrote old


@pytest.mark.parametrize("n,mode,sims,sampling", [(2, "hashprior", 20, True), (3, "uniformhash", 30, True),
                                                  (4, "hashprior", 30, False), (6, "hashprior", 40, True),
                                                  (8, "uniformhash", 40, True), (10, "hashprior", 30, True),
                                                  (12, "uniformhash", 20, True), (13, "hashprior", 20, True)])
def test_whole_games_on_the_other_board_sizes_match_the_oracle(n, mode, sims, sampling):
    """G5 holds the reference's games on 5x5 / 7x7 / 9x9 / 11x11; here the same surface (Policy -> AzaleaAgent ->
    play_game, stub network on the host, tree and rules on the GPU, numpy RNG) plays whole games on every other size
    and the oracle's play_game -- pinned to the reference by G5 and G11 -- must produce the same game bit for bit."""
    from test_oracle_golden import _stub_eval
    from azalea_amd import AzaleaAgent, HexGame
    from azalea_amd.play_game import play_game
    from oracle import oracle as orc
    cfg = dict(mode=mode, cfg_n=n, cfg_sims=sims, cfg_batch=10, cfg_c=0.5, cfg_depth=min(6, n), cfg_alpha=0.3, cfg_eps=0.25,
               cfg_temp=1.0, cfg_sampling=sampling, cfg_explore=sampling)
    seed = 1000 + n
    agent = AzaleaAgent(lambda: HexGame(n), policy=make_policy(cfg), device="cpu")
    agent.seed(seed)
    result, frame, metrics = play_game([agent], collect_data=True)
    o_result, rows, reward = orc.play_game(n, _stub_eval(mode), simulations=sims, batch_size=10, c_puct=0.5,
                                           exploration_depth=cfg["cfg_depth"], noise_alpha=0.3, noise_scale=0.25,
                                           temperature=1.0, seed=seed, move_sampling=sampling, move_exploration=sampling)
    assert result == o_result and len(frame) == len(rows)
    for i, r in enumerate(rows):
        st = frame.state[i]
        assert np.array_equal(st.board, r["board"]) and st.color == r["color"], i
        assert np.array_equal(st.legal_moves, r["legal_moves"]), i
        assert np.array_equal(bits(frame.moves_prob[i]), bits(r["moves_prob"])), i
    assert np.array_equal(np.array(frame.reward, np.float32), reward)
    assert metrics["moves_per_game"] == len(rows)
    assert abs(metrics["search_tree_nodes"] - np.mean([r["num_nodes"] for r in rows])) < 1e-6
