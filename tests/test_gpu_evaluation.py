"""Round-robin evaluation against the reference's recorded tournament (golden G10:
azalea/evaluation.py on three stub-net agents, three rounds): the same coin flips, the same
re-seeding, the same games -- per-game outcomes and final tallies."""
import os

import numpy as np
import pytest

from test_gpu_policy_parity import StubNet

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def stub_agents(z):
    from azalea_amd.azalea_agent import AzaleaAgent
    from azalea_amd.game.hex import HexGame
    from azalea_amd.policy import Policy
    n = int(z["board"])
    agents = []
    for mode, sims, c, sampling, explore in zip(z["spec_mode"], z["spec_sims"], z["spec_c"], z["spec_sampling"],
                                                z["spec_explore"]):
        p = Policy()
        p.net = StubNet(str(mode))
        p.network_type, p.board_size, p.num_blocks, p.base_chans = "stub", n, 0, 0
        p.simulations, p.search_batch_size, p.exploration_coef = int(sims), 10, float(c)
        p.exploration_depth, p.exploration_noise_alpha = 4, 0.3
        p.exploration_noise_scale, p.exploration_temperature = 0.25, 1.0
        p.settings["move_sampling"] = bool(sampling)
        p.settings["move_exploration"] = bool(explore)
        agents.append(AzaleaAgent(lambda n=n: HexGame(n), policy=p, device="cpu"))
    return agents


def test_round_robin_matches_reference():
    from azalea_amd import evaluation
    z = np.load(os.path.join(GOLDEN, "g10_tournament.npz"))
    agents = stub_agents(z)
    played = []
    real_worker = evaluation.worker

    def logging_worker(pair, ags, seed):
        pair2, outcome = real_worker(pair, ags, seed)
        played.append((pair[0], pair[1], seed, int(outcome)))
        return pair2, outcome
    evaluation.worker = logging_worker
    try:
        outcomes = evaluation.evaluate(agents, 3)
    finally:
        evaluation.worker = real_worker
    want = [(int(g[0]), int(g[1]), int(g[2]), int(g[4])) for g in z["games"]]
    assert played == want
    for pair, tally in zip(z["pairs"], z["tallies"]):
        assert [int(x) for x in outcomes[tuple(int(v) for v in pair)]] == [int(x) for x in tally]
    assert evaluation.gen_pairs(4) == [(0, 1), (0, 2), (1, 2), (0, 3), (1, 3), (2, 3)]


def net_agents(n=5):
    import torch
    from azalea_amd.azalea_agent import AzaleaAgent
    from azalea_amd.game.hex import HexGame
    from azalea_amd.policy import Policy
    agents = []
    for seed, sims, c, sampling, explore in ((1, 20, 0.5, True, True), (2, 30, 0.75, True, False), (3, 40, 1.0, False, False)):
        torch.manual_seed(seed)
        p = Policy()
        p.initialize(dict(device="cuda:0", network="HexNetwork", board_size=n, num_blocks=1, base_chans=64,
                          simulations=sims, search_batch_size=10, exploration_coef=c, exploration_depth=4,
                          exploration_noise_alpha=0.3, exploration_noise_scale=0.25, exploration_temperature=1.0))
        for m in p.net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.3, 1.7)
        p.settings["move_sampling"] = sampling
        p.settings["move_exploration"] = explore
        agents.append(AzaleaAgent(lambda n=n: HexGame(n), policy=p, device="cuda:0"))
    return agents


@pytest.mark.parametrize("n", [5, 3, 8])
def test_batched_tournament_plays_the_same_games(n):
    """All games of the round robin resident on the GPU at once (per-agent engines, only the slots
    to move are searched) give the tallies of the one-game-at-a-time schedule: same seeds, same
    per-game random streams, same network outputs per board whatever the batch."""
    from azalea_amd import evaluation
    agents = net_agents(n)
    seq = evaluation.evaluate(agents, 4)
    bat = evaluation.evaluate_batched(agents, 4)
    assert set(seq) == set(bat) == set(evaluation.gen_pairs(3))
    for pair in seq:
        assert list(map(int, seq[pair])) == list(map(int, bat[pair])), pair
    assert sum(sum(v) for v in bat.values()) == 12
