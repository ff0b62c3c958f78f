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
