"""azalea_amd.network.HexNetwork (the weight-holding/training module) must be interchangeable
with the reference's: same state_dict layout, same forward (golden G3 vectors).  CPU only."""
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("tag", ["5_1x8", "13_2x32", "11_6x64"])
def test_module_matches_reference_golden(tag):
    from azalea_amd.network import HexNetwork
    z = np.load(os.path.join(GOLDEN, "g3_forward_%s.npz" % tag))
    n, blocks, chans = [int(x) for x in z["cfg"]]
    net = HexNetwork(board_size=n, num_blocks=blocks, base_chans=chans).eval()
    ref = {k[2:]: torch.tensor(z[k]) for k in z.files if k.startswith("w:")}
    own = net.state_dict()
    assert list(own.keys()) == list(ref.keys())          # same names, same order
    for k in own:
        assert tuple(own[k].shape) == tuple(ref[k].shape), k
    net.load_state_dict(ref)
    with torch.no_grad():
        out = net.run({"board": torch.tensor(z["board"]), "legal_moves": torch.tensor(z["legal_moves"])})
    legal = z["legal_moves"] > 0
    assert np.abs(out["value"].numpy() - z["value"]).max() <= 1e-5
    assert np.abs(out["moves_logprob"].numpy() - z["moves_logprob"])[legal].max() <= 1e-5


def test_module_training_step_runs():
    from azalea_amd.network import HexNetwork
    net = HexNetwork(board_size=5, num_blocks=1, base_chans=8)
    batch = {"board": torch.randint(0, 3, (4, 5, 5), dtype=torch.int32),
             "legal_moves": torch.arange(1, 26, dtype=torch.int32).repeat(4, 1),
             "moves_prob": torch.full((4, 25), 1 / 25.0), "reward": torch.ones(4)}
    out, loss = net.run(batch, compute_loss=True)
    loss.backward()
    assert np.isfinite(loss.item()) and "value_loss" in out and "moves_loss" in out
