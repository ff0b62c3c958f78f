"""Checkpoint compatibility: the reference's shipped policy file through
azalea_amd.Policy.load must give what the reference's own Policy.load gives (golden G8: schema,
per-tensor digests, forward outputs), and checkpoints written here must load back.  CPU only; the
file itself is read from the reference tree when it is present (it is not on the GPU box)."""
import hashlib
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CKPT = os.environ.get("AZX_REF_CHECKPOINT", "/root/reference/models/hex11-20180712-3362.policy.pth")
HYPER_TYPES = {"network_type": str, "board_size": int, "num_blocks": int, "base_chans": int,
               "simulations": int, "search_batch_size": int, "exploration_depth": int}


def golden():
    return np.load(os.path.join(GOLDEN, "g8_checkpoint.npz"))


@pytest.mark.skipif(not os.path.exists(CKPT), reason="reference checkpoint not present on this machine")
def test_shipped_checkpoint_loads_like_the_reference():
    from azalea_amd.policy import Policy
    z = golden()
    assert hashlib.sha256(open(CKPT, "rb").read()).hexdigest() == str(z["file_sha256"])
    policy = Policy.load(CKPT, device="cpu")
    for name, want in zip(z["hyper_names"], z["hyper_values"]):
        got = getattr(policy, str(name))
        cast = HYPER_TYPES.get(str(name), float)
        assert cast(got) == cast(str(want)), name
    sd = policy.net.state_dict()
    assert sorted(sd) == list(z["tensor_names"])
    for name, want, shape in zip(z["tensor_names"], z["tensor_sha256"], z["tensor_shapes"]):
        t = sd[str(name)]
        assert str(tuple(t.shape)) == str(shape), name
        assert hashlib.sha256(np.ascontiguousarray(t.numpy()).tobytes()).hexdigest() == str(want), name
    assert not policy.net.training
    with torch.no_grad():
        o = policy.net.run({"board": torch.tensor(z["board"]), "legal_moves": torch.tensor(z["legal_moves"])})
    legal = z["legal_moves"] > 0
    assert np.abs(o["value"].numpy() - z["value"]).max() <= 1e-5
    assert np.abs(o["moves_logprob"].numpy() - z["moves_logprob"])[legal].max() <= 1e-5


def test_checkpoint_round_trip_keeps_the_reference_schema(tmp_path):
    """policy_trainer.py:161-181 writes {'policy': policy.state_dict(), ...}; policy.py:85-130 reads it."""
    from azalea_amd.policy import Policy
    z = golden()
    n, blocks, chans = [int(x) for x in z["cfg"]]
    p = Policy()
    p.initialize(dict(device="cpu", network="HexNetwork", board_size=n, num_blocks=blocks, base_chans=chans,
                      simulations=800, search_batch_size=10, exploration_coef=0.75, exploration_depth=15,
                      exploration_noise_alpha=0.03, exploration_noise_scale=0.25, exploration_temperature=1.0,
                      seed=5))
    p.net.load_state_dict({k[2:]: torch.tensor(z[k]) for k in z.files if k.startswith("w:")})
    state = p.state_dict()
    assert set(state) == {"net", "rng", *[str(k) for k in z["hyper_names"]]}
    path = str(tmp_path / "ckpt.policy.pth")
    torch.save({"policy": state}, path)
    q = Policy.load(path, device="cpu")
    for k in z["hyper_names"]:
        assert getattr(q, str(k)) == getattr(p, str(k)), k
    for k, v in p.net.state_dict().items():
        assert torch.equal(v, q.net.state_dict()[k]), k
    assert np.array_equal(q.rng.get_state()[1], p.rng.get_state()[1])
    with torch.no_grad():
        o = q.net.run({"board": torch.tensor(z["board"]), "legal_moves": torch.tensor(z["legal_moves"])})
    assert np.abs(o["value"].numpy() - z["value"]).max() <= 1e-5
