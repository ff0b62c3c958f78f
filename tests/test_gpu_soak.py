"""A short soak of the product surface (tests/soak_selfplay.py): Player.read frames and DeviceReplayBuffer
refills interleaved with optimizer steps, every returned game replayed move by move through the CPU oracle's
rules.  AZX_SOAK_SECONDS lengthens it (profiles/r2_soak.json is a 240 s run at configs[2])."""
import os

import pytest

pytestmark = pytest.mark.gpu


def test_short_selfplay_soak():
    import soak_selfplay
    out = soak_selfplay.run(float(os.environ.get("AZX_SOAK_SECONDS", "12")), 512, sims=100, read_size=1500)
    assert out["games_checked_move_by_move"] >= 10 and out["device_refills"] >= 1 and out["game_errors"] == 0
