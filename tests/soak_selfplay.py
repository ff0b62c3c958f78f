#!/usr/bin/env python3
"""Soak of the product surface (test infrastructure: it uses the CPU oracle as the checker, so it lives under
tests/): configs[2] self-play for a while --
Player.read frames and DeviceReplayBuffer.consume refills interleaved with optimizer steps that change the
weights (eager ones on the frames, hand-written ones on the HBM ring) -- with every returned game replayed through the CPU oracle's rules (tests' checker; never the
thing measured): each row is the position reached by the moves before it, the stone a row adds is a legal
move, the game is not over before its last row and the last mover has a winning move there.

    python tests/soak_selfplay.py [seconds] [games]        (profiles/r2_soak.json: 240 s, 4096 games)
tests/test_gpu_soak.py runs a short one."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from azalea_amd import AzaleaAgent, HexGame, Player, Policy
from azalea_amd.device_replay import DeviceReplayBuffer
from azalea_amd.policy_trainer import supervised_step
from azalea_amd.prep import torch_batch_replays
from oracle import oracle as orc


def run(budget=180.0, G=4096, sims=400, read_size=3000):
    n = 11
    cfg = dict(device="cuda", network="HexNetwork", board_size=n, num_blocks=6, base_chans=64, simulations=sims,
               search_batch_size=10, exploration_coef=0.5, exploration_depth=15, exploration_noise_alpha=0.03,
               exploration_noise_scale=0.25, exploration_temperature=1.0, seed=7)
    torch.manual_seed(0)
    policy = Policy()
    policy.initialize(cfg)
    policy.settings.update(move_sampling=True, move_exploration=True)
    agent = AzaleaAgent(lambda: HexGame(n), policy=policy, device="cuda")
    player = Player(None, [agent], n_games=G)
    opt = torch.optim.SGD(policy.net.parameters(), lr=1e-3, momentum=0.9)


    def check_games(frame):
        """every game of a frame replayed through the oracle's rules; returns (#games, #rows)"""
        games, start = 0, 0
        P = len(frame)
        stones = np.array([int((s.board > 0).sum()) for s in frame.state])
        bounds = [i for i in range(P) if stones[i] == 0] + [P]
        for a, b in zip(bounds[:-1], bounds[1:]):
            h = orc.Hex(n)
            for i in range(a, b):
                st = frame.state[i]
                assert np.array_equal(h.board, st.board) and h.result == 0 and st.color == (i - a) % 2
                assert np.array_equal(h.legal_moves(), st.legal_moves)
                p = frame.moves_prob[i]
                assert len(p) == len(st.legal_moves) and abs(float(p.sum()) - 1.0) < 1e-5 and (p >= 0).all()
                if i + 1 < b:
                    diff = np.flatnonzero(frame.state[i + 1].board.ravel() != st.board.ravel())
                    assert len(diff) == 1
                    h.step(int(diff[0]) + 1)
            wins = 0
            for mv in h.legal_moves():
                h2 = h.copy()
                h2.step(int(mv))
                wins += h2.result != 0
            assert wins >= 1 and frame.reward[b - 1] == 1.0
            rw = np.array(frame.reward[a:b])
            assert np.array_equal(rw, np.where((np.arange(b - a) % 2) == ((b - a - 1) % 2), 1.0, -1.0))
            games += 1
        return games, P


    t0 = time.time()
    tot_games = tot_rows = reads = refills = steps = 0
    errors = 0.0
    buf = None
    nstep = None
    native_steps = 0
    lengths = []
    while time.time() - t0 < budget:
        frame, m = player.read(read_size)
        g, p = check_games(frame)
        assert g == m["games"] and p == m["moves_per_game"] and np.isfinite(list(m.values())).all()
        errors += m.get("game_error", 0)
        tot_games += g; tot_rows += p; reads += 1
        lengths.append(p / g)
        # a few optimizer steps on the fresh rows: the next read must pick the new weights up
        policy.net.train()
        for k in range(2):
            idx = np.random.RandomState(steps).randint(0, len(frame), 64)
            batch = torch_batch_replays([frame[int(i)] for i in idx])
            out, loss = supervised_step(policy.net, batch, train=True, optimizer=opt, device="cuda")
            assert np.isfinite(loss)
            steps += 1
        policy.net.eval()
        if buf is None:
            buf = DeviceReplayBuffer(player.device_engine(), capacity=60000)
        buf.fresh_counter = 0
        mm = buf.consume(float(read_size) * 2 / 3, player)
        refills += 1
        assert mm["games"] >= 1 and np.isfinite(list(mm.values())).all()
        # the hand-written training step on the ring (collated on its stream, nothing synchronised): the next read and
        # the next refill play with what it left in the module
        if nstep is None:
            from azalea_amd.native_train import NativeTrainStep
            nstep = NativeTrainStep(policy.net, opt, 128, "cuda")
        policy.net.train()
        for k in range(16):
            l3, _ = nstep.step_from_ring(buf, np.random.RandomState(1000 + native_steps).randint(0, len(buf), 128))
            native_steps += 1
        assert np.isfinite(l3.cpu().numpy()).all()
        policy.net.eval()
        rows = buf.rows(np.random.RandomState(refills).randint(0, len(buf), 512))
        k = (rows["board"].reshape(512, -1) == 0).sum(1)
        assert np.array_equal(k, (rows["legal_moves"] > 0).sum(1)) and np.abs(rows["moves_prob"].sum(1) - 1).max() < 1e-5
        assert set(np.unique(rows["reward"])) <= {-1.0, 1.0}
    E = player.device_engine()
    c = E.debug_counters()
    out = {"seconds": time.time() - t0, "reads": reads, "games_checked_move_by_move": tot_games, "rows": tot_rows,
           "device_refills": refills, "ring_rows": len(buf), "optimizer_steps": steps, "native_train_steps": native_steps,
           "game_errors": errors,
           "mean_game_length": float(np.mean(lengths)), "engine_games_finished": int(c[6]), "engine_plies": int(c[8]),
           "engine_selects": int(c[0])}
    assert out["engine_selects"] == out["engine_plies"] * (sims // 10 + 1) * 10
    player.stop()
    return out


if __name__ == "__main__":
    print(json.dumps(run(float(sys.argv[1]) if len(sys.argv) > 1 else 180.0,
                         int(sys.argv[2]) if len(sys.argv) > 2 else 4096)))
