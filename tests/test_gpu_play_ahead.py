"""Self-play beside training on one GPU (azalea_amd/play_ahead.py): the device side.

* `azx_reserve_cus` moves the engine's streams onto a CU mask and back: the games are the same bit for bit (a mask changes
  where blocks run, not what they compute), the kernel description reports the reservation;
* `azx_replay_put_records_async` on a caller's stream leaves the ring exactly as the blocking put does;
* the play-ahead loop end to end at a small size: `DeviceReplayBuffer.consume` fed by the play thread while
  NativeTrainStep steps run on a high-priority stream -- rows arrive, weights are refreshed, the loss is finite, and the
  rows taken are whole games in harvest order.
(The thread protocol itself is tests/test_play_ahead.py, CPU; that training in this mode makes a stronger player is
tests/test_gpu_train_strength.py.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _engine(n=7, games=64, sims=40, **kw):
    from azalea_amd import engine as eng
    return eng.Engine(board_size=n, n_games=games, simulations=sims, search_batch_size=10, exploration_coef=0.5,
                      exploration_depth=6, noise_alpha=0.3, noise_scale=0.25, temperature=1.0,
                      evaluator=kw.pop("evaluator", eng.EVAL_UNIFORM_HASH), seed=77, **kw)


@pytest.mark.parametrize("kind", ["uniform", "resnet"])
def test_reserved_cus_change_where_the_engine_runs_not_what_it_plays(kind):
    from azalea_amd import engine as eng
    from azalea_amd.network import HexNetwork
    rows = []
    for reserve in (0, 4, 0):
        kw = dict(evaluator=eng.EVAL_RESNET, num_blocks=2, base_chans=64) if kind == "resnet" else {}
        E = _engine(**kw)
        if kind == "resnet":
            torch.manual_seed(5)
            net = HexNetwork(board_size=7, num_blocks=2, base_chans=64).eval()
            E.set_weights({k: v.detach().numpy() for k, v in net.state_dict().items()})
        if reserve:
            assert E.reserve_cus(1) == 32 and "reserved_cus=32" in E.kernel_info()      # rounded up to one per shader engine
            assert E.reserve_cus(0) == 0 and "reserved_cus=0" in E.kernel_info()
            assert E.reserve_cus(reserve) == 8 * reserve
        r, st = E.play(64 * 49, max_plies=30)
        assert st["game_errors"] == 0 and st["plies"] == 64 * 30
        # whole games are appended to the queue in the order their waves finish a move -- timing; a game's own rows are
        # contiguous and in ply order.  Compare game by game.
        order = np.argsort(r["game_uid"], kind="stable")
        rows.append({k: v[order] for k, v in r.items()})
        E.close()
    assert len(np.unique(rows[0]["game_uid"])) >= 4 and len(rows[0]["reward"]) >= 100
    for k in rows[0]:
        assert np.array_equal(rows[0][k], rows[1][k]), k
        assert np.array_equal(rows[0][k], rows[2][k]), k


def test_reserve_cus_rejects_what_it_cannot_give():
    from azalea_amd._lib import AzxError
    E = _engine()
    with pytest.raises(AzxError):
        E.reserve_cus(32)
    with pytest.raises(AzxError):
        E.reserve_cus(-1)
    E.close()


def test_async_ring_put_on_a_caller_stream_equals_the_blocking_put():
    from azalea_amd.device_replay import DeviceReplayBuffer
    E = _engine()
    n, _ = E.play_device(300)
    rec = torch.empty((n, E.record_bytes), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    E.rows_pack(0, n, rec.data_ptr())
    held = []
    for use_async in (False, True):
        buf = DeviceReplayBuffer(E, capacity=n - 37, shared=False)      # smaller than the rows: the ring wraps
        if use_async:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                half = n // 2
                E.replay_put_records_async(half, rec.data_ptr(), s.cuda_stream)
                E.replay_put_records_async(n - half, rec.data_ptr() + half * E.record_bytes, s.cuda_stream)
            s.synchronize()
        else:
            E.replay_put_records(n, rec.data_ptr())
        assert len(buf) == n - 37 and buf.write_idx == n % (n - 37)
        held.append(buf.rows())
    for k in held[0]:
        assert np.array_equal(held[0][k], held[1][k]), k
    E.close()


def test_play_ahead_feeds_the_ring_while_the_native_step_trains():
    from torch import optim
    from azalea_amd import AzaleaAgent, HexGame, Player, Policy
    from azalea_amd.device_replay import DeviceReplayBuffer
    from azalea_amd.native_train import NativeTrainStep
    from azalea_amd.play_ahead import PlayAhead
    dev, n, B = "cuda:0", 7, 64
    torch.manual_seed(0)
    policy = Policy()
    policy.initialize(dict(device=dev, network="HexNetwork", board_size=n, num_blocks=2, base_chans=32, simulations=40,
                           search_batch_size=10, exploration_coef=0.5, exploration_depth=6, exploration_noise_alpha=0.3,
                           exploration_noise_scale=0.25, exploration_temperature=1.0, seed=1))
    policy.net.to(dev).train()
    policy.settings.update(move_sampling=True, move_exploration=True)
    player = Player(None, [AzaleaAgent(lambda: HexGame(n), policy=policy, device=dev)], n_games=256, gather=False)
    E = player.device_engine()
    buf = DeviceReplayBuffer(E, 20000, shared=False)
    buf.consume(4000, player)                        # inline fill first (the deterministic mode)
    buf.fresh_counter = 0
    size0 = len(buf)
    opt = optim.SGD(policy.net.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
    gs = NativeTrainStep(policy.net, opt, B, torch.device(dev))
    ahead = PlayAhead(player, E, ahead_rows=512, weight_sync_steps=10, reserve_cus=4)
    buf.ahead = ahead
    ahead.start()
    assert "reserved_cus=32" in E.kernel_info()
    side = torch.cuda.Stream(dev, priority=-1)
    side.wait_stream(torch.cuda.current_stream())
    rng = np.random.RandomState(0)
    rows_taken, losses = 0, []
    with torch.cuda.stream(side):
        for i in range(400):
            l3, _ = gs.step_from_ring(buf, rng.randint(0, len(buf), B))
            ahead.after_step()
            m = buf.consume(B / 4.0, player)
            if m:
                rows_taken += int(m["moves_per_game"])
                assert m["games"] >= 1 and m["game_error"] == 0
            if i % 100 == 99:
                losses.append(float(l3[0].item()))
    torch.cuda.synchronize()
    ahead.stop()
    buf.ahead = None
    c = ahead.counters()
    assert "reserved_cus=0" in E.kernel_info()       # the engine has its CUs back
    assert rows_taken >= 400 * B / 4.0 - B and len(buf) == min(20000, size0 + rows_taken)
    assert c["productions"] >= 10 and c["weight_syncs"] >= 5 and c["rows"] >= rows_taken
    assert c["max_backlog_rows"] < 512 + 256 * 49    # the bound (+ at most one harvest)
    assert all(np.isfinite(x) for x in losses) and losses[-1] < losses[0] + 0.5
    # the rows the ring took are whole games of the engine: every board consistent with its colour to move
    rows = buf.rows(np.arange(size0, min(len(buf), size0 + 2000)))
    stones = (rows["board"].reshape(len(rows["color"]), -1) > 0).sum(1)
    assert np.array_equal(stones % 2, rows["color"])
    gs.close()
    player.stop()
