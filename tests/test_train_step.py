"""The trainer's inner step against the reference's recorded one (golden G9:
azalea/policy_trainer.py:123-142 on three fixed batches, SGD + momentum + weight decay,
train-mode BatchNorm): losses, outputs and the updated tensors."""
import os

import numpy as np
import pytest
import torch

from replay_golden import load_g7, source_frame

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def run_steps(device, tol_w):
    from azalea_amd.network import HexNetwork
    from azalea_amd.policy_trainer import supervised_step
    from azalea_amd.prep import torch_batch_replays
    z = np.load(os.path.join(GOLDEN, "g9_train_step.npz"))
    frame = source_frame(load_g7())
    net = HexNetwork(board_size=11, num_blocks=2, base_chans=16)
    net.load_state_dict({k[3:]: torch.tensor(z[k]) for k in z.files if k.startswith("w0:")})
    net.to(device)
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    for step, ids in enumerate(z["batch_idx"]):
        batch = torch_batch_replays([frame[int(i)] for i in ids])
        o, loss = supervised_step(net, batch, train=True, optimizer=opt, device=device)
        assert abs(loss - float(z["step%d_loss" % step])) <= tol_w * 10
        assert abs(o["value_loss"] - float(z["step%d_value_loss" % step])) <= tol_w * 10
        assert abs(o["moves_loss"] - float(z["step%d_moves_loss" % step])) <= tol_w * 10
        assert np.abs(o["value"].cpu().numpy() - z["step%d_value" % step]).max() <= tol_w * 10
        legal = batch["legal_moves"].cpu().numpy() > 0
        assert np.abs(o["moves_logprob"].cpu().numpy() - z["step%d_moves_logprob" % step])[legal].max() <= tol_w * 10
    for k, v in net.state_dict().items():
        want = z["w3:" + k]
        if want.dtype.kind == "f":
            assert np.abs(v.cpu().numpy() - want).max() <= tol_w, k
        else:
            assert np.array_equal(v.cpu().numpy(), want), k
    batch = torch_batch_replays([frame[int(i)] for i in z["batch_idx"][0]])
    o, loss = supervised_step(net, batch, train=False, device=device)
    assert not net.training
    assert abs(loss - float(z["eval_loss"])) <= tol_w * 10
    assert np.abs(o["value"].cpu().numpy() - z["eval_value"]).max() <= tol_w * 10


def test_graphed_step_forward_is_the_eager_forward():
    """The two things GraphedTrainStep changes about the forward pass, checked on the CPU: the embedding applied as
    masks times the embedding matrix equals nn.Embedding bit for bit, and padding legal_moves / moves_prob to all
    cells leaves every legal log-probability and both losses where they were (the padded logits are -99)."""
    from azalea_amd.network import HexNetwork
    from azalea_amd.policy_trainer import embed_by_masks
    from azalea_amd.prep import torch_batch_replays
    import torch.nn.functional as F
    z = np.load(os.path.join(GOLDEN, "g9_train_step.npz"))
    frame = source_frame(load_g7())
    net = HexNetwork(board_size=11, num_blocks=2, base_chans=16).eval()
    net.load_state_dict({k[3:]: torch.tensor(z[k]) for k in z.files if k.startswith("w0:")})
    batch = torch_batch_replays([frame[int(i)] for i in z["batch_idx"][0]])
    with torch.no_grad():
        want = net.encoder(batch["board"].long()).permute(0, 3, 1, 2).contiguous()
        got = embed_by_masks(net, batch["board"])
        assert torch.equal(want, got)
        B, k = batch["legal_moves"].shape
        lm = torch.zeros((B, 121), dtype=batch["legal_moves"].dtype)
        mp = torch.zeros((B, 121), dtype=batch["moves_prob"].dtype)
        lm[:, :k], mp[:, :k] = batch["legal_moves"], batch["moves_prob"]
        a = net.forward(batch["board"], batch["legal_moves"])
        b = net.forward_embedded(got, lm)
        legal = batch["legal_moves"] > 0
        assert torch.equal(a["value"], b["value"])
        assert float((a["moves_logprob"] - b["moves_logprob"][:, :k])[legal].abs().max()) <= 1e-6
        la = -(batch["moves_prob"] * a["moves_logprob"]).sum() / B
        lb = -(mp * b["moves_logprob"]).sum() / B
        assert abs(float(la) - float(lb)) <= 1e-6


def test_supervised_step_matches_reference_cpu():
    run_steps("cpu", 2e-6)


@pytest.mark.gpu
def test_supervised_step_matches_reference_gpu():
    run_steps("cuda:0", 2e-4)


@pytest.mark.gpu
def test_graphed_train_step_matches_the_eager_step():
    """GraphedTrainStep (the step captured as a HIP graph, inputs padded to all cells, embedding by masks) against
    supervised_step on the same batches: the three G9 batches cycled four times from G9's initial weights, the learning
    rate changed after the eighth step (the graph is captured at step 4 and again at step 9).  Per-step losses and
    outputs, then every updated tensor."""
    from azalea_amd.network import HexNetwork
    from azalea_amd.policy_trainer import GraphedTrainStep, supervised_step
    from azalea_amd.prep import torch_batch_replays
    z = np.load(os.path.join(GOLDEN, "g9_train_step.npz"))
    frame = source_frame(load_g7())
    dev = "cuda:0"
    nets, opts = [], []
    for _ in range(2):
        net = HexNetwork(board_size=11, num_blocks=2, base_chans=16)
        net.load_state_dict({k[3:]: torch.tensor(z[k]) for k in z.files if k.startswith("w0:")})
        net.to(dev)
        nets.append(net)
        opts.append(torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4))
    B = len(z["batch_idx"][0])
    TOLG = 1e-4      # graphed vs eager on the same device: same kernels but for the embedding and the padding
    gstep = GraphedTrainStep(nets[1], opts[1], B, dev)
    worst = 0.0
    for step in range(12):
        if step == 8:
            for o in opts:
                o.param_groups[0]["lr"] = 0.03
        ids = z["batch_idx"][step % 3]
        batch = torch_batch_replays([frame[int(i)] for i in ids])
        o, loss = supervised_step(nets[0], dict(batch), train=True, optimizer=opts[0], device=dev)
        gl = gstep.step({k: v.to(dev) for k, v in batch.items()}).cpu().numpy()
        k = batch["legal_moves"].shape[1]
        go = gstep.outputs(k)
        assert abs(float(gl[0]) - loss) <= TOLG and abs(float(gl[1]) - o["value_loss"]) <= TOLG and abs(float(gl[2]) - o["moves_loss"]) <= TOLG
        legal = batch["legal_moves"].numpy() > 0
        worst = max(worst, float(np.abs(go["value"].cpu().numpy() - o["value"].cpu().numpy()).max()),
                    float(np.abs(go["moves_logprob"].cpu().numpy() - o["moves_logprob"].cpu().numpy())[legal].max()))
        if step < 3:      # the first three are also the reference's recorded steps
            assert abs(float(gl[0]) - float(z["step%d_loss" % step])) <= 2e-3  # the GPU tolerance of the golden itself (2e-4 x 10)
    assert gstep.captures == 2
    assert worst <= TOLG, worst
    a, b = nets[0].state_dict(), nets[1].state_dict()
    for name, v in a.items():
        if v.dtype.is_floating_point:
            assert float((v - b[name]).abs().max()) <= TOLG, name
        else:
            assert torch.equal(v, b[name]), name


@pytest.mark.gpu
def test_ring_fed_graphed_step_equals_the_batch_fed_one():
    """GraphedTrainStep.step_from_ring (azx_replay_collate writing straight into the step's static inputs) against
    GraphedTrainStep.step on DeviceReplayBuffer.sample of the same rows: identical inputs -- checked bit for bit on the
    static tensors -- hence the same losses and weights over eight steps (to 1e-5: MIOpen's backward kernels sum in no
    fixed order); collate_into rejects a buffer of the wrong shape."""
    from azalea_amd import engine as eng
    from azalea_amd.device_replay import DeviceReplayBuffer
    from azalea_amd.network import HexNetwork
    from azalea_amd.policy_trainer import GraphedTrainStep
    dev = torch.device("cuda", 0)
    E = eng.Engine(board_size=5, n_games=64, simulations=20, search_batch_size=10, evaluator=eng.EVAL_UNIFORM,
                   noise_scale=0.25)
    buf = DeviceReplayBuffer(E, 2000, shared=False)
    E.replay_fill(1500)
    B = 32
    nets, steps = [], []
    for _ in range(2):
        torch.manual_seed(4)
        net = HexNetwork(board_size=5, num_blocks=1, base_chans=16).to(dev)
        opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
        nets.append(net)
        steps.append(GraphedTrainStep(net, opt, B, dev))
    rng = np.random.RandomState(0)
    for i in range(8):
        idx = rng.randint(0, len(buf), B)
        la = steps[0].step(buf.sample(idx)).cpu().numpy().copy()
        lb, k = steps[1].step_from_ring(buf, idx)
        assert np.abs(la - lb.cpu().numpy()).max() <= 1e-5 and 1 <= k <= 25
        for name in ("board", "legal_moves", "moves_prob", "reward"):
            assert torch.equal(getattr(steps[0], name), getattr(steps[1], name)), name
    for (na, a), (_, b) in zip(nets[0].state_dict().items(), nets[1].state_dict().items()):
        assert float((a.double() - b.double()).abs().max()) <= 1e-5, na
    with pytest.raises(ValueError):
        buf.collate_into(np.arange(B), dict(color=steps[1]._color, legal_moves=steps[1].legal_moves[:, :10].contiguous(),
                                            result=steps[1]._result, board=steps[1].board,
                                            moves_prob=steps[1].moves_prob, reward=steps[1].reward))
    E.close()


@pytest.mark.gpu
def test_policy_engine_sees_weights_updated_by_graph_replays():
    """A graph replay updates the parameters without moving their autograd version counters, which is what
    Policy._sync_weights watches: after graphed training steps the Policy's own engine (parity mode: choose_action,
    evaluate) must evaluate with the NEW weights."""
    from azalea_amd.policy import Policy
    from azalea_amd.policy_trainer import GraphedTrainStep
    from azalea_amd.prep import torch_batch_replays
    dev = "cuda:0"
    torch.manual_seed(11)
    p = Policy()
    p.initialize(dict(device=dev, network="HexNetwork", board_size=11, num_blocks=2, base_chans=64, simulations=20,
                      search_batch_size=10, exploration_coef=0.5, exploration_depth=4, exploration_noise_alpha=0.3,
                      exploration_noise_scale=0.25, exploration_temperature=1.0))
    p.net.to(dev)
    eng_ = p._get_engine(11)
    rng = np.random.RandomState(3)
    board = rng.randint(0, 3, (6, 11, 11)).astype(np.int32)
    board[rng.rand(6, 11, 11) < 0.5] = 0
    lm = np.zeros((6, 121), np.int32)
    for i in range(6):
        e = np.flatnonzero(board[i].ravel() == 0) + 1
        lm[i, :len(e)] = e
    legal = lm > 0

    def engine_out():
        p._sync_weights(eng_)
        v, lp = eng_.forward(board, lm)
        return np.concatenate([v, lp[legal]])

    def torch_out():
        p.net.eval()
        with torch.no_grad():
            o = p.net(torch.tensor(board, device=dev), torch.tensor(lm, device=dev))
        return np.concatenate([o["value"].cpu().numpy(), o["moves_logprob"].cpu().numpy()[legal]])

    assert np.abs(engine_out() - torch_out()).max() <= 1e-4
    z = np.load(os.path.join(GOLDEN, "g9_train_step.npz"))
    frame = source_frame(load_g7())
    opt = torch.optim.SGD(p.net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    B = len(z["batch_idx"][0])
    gs = GraphedTrainStep(p.net, opt, B, dev)
    for step in range(5):                                  # three eager warm-up steps, then capture + replay
        batch = torch_batch_replays([frame[int(i)] for i in z["batch_idx"][step % 3]])
        gs.step({k: v.to(dev) for k, v in batch.items()})
    t1 = torch_out()
    assert np.abs(engine_out() - t1).max() <= 1e-4          # synced after the capture: versions as they will stay
    for step in range(5, 13):                              # replays only
        batch = torch_batch_replays([frame[int(i)] for i in z["batch_idx"][step % 3]])
        gs.step({k: v.to(dev) for k, v in batch.items()})
    t2 = torch_out()
    assert np.abs(t2 - t1).max() > 1e-2                    # the weights did move
    assert np.abs(engine_out() - t2).max() <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["eager", "graph", "native"])
def test_train_loop_with_device_replay(tmp_path, mode):
    """policy_trainer.train end to end on the GPU: engine self-play -> HBM replay ring -> GPU collate
    -> supervised_step (or, with config["train_step_graph"], the captured step; with config["train_step_native"], the
    hand-written one); the checkpoint it writes loads back through Policy.load."""
    from azalea_amd.policy import Policy
    from azalea_amd.policy_trainer import initialize_replay_buffer, train
    from azalea_amd.game.hex import HexGame
    config = dict(seed=3, device="cuda:0", replaybuf_oversampling=2, batch_size=32, game="azalea_amd.game.hex.HexGame",
                  board_size=5, replaybuf_size=128, lr_initial=0.05, momentum=0.9, l2_regularization=1e-4,
                  lr_decay_epochs=2, lr_decay=0.5, total_epochs=4, network="HexNetwork", num_blocks=1,
                  base_chans=16 if mode == "native" else 8,
                  simulations=20, search_batch_size=10, exploration_coef=0.5, exploration_depth=4,
                  exploration_noise_alpha=0.3, exploration_noise_scale=0.25, exploration_temperature=1.0,
                  log_interval=2, model_checkpoint_interval=0, selfplay_games=16, train_step_graph=mode == "graph",
                  train_step_native=mode == "native")
    policy = Policy()
    policy.initialize(config)
    before = {k: v.clone() for k, v in policy.net.state_dict().items()}
    buf = initialize_replay_buffer(None, lambda: HexGame(5), config["replaybuf_size"])
    n0 = len(buf)
    hist = {}
    path = train(policy, config, str(tmp_path), replaybuf=buf, device_replay=True, history=hist)
    assert os.path.exists(path)
    # the reference steps StepLR at the top of each epoch (policy_trainer.py:81) under torch 0.4.1 / 1.0.1, whose
    # constructor leaves last_epoch at -1: epoch e = 1..4 trains with lr_initial * lr_decay ** ((e - 1) // lr_decay_epochs)
    assert np.allclose(hist["lr"], [0.05 * 0.5 ** ((e - 1) // 2) for e in (1, 2, 3, 4)])
    after = policy.net.state_dict()
    assert any(not torch.equal(before[k].cpu(), after[k].cpu()) for k in before)
    q = Policy.load(path, device="cpu")
    for k, v in after.items():
        assert torch.equal(v.cpu(), q.net.state_dict()[k]), k
    assert n0 >= config["replaybuf_size"]


def test_native_train_step_refuses_what_it_does_not_cover():
    """NativeTrainStep is HIP-only and specific: a CPU device, another optimizer or another module is a ValueError up
    front (the eager / captured steps apply there), never a silent fallback."""
    from azalea_amd.native_train import NativeTrainStep
    from azalea_amd.network import HexNetwork
    net = HexNetwork(board_size=5, num_blocks=1, base_chans=16)
    sgd = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9)
    with pytest.raises(ValueError):
        NativeTrainStep(net, sgd, 8, "cpu")
    with pytest.raises(ValueError):
        NativeTrainStep(net, torch.optim.Adam(net.parameters()), 8, "cuda:0")
    with pytest.raises(ValueError):
        NativeTrainStep(net, torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, nesterov=True), 8, "cuda:0")
    with pytest.raises(ValueError):
        NativeTrainStep(torch.nn.Linear(3, 3), sgd, 8, "cuda:0")
    with pytest.raises(ValueError):
        NativeTrainStep(net, torch.optim.SGD(list(net.parameters())[:3], lr=0.1), 8, "cuda:0")


def test_native_step_envelope_is_what_train_picks_by():
    """native_train.unsupported_reason (no GPU needed for the shape logic): 16 / 32 / 64 channels up to 11x11, 64 also on
    12x12 / 13x13 (through the wide step), 128 / 256 from 3x3 to 13x13; anything else names the reason -- train() then
    takes the stock step (make_train_step)."""
    from azalea_amd.native_train import SUPPORTED_SHAPES
    ok = [(11, 64), (2, 16), (9, 32), (13, 256), (13, 128), (3, 128), (11, 256), (13, 64), (12, 64)]
    bad = [(13, 32), (12, 16), (2, 128), (14, 256), (14, 64), (11, 48), (11, 512)]
    assert all(k in SUPPORTED_SHAPES for k in ok) and not any(k in SUPPORTED_SHAPES for k in bad)
    import torch
    from azalea_amd.network import HexNetwork
    from azalea_amd.native_train import unsupported_reason
    net = HexNetwork(board_size=5, num_blocks=1, base_chans=16)
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    assert "CUDA" in unsupported_reason(net, opt, "cpu")
    from azalea_amd.policy_trainer import make_train_step
    assert make_train_step(net, opt, 4, "cpu", {}) == (None, "eager")


def test_to_mover_view_flips_only_the_second_players_rows():
    """config["train_mover_view"] on host batches: game.flip_player_board_moves on the rows with color == 1
    (what mcts.py:178-181 does before every evaluation), everything else as collated."""
    import numpy as np
    import torch
    from azalea_amd.game.hex import HexGame
    from azalea_amd.policy_trainer import to_mover_view
    rng = np.random.RandomState(4)
    n, B = 5, 9
    board = rng.randint(0, 3, (B, n, n)).astype(np.int32)
    moves = np.zeros((B, n * n), np.int32)
    for i in range(B):
        e = np.flatnonzero(board[i].ravel() == 0) + 1
        moves[i, :len(e)] = e
    color = np.array([0, 1, 1, 0, 1, 0, 0, 1, 1])
    batch = dict(board=torch.tensor(board), legal_moves=torch.tensor(moves), color=torch.tensor(color),
                 moves_prob=torch.rand(B, n * n), reward=torch.ones(B))
    keep = {k: v.clone() for k, v in batch.items()}
    out = to_mover_view(batch, HexGame)
    m = color == 1
    fb, fm = HexGame.flip_player_board_moves(board[m], moves[m])
    assert np.array_equal(out["board"][m].numpy(), fb) and np.array_equal(out["legal_moves"][m].numpy(), fm)
    assert torch.equal(out["board"][~torch.tensor(m)], keep["board"][~torch.tensor(m)])
    assert torch.equal(out["moves_prob"], keep["moves_prob"]) and torch.equal(out["color"], keep["color"])
    # in the flipped view the mover's stones are colour 1 and every listed move is an empty cell of the flipped board
    for i in np.flatnonzero(m):
        k = int((moves[i] > 0).sum())
        cells = out["legal_moves"][i, :k].numpy() - 1
        assert np.all(out["board"][i].reshape(-1).numpy()[cells] == 0)
        assert int((out["board"][i] == 1).sum()) == int((board[i] == 2).sum())
