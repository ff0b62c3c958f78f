"""The hand-written training step (csrc/train_kernels.hip, azalea_amd/native_train.py) against (a) torch autograd layer
by layer -- every intermediate the kernels keep (pre-BatchNorm outputs, activations, masked gradients, parameter
gradients) --, (b) the reference's own recorded step (golden G9: azalea/policy_trainer.py:123-142 run by importing the
reference; losses, outputs, updated tensors incl. BatchNorm running statistics) and (c) the eager PyTorch step over
twelve steps at the reference's training shape (6x64, batch 128)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from replay_golden import load_g7, source_frame

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda:0"


def _random_batch(n, B, seed):
    rng = np.random.RandomState(seed)
    cells = n * n
    board = rng.randint(0, 3, (B, n, n)).astype(np.int32)
    board[rng.rand(B, n, n) < 0.4] = 0
    lm = np.zeros((B, cells), np.int32)
    mp = np.zeros((B, cells), np.float32)
    for i in range(B):
        e = np.flatnonzero(board[i].ravel() == 0) + 1
        lm[i, :len(e)] = e
        p = rng.dirichlet(np.full(len(e), 0.3)).astype(np.float32)
        mp[i, :len(e)] = p
    reward = rng.choice([-1.0, 1.0], B).astype(np.float32)
    k = int((lm > 0).sum(1).max())
    return dict(board=torch.tensor(board), legal_moves=torch.tensor(lm[:, :k]), moves_prob=torch.tensor(mp[:, :k]),
                reward=torch.tensor(reward))


def _net(n, blocks, chans, seed=0):
    from azalea_amd.network import HexNetwork
    torch.manual_seed(seed)
    net = HexNetwork(board_size=n, num_blocks=blocks, base_chans=chans)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():      # BatchNorm affine terms away from (1, 0) so that their gradients matter
        for name, p in net.named_parameters():
            if "bn" in name and name.endswith("weight"):
                p.copy_(0.5 + torch.rand(p.shape, generator=g))
            elif "bn" in name and name.endswith("bias"):
                p.copy_(0.2 * torch.randn(p.shape, generator=g))
    return net.to(DEV)


@pytest.mark.parametrize("n,blocks,chans,B", [(11, 2, 64, 8), (11, 1, 16, 5), (9, 2, 32, 7), (11, 6, 64, 128),
                                              (5, 1, 128, 4), (7, 2, 256, 6), (11, 2, 128, 9), (11, 3, 256, 16),
                                              (13, 1, 128, 5), (13, 2, 256, 7), (12, 1, 256, 3),
                                              (13, 2, 64, 6), (12, 1, 64, 5), (13, 6, 64, 128)])      # 64 channels on 12x12 / 13x13: the wide step
def test_every_intermediate_matches_autograd(n, blocks, chans, B):
    from azalea_amd.native_train import NativeTrainStep
    net, ref = _net(n, blocks, chans), _net(n, blocks, chans)
    batch = {k: v.to(DEV) for k, v in _random_batch(n, B, 5).items()}
    cells, L = n * n, 2 * blocks
    # ---- torch: the same forward with hooks on every pre-BN output and activation ----
    ref.train()
    raws, acts = {}, {}

    def nhwc(t):
        return t.permute(0, 2, 3, 1).reshape(B, cells, -1)
    x0 = ref.encoder(batch["board"].long()).permute(0, 3, 1, 2).contiguous()
    r = ref.conv1(x0); r.retain_grad(); raws[0] = r
    x = F.relu(ref.bn1(r)); x.retain_grad(); acts[0] = x
    for k, blk in enumerate(ref.resblocks):
        r1 = blk.conv1(x); r1.retain_grad(); raws[2 * k + 1] = r1
        y = F.relu(blk.bn1(r1)); y.retain_grad(); acts[2 * k + 1] = y
        r2 = blk.conv2(y); r2.retain_grad(); raws[2 * k + 2] = r2
        x = F.relu(blk.bn2(r2) + x); x.retain_grad(); acts[2 * k + 2] = x
    v = F.relu(ref.value_bn1(ref.value_conv1(x))).flatten(1)
    v = ref.value_fc3(F.relu(ref.value_fc2(v)))
    value = torch.tanh(v).squeeze(1)
    p = F.relu(ref.move_bn1(ref.move_conv1(x))).flatten(1)
    logit = ref.move_fc(p)
    lm = batch["legal_moves"]
    logit = torch.gather(logit, 1, (lm - 1).clamp(min=0).long()).masked_fill(lm == 0, -99)
    logprob = F.log_softmax(logit, dim=1)
    value_loss = F.mse_loss(value, batch["reward"])
    moves_loss = -(batch["moves_prob"] * logprob).sum() / B
    (value_loss + moves_loss).backward()
    # ---- native: one step with lr = 0 (nothing moves; every buffer stays inspectable) ----
    opt = torch.optim.SGD(net.parameters(), lr=0.0, momentum=0.9, weight_decay=0.0)
    step = NativeTrainStep(net, opt, B, DEV)
    loss = step.step(batch).cpu().numpy()
    torch.cuda.synchronize()
    C = chans

    def close(a, b, tol, what):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        scale = max(1.0, float(np.abs(b).max()))
        err = float(np.abs(a - b).max()) / scale
        assert err <= tol, "%s: %.3g (scale %.3g)" % (what, err, scale)
    assert abs(loss[1] - float(value_loss.detach())) <= 1e-5 and abs(loss[2] - float(moves_loss.detach())) <= 1e-5
    for l in range(L + 1):
        close(step.debug("raw%d" % l).reshape(B, cells, C), nhwc(raws[l].detach()).cpu().numpy(), 2e-5, "raw%d" % l)
        close(step.debug("act%d" % l).reshape(B, cells, C), nhwc(acts[l].detach()).cpu().numpy(), 2e-5, "act%d" % l)
    legal = (lm > 0).cpu().numpy()
    k = lm.shape[1]
    close(step.out_value.cpu().numpy(), value.detach().cpu().numpy(), 1e-5, "value")
    close(step.out_logprob.cpu().numpy()[:, :k][legal], logprob.detach().cpu().numpy()[legal], 1e-5, "logprob")
    # g_l = dL/d(activation l) masked by its ReLU; torch keeps dL/dact: mask it the same way.  Gradients are ~1e-3 ..
    # 1e-6 in absolute terms, so they are compared relative to their own size: in the 2-norm (an activation within
    # rounding of zero may pass the ReLU on one side and not on the other: a handful of the 10^6 entries of a layer,
    # each a legitimate fp32 answer) and, loosely, entry by entry
    def rel(got, want, what, tol2, tolmax):
        got, want = np.asarray(got, np.float64).ravel(), np.asarray(want, np.float64).ravel()
        n2, mx = float(np.linalg.norm(want)), float(np.abs(want).max())
        assert float(np.linalg.norm(got - want)) <= tol2 * max(n2, 1e-30), "%s: |d|2 %.3g of %.3g" % (what, np.linalg.norm(got - want), n2)
        assert float(np.abs(got - want).max()) <= tolmax * max(mx, 1e-30), "%s: max %.3g of %.3g" % (what, np.abs(got - want).max(), mx)
    # (at batch 128 a layer has 10^6 entries and a few kink decisions do differ -- they add up down the chain; the small
    # shapes pin the arithmetic tightly, the training shape is held to the size of that effect and, below, to fp64)
    big = B * cells * C > 200000
    for l in range(L, -1, -1):
        want = (nhwc(acts[l].grad) * (nhwc(acts[l].detach()) > 0)).cpu().numpy()
        rel(step.debug("g%d" % l), want, "g%d" % l, 2e-2 if big else 2e-4, 1.0 if big else 2e-2)
    for name, prm in ref.named_parameters():
        rel(step.debug("grad:" + name), prm.grad.detach().cpu().numpy(), "grad " + name, 5e-3 if big else 2e-4, 5e-2 if big else 1e-3)
    if big:
        # calibration against float64 autograd: both fp32 computations sit a few kink decisions (~1e-3 of the gradient's
        # norm) away from it -- the native one within a small multiple of torch's own distance
        ref64 = _net(n, blocks, chans).double().train()
        o64 = ref64.forward(batch["board"], batch["legal_moves"])
        l64 = F.mse_loss(o64["value"], batch["reward"].double()) - (batch["moves_prob"].double() * o64["moves_logprob"]).sum() / B
        l64.backward()
        for (name, p32), (_, p64) in zip(ref.named_parameters(), ref64.named_parameters()):
            truth = p64.grad.cpu().numpy().ravel()
            e_torch = float(np.linalg.norm(p32.grad.double().cpu().numpy().ravel() - truth))
            e_native = float(np.linalg.norm(step.debug("grad:" + name).astype(np.float64) - truth))
            assert e_native <= max(5.0 * e_torch, 5e-3 * float(np.linalg.norm(truth))), (name, e_native, e_torch)
    # lr = 0, weight decay 0: nothing but the BatchNorm statistics moved
    for (name, a), (_, b) in zip(net.state_dict().items(), ref.state_dict().items()):
        if a.dtype.is_floating_point:
            close(a.cpu().numpy(), b.cpu().numpy(), 2e-5, name)
        else:
            assert torch.equal(a, b), name
    step.close()


@pytest.mark.parametrize("passes", [("FWD", "BWD", "WGRAD"), ("BWD",), ("FWD", "WGRAD")])
def test_exact_fp32_kernels_stay_green(monkeypatch, passes):
    """AZX_TRAIN_FWD / _BWD / _WGRAD=fp32 select the exact-fp32 MFMA kernel of a pass (read when the trainer is
    created): every combination shares the prologues, the partial sums and the epilogues with the split-f16 kernels
    and is held to the same comparison."""
    for name in passes:
        monkeypatch.setenv("AZX_TRAIN_" + name, "fp32")
    test_every_intermediate_matches_autograd(11, 2, 64, 8)
    test_every_intermediate_matches_autograd(9, 2, 32, 7)
    # the smallest boards: 2x2 has two k-steps for the filter gradient's four waves, 3x3 five
    test_odd_batches_and_boards(2, 1, 64, 3)
    test_odd_batches_and_boards(3, 1, 16, 1)
    test_odd_batches_and_boards(7, 2, 32, 300)


@pytest.mark.parametrize("on", ["0", "1"])
def test_both_wide_filter_gradient_kernels_stay_green(monkeypatch, on):
    """AZX_TRAIN_WGRAD2 (read when the trainer is created): k_tw_wgrad2 is the default at 256 channels, k_tw_wgrad at 128 --
    each is also held to autograd where the other is the default (0: k_tw_wgrad at 256 channels; 1: k_tw_wgrad2 at 128)."""
    monkeypatch.setenv("AZX_TRAIN_WGRAD2", on)
    if on == "0":
        test_every_intermediate_matches_autograd(13, 2, 256, 7)
        test_odd_batches_and_boards(9, 2, 256, 70)
    else:
        test_every_intermediate_matches_autograd(13, 1, 128, 5)
        test_every_intermediate_matches_autograd(11, 2, 128, 9)
        test_odd_batches_and_boards(13, 1, 128, 67)


@pytest.mark.parametrize("n,blocks,chans,B", [(11, 1, 64, 200), (7, 2, 32, 300), (3, 1, 16, 1), (11, 2, 16, 131), (2, 1, 64, 3),
                                              (11, 1, 128, 131), (3, 1, 256, 1), (9, 2, 256, 70), (13, 1, 128, 67),
                                              (11, 2, 64, 257), (5, 2, 16, 1031), (9, 1, 128, 261), (5, 2, 256, 515),
                                              (12, 1, 256, 9), (4, 1, 256, 3), (6, 1, 256, 20)])     # even boards: k_tw_wgrad2's row halves are N / 2 + N / 2
def test_odd_batches_and_boards(n, blocks, chans, B):
    """Batches that are not a multiple of anything the kernels tile by -- more boards than one round of partial-sum
    loads covers (> 128), more than TRN_PRESUM_BATCH = 256 (the batch sums then come from k_trn_totals behind every
    producer), filter-gradient groups of unequal size (B not a multiple of 64), a single board -- and the smallest boards: loss, outputs and every parameter gradient against float64 autograd, within a small multiple of
    torch's own fp32 distance."""
    from azalea_amd.native_train import NativeTrainStep
    batch = {k: v.to(DEV) for k, v in _random_batch(n, B, 21).items()}
    grads, outs = {}, {}
    for dtype in (torch.float32, torch.float64):
        net = _net(n, blocks, chans, seed=9).to(dtype).train()
        o = net.forward(batch["board"], batch["legal_moves"])
        loss = F.mse_loss(o["value"], batch["reward"].to(dtype)) - (batch["moves_prob"].to(dtype) * o["moves_logprob"]).sum() / B
        loss.backward()
        grads[dtype] = {name: p.grad.double().cpu().numpy().ravel() for name, p in net.named_parameters()}
        outs[dtype] = (float(loss.detach()), o["value"].detach().double().cpu().numpy(), o["moves_logprob"].detach().double().cpu().numpy())
    net = _net(n, blocks, chans, seed=9).train()
    step = NativeTrainStep(net, torch.optim.SGD(net.parameters(), lr=0.0, momentum=0.9), B, DEV)
    loss = step.step(batch).cpu().numpy()
    torch.cuda.synchronize()
    k = batch["legal_moves"].shape[1]
    legal = (batch["legal_moves"] > 0).cpu().numpy()
    assert abs(float(loss[0]) - outs[torch.float64][0]) <= 2e-5 * max(1.0, abs(outs[torch.float64][0]))
    assert np.abs(step.out_value.cpu().numpy() - outs[torch.float64][1]).max() <= 2e-5
    assert np.abs(step.out_logprob.cpu().numpy()[:, :k][legal] - outs[torch.float64][2][legal]).max() <= 5e-5
    for name, truth in grads[torch.float64].items():
        got = step.debug("grad:" + name).astype(np.float64)
        nt = float(np.linalg.norm(truth))
        e_torch = float(np.linalg.norm(grads[torch.float32][name] - truth))
        e_native = float(np.linalg.norm(got - truth))
        # (past ~2e5 activations per layer a few ReLU decisions at rounding distance from zero differ between any two
        # fp32 computations, see test_every_intermediate_matches_autograd)
        tol = 5e-3 if B * n * n * chans > 200000 else 2e-5
        assert e_native <= max(5.0 * e_torch, tol * nt, 1e-12), (name, e_native, e_torch, nt)
    step.close()


@pytest.mark.parametrize("kind", ["tiny", "huge"])
@pytest.mark.parametrize("n,blocks,chans,B", [(5, 2, 16, 6), (11, 3, 64, 16)])
def test_gradient_range_is_managed(kind, n, blocks, chans, B):
    """The backward pass runs on split-f16 operands scaled per layer by a power of two taken from max |g_l|
    (train_kernels.hip, ROLE_BWD16 / k_trn_wgrad16).  Two networks whose gradients sit far outside the f16 range:
    'tiny' -- every tower filter x 1e-6 (nothing an unscaled f16 pair resolves), so each BatchNorm works in its eps
    regime and the gradients inside the blocks are ~1e-7; 'huge' -- rewards of +-1e6, the value head's gradients ~1e6
    times the usual (the loss is linear in them: no forward rounding is amplified).  Parameter gradients against
    float64 autograd, no further from it than a small multiple of torch's own fp32 distance."""
    from azalea_amd.native_train import NativeTrainStep

    def make(dtype):
        net = _net(n, blocks, chans, seed=7)
        with torch.no_grad():
            if kind == "tiny":
                for blk in net.resblocks:
                    blk.conv1.weight.mul_(1e-6)
                    blk.conv2.weight.mul_(1e-6)
            else:
                pass                   # (the rewards below)
        return net.to(dtype).train()
    batch = {k: v.to(DEV) for k, v in _random_batch(n, B, 11).items()}
    if kind == "huge":
        batch["reward"] = batch["reward"] * 1e6
    grads = {}
    for dtype in (torch.float32, torch.float64):
        net = make(dtype)
        o = net.forward(batch["board"], batch["legal_moves"])
        loss = F.mse_loss(o["value"], batch["reward"].to(dtype)) - (batch["moves_prob"].to(dtype) * o["moves_logprob"]).sum() / B
        loss.backward()
        grads[dtype] = {name: p.grad.double().cpu().numpy().ravel() for name, p in net.named_parameters()}
    net = make(torch.float32)
    step = NativeTrainStep(net, torch.optim.SGD(net.parameters(), lr=0.0, momentum=0.9), B, DEV)
    step.step(batch)
    torch.cuda.synchronize()
    sizes = []
    for name, truth in grads[torch.float64].items():
        got = step.debug("grad:" + name).astype(np.float64)
        assert np.isfinite(got).all(), name
        nt = float(np.linalg.norm(truth))
        e_torch = float(np.linalg.norm(grads[torch.float32][name] - truth))
        e_native = float(np.linalg.norm(got - truth))
        assert e_native <= max(5.0 * e_torch, 1e-5 * nt), (name, e_native, e_torch, nt)
        sizes.append(nt)
    g_last = float(np.abs(step.debug("g%d" % (2 * blocks))).max())
    g_inner = float(np.abs(step.debug("g1")).max())
    if kind == "tiny":
        assert 0 < g_inner < 6e-5, g_inner          # below the f16 normal range: an unscaled split keeps no low part
    else:
        assert g_last > 1e3, g_last                 # and beyond it once 576 taps x channels are summed
    step.close()


def test_native_step_matches_the_reference_recorded_steps():
    """Golden G9: three supervised_step calls of the reference (SGD 0.1 / 0.9 / 1e-4, train-mode BatchNorm) from the
    recorded initial weights -- the bounds the eager GPU step is held to (test_train_step.py: 2e-4 on the updated
    tensors, ten times that on the per-step outputs)."""
    from azalea_amd.native_train import NativeTrainStep
    from azalea_amd.network import HexNetwork
    from azalea_amd.policy_trainer import supervised_step
    from azalea_amd.prep import torch_batch_replays
    z = np.load(os.path.join(GOLDEN, "g9_train_step.npz"))
    frame = source_frame(load_g7())
    net = HexNetwork(board_size=11, num_blocks=2, base_chans=16)
    net.load_state_dict({k[3:]: torch.tensor(z[k]) for k in z.files if k.startswith("w0:")})
    net.to(DEV)
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    B = len(z["batch_idx"][0])
    step = NativeTrainStep(net, opt, B, DEV)
    tol = 2e-4
    for s, ids in enumerate(z["batch_idx"]):
        batch = torch_batch_replays([frame[int(i)] for i in ids])
        loss = step.step({k: v.to(DEV) for k, v in batch.items()}).cpu().numpy()
        assert abs(loss[0] - float(z["step%d_loss" % s])) <= tol * 10
        assert abs(loss[1] - float(z["step%d_value_loss" % s])) <= tol * 10
        assert abs(loss[2] - float(z["step%d_moves_loss" % s])) <= tol * 10
        k = batch["legal_moves"].shape[1]
        o = step.outputs(k)
        assert np.abs(o["value"].cpu().numpy() - z["step%d_value" % s]).max() <= tol * 10
        legal = batch["legal_moves"].numpy() > 0
        assert np.abs(o["moves_logprob"].cpu().numpy() - z["step%d_moves_logprob" % s])[legal].max() <= tol * 10
    for k, v in net.state_dict().items():
        want = z["w3:" + k]
        if want.dtype.kind == "f":
            assert np.abs(v.cpu().numpy() - want).max() <= tol, k
        else:
            assert np.array_equal(v.cpu().numpy(), want), k
    # the optimizer's own state is what the kernels used: the eval-mode pass after the three steps agrees too
    batch = torch_batch_replays([frame[int(i)] for i in z["batch_idx"][0]])
    o, loss = supervised_step(net, batch, train=False, device=DEV)
    assert abs(loss - float(z["eval_loss"])) <= tol * 10
    step.close()


@pytest.mark.parametrize("blocks,chans,B", [(6, 64, 128), (3, 256, 64)])
def test_native_step_tracks_the_eager_step_over_twelve_steps(blocks, chans, B):
    """6x64 on 11x11, batch 128 (config/hex11_train_config.yml) -- and a wide tower, 3x256 --, SGD 0.1 -> 0.03 after eight steps, against
    policy_trainer.supervised_step: twelve steps of the eager trajectory, the native step taken from the SAME state
    each time (weights, BatchNorm buffers, momentum copied over before the step -- two fp32 trajectories at lr 0.1
    drift apart by themselves, which would measure the optimisation's sensitivity, not the step): losses, outputs,
    then every tensor and every momentum buffer after the step."""
    from azalea_amd.native_train import NativeTrainStep
    from azalea_amd.policy_trainer import supervised_step
    n = 11
    nets = [_net(n, blocks, chans, seed=2), _net(n, blocks, chans, seed=2)]
    opts = [torch.optim.SGD(m.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4) for m in nets]
    step = NativeTrainStep(nets[1], opts[1], B, DEV)
    worst = {"loss": 0.0, "out": 0.0, "tensor": 0.0, "momentum": 0.0}
    for i in range(12):
        if i == 8:
            for o in opts:
                o.param_groups[0]["lr"] = 0.03
        with torch.no_grad():                       # same state in
            for (_, a), (_, b) in zip(nets[0].state_dict().items(), nets[1].state_dict().items()):
                b.copy_(a)
            for pa, pb in zip(nets[0].parameters(), nets[1].parameters()):
                ma = opts[0].state.get(pa, {}).get("momentum_buffer")
                if ma is not None:
                    opts[1].state[pb]["momentum_buffer"].copy_(ma)
        batch = _random_batch(n, B, 100 + i % 4)
        o, loss = supervised_step(nets[0], dict(batch), train=True, optimizer=opts[0], device=DEV)
        nl = step.step({k: v.to(DEV) for k, v in batch.items()}).cpu().numpy()
        worst["loss"] = max(worst["loss"], abs(float(nl[0]) - loss), abs(float(nl[1]) - o["value_loss"]), abs(float(nl[2]) - o["moves_loss"]))
        legal = batch["legal_moves"].numpy() > 0
        k = batch["legal_moves"].shape[1]
        no = step.outputs(k)
        worst["out"] = max(worst["out"], float(np.abs(no["value"].cpu().numpy() - o["value"].cpu().numpy()).max()),
                           float(np.abs(no["moves_logprob"].cpu().numpy() - o["moves_logprob"].cpu().numpy())[legal].max()))
        a, b = nets[0].state_dict(), nets[1].state_dict()
        for name, v in a.items():
            if v.dtype.is_floating_point:
                worst["tensor"] = max(worst["tensor"], float((v - b[name]).abs().max()))
            else:
                assert torch.equal(v, b[name]), name
        for pa, pb in zip(nets[0].parameters(), nets[1].parameters()):
            ma, mb = opts[0].state[pa]["momentum_buffer"], opts[1].state[pb]["momentum_buffer"]
            worst["momentum"] = max(worst["momentum"], float((ma - mb).abs().max()) / max(1e-6, float(ma.abs().max())))
    assert step.steps == 12
    assert worst["loss"] <= 1e-4 and worst["out"] <= 1e-4, worst
    assert worst["tensor"] <= 2e-4, worst         # G9's bound on an updated tensor
    # relative to the buffer's largest entry (ReLU-kink decisions, see above; the wide case has half the batch to average
    # them over and sums its head / stem gradients with f64 atomics in arrival order: measured 1.1e-2 .. 2.6e-2 run to run)
    assert worst["momentum"] <= (2e-2 if chans <= 64 else 5e-2), worst
    # the momentum buffers are the optimizer's own tensors: its state_dict is a normal SGD checkpoint
    sd = opts[1].state_dict()
    assert len(sd["state"]) == len(list(nets[1].parameters()))
    step.close()


@pytest.mark.parametrize("chans", [16, 128])
def test_native_and_eager_training_learn_the_same_thing(chans):
    """A sanity check no parity bound can give: 150 steps on a fixed set of 64 rows (5x5, 2x16 -- and 2x128, the wide
    step) -- the hand-written step and the eager one both drive the loss down, to the same level (their fp32
    trajectories are not bitwise twins, the optimisation is the same)."""
    from azalea_amd.native_train import NativeTrainStep
    from azalea_amd.policy_trainer import supervised_step
    n, B = 5, 32
    data = [_random_batch(n, B, 300 + i) for i in range(2)]
    finals, firsts = [], []
    for kind in ("eager", "native"):
        net = _net(n, 2, chans, seed=4)
        opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
        step = NativeTrainStep(net, opt, B, DEV) if kind == "native" else None
        losses = []
        for i in range(150):
            batch = data[i % 2]
            if step is None:
                _, loss = supervised_step(net, dict(batch), train=True, optimizer=opt, device=DEV)
            else:
                loss = float(step.step({k: v.to(DEV) for k, v in batch.items()})[0].item())
            losses.append(loss)
        firsts.append(np.mean(losses[:4]))
        finals.append(np.mean(losses[-10:]))
        assert np.isfinite(losses).all()
        if step is not None:
            step.close()
    assert finals[0] < 0.6 * firsts[0] and finals[1] < 0.6 * firsts[1], (firsts, finals)      # both learn (memorise) the rows
    assert abs(finals[0] - finals[1]) <= 0.1 * max(finals), (firsts, finals)


def test_c_abi_error_behaviour_of_the_trainer():
    """azx_train_* through ctypes: shapes the kernels do not cover are AZX_EINVAL at create, a step before the bind is
    AZX_ESTATE, a bind that lacks a tensor (or a parameter's momentum buffer) is AZX_EINVAL naming it -- never a
    silent fallback."""
    import ctypes as C
    from azalea_amd import _lib
    L = _lib.lib()

    def create(n, blocks, chans, batch):
        h = C.c_void_p()
        cfg = _lib.TrainConfig(n, blocks, chans, batch, 0)
        return L.azx_train_create(C.byref(cfg), C.byref(h)), h
    for bad in ((13, 6, 32, 8), (14, 6, 64, 8), (11, 6, 48, 8), (11, 0, 64, 8), (11, 6, 64, 0), (11, 20, 64, 8)):
        rc, _ = create(*bad)
        assert rc == -1, bad                                      # AZX_EINVAL
    rc, h = create(5, 1, 16, 4)
    assert rc == 0
    assert L.azx_train_step(h, 0.1, 0.9, 0.0, None) == -4        # AZX_ESTATE: nothing bound
    net = _net(5, 1, 16)
    sd = net.state_dict()
    names = [k for k in sd if k != "resblocks.0.bn2.running_var"]
    moms = {k: torch.zeros_like(p) for k, p in net.named_parameters()}

    def bind(names, with_momentum=True):
        n = len(names)
        return L.azx_train_bind(h, n, (C.c_char_p * n)(*[k.encode() for k in names]),
                                (C.c_void_p * n)(*[sd[k].data_ptr() for k in names]),
                                (C.c_int64 * n)(*[sd[k].numel() for k in names]),
                                (C.c_void_p * n)(*[(moms[k].data_ptr() if (k in moms and with_momentum) else None) for k in names]))
    assert bind(names) == -1 and b"resblocks.0.bn2.running_var" in L.azx_last_error()
    assert bind(list(sd), with_momentum=False) == -1 and b"momentum" in L.azx_last_error()
    assert bind(list(sd)) == 0
    assert L.azx_train_step(h, 0.0, 0.9, 0.0, None) == 0
    torch.cuda.synchronize()
    L.azx_train_destroy(h)
