"""Trainer pieces around the self-play path (azalea/policy_trainer.py), SURVEY 8(f).4.

`supervised_step` is the reference's inner step (policy_trainer.py:123-142) on stock
PyTorch-ROCm: forward + loss of network.py:92-102, backward, optimizer step.  `train` keeps the
reference loop's control flow (policy_trainer.py:23-119: SGD + StepLR, one `consume` per step,
checkpoints in the reference schema) and can keep the replay buffer in HBM
(`device_replay=True`: DeviceReplayBuffer, minibatches collated on the GPU); logging goes through
`logging` instead of the reference's tensorboard monitor, which is out of scope.
"""
import contextlib
import logging
import os
import time
from functools import partial

import numpy as np
import torch
from torch import optim
from torch.optim import lr_scheduler
from torch.utils.data import DataLoader

from . import distributed as azdist
from .azalea_agent import AzaleaAgent
from .parallel_player import Player
from .prep import torch_batch_replays
from .replay_buffer import ReplayBuffer
from .utils import import_and_get


def supervised_step(model, batch, *, train=False, optimizer=None, device="cpu"):
    """One batch through the network with its loss (policy_trainer.py:123-142)."""
    model.train(bool(train))
    with torch.set_grad_enabled(bool(train)):
        if train:
            optimizer.zero_grad()
        for k in batch:
            batch[k] = batch[k].to(device)
        output, loss = model.run(batch, compute_loss=True)
        if train:
            loss.backward()
            optimizer.step()
    return output, loss.item()


def embed_by_masks(model, board):
    """`model.encoder(board.long())` as (board == v) masks times the embedding matrix: [B,N,N] -> [B,4,N,N], the same
    values bit for bit (1*w + 0*w' + 0*w'' = w), with a backward that is a plain reduction."""
    w = model.encoder.weight
    x = sum((board == v).unsqueeze(-1).to(w.dtype) * w[v] for v in range(w.shape[0]))
    return x.permute(0, 3, 1, 2).contiguous()


class GraphedTrainStep:
    """`supervised_step(model, batch, train=True, optimizer=...)` captured once as a HIP graph and replayed.

    At the reference's batch of 128 the step is a chain of a few hundred small dependent kernels (3.6 ms eager on an
    MI355X, of which the GPU is busy a fraction); one graph launch per step runs the same kernels back to back
    (`tools/bench_train_step.py`).  Same arithmetic as the eager step with two differences that capture needs:
    * static shapes: `legal_moves` / `moves_prob` are padded with zeros to all N*N cells instead of the batch's widest
      row (padding logits are masked to -99 either way: exp(-99 - max) is below fp32's resolution of the softmax sum);
    * the 3 -> 4 embedding is applied as (board == v) masks times the embedding matrix: the same forward values bit
      for bit, and a backward that is a plain reduction instead of `embedding_dense_backward`, whose sort-based kernels
      size their work from the indices seen at capture time (replaying them on other boards reads out of bounds).
    The optimizer's hyper-parameters and buffer addresses are baked into the captured kernels, so the step is
    re-captured when any of them changes (the scheduler's rate, momentum / weight decay edits, a reloaded state).  Losses stay on the device: `step()` returns the three loss tensors of the last replay; read them
    (`.item()`) only when logging.  CUDA only; the batch size is fixed at construction."""

    def __init__(self, model, optimizer, batch_size: int, device):
        if torch.device(device).type != "cuda":
            raise ValueError("GraphedTrainStep needs a CUDA (ROCm) device")
        self.model, self.optimizer, self.B = model, optimizer, int(batch_size)
        self.device = torch.device(device)
        n = model.board_size
        self.cells = n * n
        dev = self.device
        self.board = torch.zeros((self.B, n, n), dtype=torch.int32, device=dev)
        self.legal_moves = torch.zeros((self.B, self.cells), dtype=torch.int32, device=dev)
        self.moves_prob = torch.zeros((self.B, self.cells), dtype=torch.float32, device=dev)
        self.reward = torch.zeros(self.B, dtype=torch.float32, device=dev)
        self.loss = torch.zeros(3, dtype=torch.float32, device=dev)       # total, value, moves
        self.out_value = torch.zeros(self.B, dtype=torch.float32, device=dev)
        self.out_logprob = torch.zeros((self.B, self.cells), dtype=torch.float32, device=dev)
        self._color = torch.zeros(self.B, dtype=torch.int64, device=dev)      # collate_into's other two outputs
        self._result = torch.zeros(self.B, dtype=torch.int64, device=dev)
        self.graph = None
        self.captured_key = None
        self.captures = 0
        self._warm = 0                   # eager steps run so far (optimizer state / allocator warm-up before capture)

    def _capture_key(self):
        """Everything the captured optimizer kernels have baked in: the hyper-parameters of every group and the
        addresses of the parameters, their gradients' owners and their momentum buffers (optimizer.load_state_dict
        replaces the buffers; replaying the old graph would keep updating the dead ones)."""
        key = []
        for g in self.optimizer.param_groups:
            key.append(tuple(g.get(k) for k in ("lr", "momentum", "weight_decay", "dampening", "nesterov", "maximize")))
            for p in g["params"]:
                buf = self.optimizer.state.get(p, {}).get("momentum_buffer")
                key.append((p.data_ptr(), None if buf is None else buf.data_ptr()))
        return key

    def invalidate(self):
        """Force a re-capture on the next step (after anything the capture key cannot see changed)."""
        self.graph = None

    def _forward(self):
        return self.model.forward_embedded(embed_by_masks(self.model, self.board), self.legal_moves)

    def _step(self):
        self.optimizer.zero_grad(set_to_none=True)
        out = self._forward()
        value_loss = torch.nn.functional.mse_loss(out["value"], self.reward)
        moves_loss = -(self.moves_prob * out["moves_logprob"]).sum() / self.B
        loss = value_loss + moves_loss
        loss.backward()
        self.optimizer.step()
        self.loss.copy_(torch.stack([loss.detach(), value_loss.detach(), moves_loss.detach()]))
        self.out_value.copy_(out["value"].detach())
        self.out_logprob.copy_(out["moves_logprob"].detach())

    def _load(self, batch):
        k = batch["legal_moves"].shape[1]
        if len(batch["reward"]) != self.B:
            raise ValueError("GraphedTrainStep was built for batches of %d rows, got %d" % (self.B, len(batch["reward"])))
        self.board.copy_(batch["board"].reshape(self.board.shape))
        self.reward.copy_(batch["reward"])
        self.legal_moves.zero_()
        self.moves_prob.zero_()
        self.legal_moves[:, :k].copy_(batch["legal_moves"])
        self.moves_prob[:, :k].copy_(batch["moves_prob"])

    def _capture(self):
        self.model.train(True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._step()
        self.captured_key = self._capture_key()
        self.captures += 1

    def step_from_ring(self, replaybuf, indices):
        """One training step on the rows `indices` of a DeviceReplayBuffer: azx_replay_collate writes them straight
        into the step's static input tensors (no batch tensors allocated, nothing copied).  Returns (loss tensors,
        widest row of the batch)."""
        k = replaybuf.collate_into(indices, dict(color=self._color, legal_moves=self.legal_moves, result=self._result,
                                                  board=self.board, moves_prob=self.moves_prob, reward=self.reward))
        return self._run(), k

    def step(self, batch):
        """One training step on `batch` (the dict `torch_batch_replays` / `DeviceReplayBuffer.sample` produce).
        The first three calls run eagerly on a side stream (optimizer state and allocator warm-up, as
        torch.cuda.graphs asks), the fourth captures."""
        self._load(batch)
        return self._run()

    def _run(self):
        if self.graph is not None and self._capture_key() != self.captured_key:
            self.graph = None          # the scheduler moved, or the optimizer was reloaded / re-configured: capture again
        if self.graph is None:
            self.model.train(True)
            if self.captures == 0 and self._warm < 3:
                side = torch.cuda.Stream(device=self.device)
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):
                    self._step()
                torch.cuda.current_stream(self.device).wait_stream(side)
                self._warm += 1
                return self.loss
            torch.cuda.synchronize(self.device)
            self._capture()        # capturing RUNS nothing: the captured step is this batch's, replayed below
        self.graph.replay()
        # a replay updates the parameters without going through the dispatcher, so their autograd version counters --
        # which Policy._sync_weights watches to know when to re-pack the engine's weights -- do not move: say so here
        self.model.weight_updates_outside_autograd = getattr(self.model, "weight_updates_outside_autograd", 0) + 1
        return self.loss

    def outputs(self, k: int):
        """(value, moves_logprob[:, :k]) of the last step, as `Network.run` returns them."""
        return {"value": self.out_value, "moves_logprob": self.out_logprob[:, :k]}


def to_mover_view(batch, game_class):
    """config["train_mover_view"] on a host batch: the rows of the second player (color == 1) as the search hands them
    to the network (mcts.py:178-181) -- game.flip_player_board_moves on board and legal_moves, moves_prob untouched
    (the moves keep their list positions).  In place; returns the batch."""
    mask = batch["color"] == 1
    if bool(mask.any()):
        board, moves = batch["board"], batch["legal_moves"]
        fb, fm = game_class.flip_player_board_moves(board[mask].cpu().numpy(), moves[mask].cpu().numpy())
        board[mask] = torch.as_tensor(fb, dtype=board.dtype, device=board.device)
        moves[mask] = torch.as_tensor(fm, dtype=moves.dtype, device=moves.device)
    return batch


def initialize_replay_buffer(pool, game_factory, size: int) -> ReplayBuffer:
    """Fill a buffer with random-mover games (policy_trainer.py:145-158)."""
    player = Player(pool, [AzaleaAgent(game_factory)])
    examples, metrics = player.read(size)
    player.stop()
    buf = ReplayBuffer(examples)
    logging.info("replaybuf initialized with %s games and %d examples", metrics["games"], len(buf))
    return buf


def serve_selfplay(player, replaybuf=None, timeout=None) -> int:
    """Ranks != 0 of a LOCK-STEP training job: play what rank 0 announces, with the weights rank 0 broadcasts, until
    it says stop (azalea_amd/distributed.py).  The deterministic mode; azalea_amd/actor_learner.py is the one in which
    self-play runs beside training.  Raises distributed.LeaderLost when rank 0 goes silent for `timeout` seconds
    (AZX_FOLLOW_TIMEOUT) or announces that it is aborting.  Returns the number of productions served."""
    served = 0
    while True:
        op, arg = azdist.follow(timeout)
        if op == azdist.OP_STOP:
            break
        if op == azdist.OP_READ:
            player.read(arg)                       # the rows are rank 0's to keep; every rank gets them anyway
        elif op == azdist.OP_REFILL:
            rows, _ = replaybuf.refill_shared(arg, player)
            replaybuf.fresh_counter += rows
        else:
            raise RuntimeError("serve_selfplay: unknown announcement %d" % op)
        served += 1
    pol = player._device_policy()
    if pol is not None:
        azdist.broadcast_weights(pol.net, src=0)   # everyone leaves with the trained network
    return served


def make_train_step(net, optimizer, batch_size, device, config):
    """The step train() runs full batches through, and its name.  By default the hand-written step
    (native_train.NativeTrainStep) wherever it applies -- HexNetwork, SGD, a board and width its kernels cover -- and
    the stock PyTorch step elsewhere: captured as a HIP graph when config["train_step_graph"], eager otherwise.
    config["train_step_native"]: False = never the hand-written step; True = it or a ValueError naming what is outside
    its envelope (raised here, before any rank has been told to wait for this one)."""
    want = config.get("train_step_native")
    if torch.device(device).type == "cuda" and want is not False:
        from .native_train import NativeTrainStep, unsupported_reason
        why = unsupported_reason(net, optimizer, device)
        if why is None:
            return NativeTrainStep(net, optimizer, batch_size, device), "native"
        if want:
            raise ValueError("train_step_native was asked for, but: " + why)
        logging.info("stock training step (%s)", why)
    if config.get("train_step_graph") and torch.device(device).type == "cuda":
        return GraphedTrainStep(net, optimizer, batch_size, device), "hip_graph"
    return None, "eager"


def save_checkpoint(policy, name, *, optimizer=None, replaybuf=None) -> str:
    """{'policy': ..., 'optimizer': ...} -> name.policy.pth (policy_trainer.py:161-181)."""
    state = {"policy": policy.state_dict()}
    if optimizer:
        state["optimizer"] = optimizer.state_dict()
    path = "%s.policy.pth" % name
    torch.save(state, path)
    logging.info("saved policy checkpoint to %s", path)
    if replaybuf:
        rpath = "%s.replaybuf.pth" % name
        torch.save(replaybuf.state_dict(), rpath)
        logging.info("saved replay buffer checkpoint to %s", rpath)
    return path


def train(policy, config, rundir, *, replaybuf=None, device_replay: bool = False, history=None) -> str:
    """The reference training loop (policy_trainer.py:23-119) over this package's Player.
    `history`: optional dict; receives the learning rate each epoch trained with under "lr", the name of the training
    step under "train_step" and, under torch.distributed, the self-play mode and the learner's / actor's counters.

    Under torch.distributed (one process per GPU) rank 0 is the trainer: it alone runs the optimizer and writes
    checkpoints; all ranks pass the same config and must call train() together.  config["selfplay_mode"]:
    * "actor_learner" (default): the other ranks play continuously into bounded backlogs with the last network rank 0
      broadcast (every config["weight_sync_steps"] steps, default 50); rank 0 does not play and pulls rows when its
      buffer asks for them (azalea_amd/actor_learner.py) -- the reference's in-flight self-play;
    * "lockstep": every shared production -- a Player.read or a device-ring refill -- is announced by rank 0, which
      broadcasts its network first, and played by all ranks; the others serve it (`serve_selfplay`).  Deterministic.
    Either way the other ranks return when rank 0 stops, holding the trained weights; if rank 0 fails they are told
    (distributed.LeaderLost) instead of being left in a collective.

    config["train_mover_view"] (default False = the reference's batches): train on every position in the view the
    search evaluates it in -- the second player's rows flipped to the first player's view (mcts.py:178-181), which
    the reference's trainer does not do (policy_trainer.py:84-85 feeds the rows' absolute boards).  DESIGN 8.6 has
    what that costs in playing strength."""
    os.makedirs("%s/checkpoints" % rundir, exist_ok=True)
    np.random.seed(config["seed"])
    torch.manual_seed(config["seed"])
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(config["seed"])          # policy_trainer.py:33
    policy.seed(config["seed"])
    device = torch.device(config["device"])
    if device.type == "cuda":
        torch.cuda.set_device(device)      # RCCL collectives (announcements, weights, records) run on the current device
    oversampling = config["replaybuf_oversampling"]
    batch_size = config["batch_size"]
    game_class = import_and_get(config["game"])
    game_factory = partial(game_class, board_size=config["board_size"])
    shared = azdist.is_distributed()
    leader = azdist.rank() == 0
    mode = None
    if shared:
        mode = config.get("selfplay_mode", "actor_learner")
        if mode not in ("actor_learner", "lockstep"):
            raise ValueError("selfplay_mode must be 'actor_learner' or 'lockstep'")
        azdist.control_group()             # collective: created where every rank is
    if replaybuf is None:
        # symmetric under torch.distributed: every rank plays its share of the random-mover games, all get all rows
        replaybuf = initialize_replay_buffer(None, game_factory, config["replaybuf_size"])
    optimizer = optim.SGD(policy.net.parameters(), lr=config["lr_initial"], momentum=config["momentum"],
                          weight_decay=config["l2_regularization"])
    scheduler = lr_scheduler.StepLR(optimizer, step_size=config["lr_decay_epochs"], gamma=config["lr_decay"])
    policy.net.to(device)
    policy.net.train()
    policy.settings["move_exploration"] = True
    policy.settings["move_sampling"] = True
    agent = AzaleaAgent(game_factory, policy=policy, device=config["device"])
    lockstep_role = ("leader" if leader else "follower") if mode == "lockstep" else None
    player = Player(None, [agent], n_games=config.get("selfplay_games"), role=lockstep_role)
    if mode == "actor_learner" and player._device_policy() is None:
        raise ValueError("selfplay_mode 'actor_learner' needs a Policy that holds a HexNetwork; use 'lockstep'")
    from_ring = False
    mover_view = bool(config.get("train_mover_view", False))
    if leader:
        logging.info("training batches: %s", "second player's rows flipped to the search's view (train_mover_view)"
                     if mover_view else "absolute boards, as the reference trains (config['train_mover_view'] = True "
                     "trains in the view the search evaluates in: DESIGN 8.6)")
    if device_replay:
        from .device_replay import DeviceReplayBuffer
        if not isinstance(replaybuf, DeviceReplayBuffer):
            replaybuf = DeviceReplayBuffer(player.device_engine(), len(replaybuf), replaybuf)
        replaybuf.mover_view = mover_view         # the collate kernel flips; host batches are flipped below
    elif mode == "actor_learner":
        player._agree_seed_base()          # a collective the actors make in device_engine(): rank 0 joins it here
    if history is not None and mode:
        history["selfplay_mode"] = mode
    if shared and not leader:
        if mode == "actor_learner":
            from .actor_learner import serve_selfplay_ahead
            stats = serve_selfplay_ahead(player, ahead_rows=config.get("selfplay_ahead_rows"),
                                         poll_plies=config.get("selfplay_poll_plies", 1))
            if history is not None:
                history["actor"] = stats
        else:
            serve_selfplay(player, replaybuf)
        player.stop()
        return "%s/checkpoints/final.policy.pth" % rundir      # written by rank 0

    # ---- rank 0 (or the only rank) from here on.  Under torch.distributed everything up to the final announcement
    # runs inside try: a failure -- a train step outside its envelope, AZX_ERANGE from a refill, SearchTreeFull, an
    # out-of-memory -- is announced (OP_ABORT) so that the other ranks leave instead of waiting in a broadcast
    learner = None
    ahead = None
    side = None
    streams = contextlib.ExitStack()       # the play-ahead mode's training stream is current from the first step to the last
    try:
        if mode == "actor_learner":
            from .actor_learner import Learner
            learner = Learner(policy.net, config.get("weight_sync_steps", 50))
            replaybuf.learner = player.learner = learner
        gstep, step_name = make_train_step(policy.net, optimizer, batch_size, device, config)
        logging.info("training step: %s", step_name)
        if history is not None:
            history["train_step"] = step_name
        if device_replay:
            batches = lambda: replaybuf.loader(batch_size)
            if gstep is not None:
                # the captured / hand-written step reads its rows straight from the ring: iterate the epoch's index
                # chunks, the same order loader() visits (random_reflect is the identity for Hex, hex.py:124-134)
                def index_chunks():
                    order = replaybuf.epoch_indices()
                    for s in range(0, len(order), batch_size):
                        yield order[s:s + batch_size]
                batches, from_ring = index_chunks, True
        else:
            loader = DataLoader(replaybuf, batch_size=batch_size, shuffle=True, pin_memory=(device.type == "cuda"),
                                num_workers=config.get("num_dataloader_workers", 0), collate_fn=torch_batch_replays)
            batches = lambda: iter(loader)
        if learner is not None:
            learner.sync_weights()         # the actors start playing with rank 0's network, whatever they were built with
        if config.get("selfplay_overlap"):
            # one GPU, one process: self-play runs WHILE the steps run (azalea_amd/play_ahead.py) -- a host thread keeps
            # the engine playing into a bounded backlog on a CU mask that leaves some CUs free, the steps run on a
            # high-priority stream and `consume` only waits when the backlog is short (process_pool.py:29-47 +
            # replay_buffer.py:121-132).  Not deterministic; the inline refill (the default) is.
            if shared or not device_replay or device.type != "cuda":
                raise ValueError("selfplay_overlap needs device_replay=True on one CUDA (ROCm) device, one process")
            from .play_ahead import PlayAhead
            ahead = PlayAhead(player, replaybuf.engine, ahead_rows=config.get("selfplay_ahead_rows"),
                              weight_sync_steps=config.get("weight_sync_steps", 50),
                              poll_plies=config.get("selfplay_poll_plies", 1),
                              reserve_cus=config.get("selfplay_reserve_cus", 0))
            replaybuf.ahead = ahead
            ahead.start()
            side = torch.cuda.Stream(device, priority=-1)
            side.wait_stream(torch.cuda.current_stream(device))
            if history is not None:
                history["selfplay_overlap"] = True
        if side is not None:
            streams.enter_context(torch.cuda.stream(side))
        loss_dev = None
        loss, step, start_time = 0.0, 0, time.time()
        for epoch in range(1, config["total_epochs"] + 1):
            # The reference calls scheduler.step() at the top of every epoch (policy_trainer.py:81) under the torch
            # it pins (0.4.1 / 1.0.1), where StepLR starts at last_epoch = -1: epoch e trains with
            # lr_initial * lr_decay ** ((e - 1) // lr_decay_epochs).  Modern torch counts the constructor as step 0,
            # so the same schedule is "no step before epoch 1, one step before every later epoch".
            if epoch > 1:
                scheduler.step()
            if history is not None:
                history.setdefault("lr", []).append(optimizer.param_groups[0]["lr"])
            for item in batches():
                batch = item
                if from_ring:                                        # item: a chunk of ring row indices
                    if len(item) == batch_size:
                        l3, _ = gstep.step_from_ring(replaybuf, item)
                        loss_dev = l3[0].clone() if loss_dev is None else loss_dev + l3[0]
                        batch = None
                    else:
                        batch = replaybuf.sample(item)              # the epoch's ragged last chunk: eager step below
                if batch is not None:
                    if mover_view and not device_replay:
                        batch = to_mover_view(batch, game_class)
                    batch = game_class.random_reflect(batch)
                    if gstep is not None and len(batch["reward"]) == batch_size:
                        l3 = gstep.step({k: v.to(device) for k, v in batch.items()})
                        loss_dev = l3[0].clone() if loss_dev is None else loss_dev + l3[0]
                    else:
                        output, loss_ = supervised_step(policy.net, batch, train=True, optimizer=optimizer, device=device)
                        loss += loss_
                if learner is not None:
                    learner.after_step()
                if ahead is not None:
                    ahead.after_step()
                replaybuf.consume(batch_size / oversampling, player)
                if config.get("log_interval") and step % config["log_interval"] == 0:
                    if loss_dev is not None:
                        loss, loss_dev = loss + float(loss_dev.item()), None
                    sps = config["log_interval"] / max(1e-9, time.time() - start_time)
                    logging.info("step %d loss %.4f steps/sec %.2f", step, loss / config["log_interval"], sps)
                    loss, start_time = 0.0, time.time()
                if config.get("model_checkpoint_interval") and step % config["model_checkpoint_interval"] == 0:
                    save_checkpoint(policy, "%s/checkpoints/checkpoint.%d" % (rundir, step), optimizer=optimizer)
                step += 1
        streams.close()
        if side is not None:
            torch.cuda.current_stream(device).wait_stream(side)
    except BaseException:
        streams.close()
        if ahead is not None:
            ahead.stop()
            replaybuf.ahead = None
        if shared:
            try:
                azdist.abort()
            except Exception:               # the control group itself is gone: the followers' timeout ends them
                logging.exception("could not announce the abort")
        raise
    if ahead is not None:
        ahead.stop()
        replaybuf.ahead = None
        if history is not None:
            history["play_ahead"] = ahead.counters()
    if history is not None and learner is not None:
        history["learner"] = dict(steps=learner.steps, pulls=learner.pulls, weight_syncs=learner.weight_syncs)
    if learner is not None:
        learner.stop()
    elif shared:
        azdist.lead(azdist.OP_STOP)
        azdist.broadcast_weights(policy.net, src=0)
    player.stop()
    return save_checkpoint(policy, "%s/checkpoints/final" % rundir)
