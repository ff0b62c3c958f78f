"""Trainer pieces around the self-play path (azalea/policy_trainer.py), SURVEY 8(f).4.

`supervised_step` is the reference's inner step (policy_trainer.py:123-142) on stock
PyTorch-ROCm: forward + loss of network.py:92-102, backward, optimizer step.  `train` keeps the
reference loop's control flow (policy_trainer.py:23-119: SGD + StepLR, one `consume` per step,
checkpoints in the reference schema) and can keep the replay buffer in HBM
(`device_replay=True`: DeviceReplayBuffer, minibatches collated on the GPU); logging goes through
`logging` instead of the reference's tensorboard monitor, which is out of scope.
"""
import logging
import os
import time
from functools import partial

import numpy as np
import torch
from torch import optim
from torch.optim import lr_scheduler
from torch.utils.data import DataLoader

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


def initialize_replay_buffer(pool, game_factory, size: int) -> ReplayBuffer:
    """Fill a buffer with random-mover games (policy_trainer.py:145-158)."""
    player = Player(pool, [AzaleaAgent(game_factory)])
    examples, metrics = player.read(size)
    player.stop()
    buf = ReplayBuffer(examples)
    logging.info("replaybuf initialized with %s games and %d examples", metrics["games"], len(buf))
    return buf


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
    `history`: optional dict; receives the learning rate each epoch trained with under "lr"."""
    os.makedirs("%s/checkpoints" % rundir, exist_ok=True)
    np.random.seed(config["seed"])
    torch.manual_seed(config["seed"])
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(config["seed"])          # policy_trainer.py:33
    policy.seed(config["seed"])
    device = torch.device(config["device"])
    oversampling = config["replaybuf_oversampling"]
    batch_size = config["batch_size"]
    game_class = import_and_get(config["game"])
    game_factory = partial(game_class, board_size=config["board_size"])
    if replaybuf is None:
        replaybuf = initialize_replay_buffer(None, game_factory, config["replaybuf_size"])
    optimizer = optim.SGD(policy.net.parameters(), lr=config["lr_initial"], momentum=config["momentum"],
                          weight_decay=config["l2_regularization"])
    scheduler = lr_scheduler.StepLR(optimizer, step_size=config["lr_decay_epochs"], gamma=config["lr_decay"])
    policy.net.to(device)
    policy.net.train()
    policy.settings["move_exploration"] = True
    policy.settings["move_sampling"] = True
    agent = AzaleaAgent(game_factory, policy=policy, device=config["device"])
    player = Player(None, [agent], n_games=config.get("selfplay_games"))
    if device_replay:
        from .device_replay import DeviceReplayBuffer
        if not isinstance(replaybuf, DeviceReplayBuffer):
            replaybuf = DeviceReplayBuffer(player.device_engine(), len(replaybuf), replaybuf)
        batches = lambda: replaybuf.loader(batch_size)
    else:
        loader = DataLoader(replaybuf, batch_size=batch_size, shuffle=True, pin_memory=(device.type == "cuda"),
                            num_workers=config.get("num_dataloader_workers", 0), collate_fn=torch_batch_replays)
        batches = lambda: iter(loader)
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
        for batch in batches():
            batch = game_class.random_reflect(batch)
            output, loss_ = supervised_step(policy.net, batch, train=True, optimizer=optimizer, device=device)
            loss += loss_
            replaybuf.consume(batch_size / oversampling, player)
            if config.get("log_interval") and step % config["log_interval"] == 0:
                sps = config["log_interval"] / max(1e-9, time.time() - start_time)
                logging.info("step %d loss %.4f steps/sec %.2f", step, loss / config["log_interval"], sps)
                loss, start_time = 0.0, time.time()
            if config.get("model_checkpoint_interval") and step % config["model_checkpoint_interval"] == 0:
                save_checkpoint(policy, "%s/checkpoints/checkpoint.%d" % (rundir, step), optimizer=optimizer)
            step += 1
    player.stop()
    return save_checkpoint(policy, "%s/checkpoints/final" % rundir)
