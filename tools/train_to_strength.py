#!/usr/bin/env python3
"""The whole loop end to end on one GPU, judged by playing strength: `policy_trainer.train` (the reference's loop,
policy_trainer.py:23-119: random-mover replay buffer, then a training step + `consume(batch / oversampling)` of fresh
self-play per step) with everything on the device -- self-play in throughput mode (k_play, device RNG), the HBM replay
ring, the hand-written training step, the device weight refresh -- and then a tournament (evaluation.py:17-36, all games
resident on the GPU: evaluate_batched) of the trained network against the network it started from, same search
settings, temperature-1 sampling for the first `exploration_depth` plies and no root noise.

    python3 tools/train_to_strength.py [--board 7] [--blocks 4] [--chans 32] [--epochs 6] [--rounds 200]

    ... --checkpoints 4     also: the reference's compare workflow (compare_cli.py:60-80) -- `train` saves a checkpoint
                            every steps / 4, `Policy.load` brings them back, a round robin (evaluate_batched) among
                            [start, checkpoints..., final] is ranked by `ranking.compute_ranking`: an Elo curve

    ... --world 3           the same under torch.distributed with 3 processes ON THIS ONE GPU (gloo; a functional soak of
                            the actor / learner topology, not a benchmark): rank 0 trains and pulls, ranks 1-2 play ahead

    ... --vs-shipped        also: the trained network against the REFERENCE'S SHIPPED MODEL (models/hex11-20180712-3362.policy.pth,
                            11x11 6x64: its tensors are the golden fixture tests/golden/g8_checkpoint.npz), same search
                            settings on both sides -- an absolute yardstick

Prints one JSON line: training seconds / steps / steps per second, rows of self-play consumed, and the tally."""
import argparse
import copy
import json
import logging
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def make_policy(args, seed):
    from azalea_amd.policy import Policy
    torch.manual_seed(seed)
    p = Policy()
    p.initialize(dict(device="cuda:0", network="HexNetwork", board_size=args.board, num_blocks=args.blocks,
                      base_chans=args.chans, simulations=args.sims, search_batch_size=10, exploration_coef=args.c,
                      exploration_depth=args.depth, exploration_noise_alpha=args.alpha, exploration_noise_scale=0.25,
                      exploration_temperature=1.0, seed=seed))
    return p


def tournament(policies, n, rounds):
    from azalea_amd import evaluation
    from azalea_amd.azalea_agent import AzaleaAgent
    from azalea_amd.game.hex import HexGame
    agents = []
    for p in policies:
        p.settings["move_sampling"] = True
        p.settings["move_exploration"] = False
        p.net.eval()
        agents.append(AzaleaAgent(lambda n=n: HexGame(n), policy=p, device="cuda:0"))
    out = evaluation.evaluate_batched(agents, rounds)
    return {"%d-%d" % k: [int(x) for x in v] for k, v in out.items()}


class LossLog(logging.Handler):
    """policy_trainer's "step %d loss %.4f steps/sec %.2f" lines (policy_trainer.py:101-107), kept as numbers."""

    def __init__(self):
        super().__init__(logging.INFO)
        self.rows = []

    def emit(self, record):
        if isinstance(record.msg, str) and record.msg.startswith("step %d loss"):
            self.rows.append([int(record.args[0]), round(float(record.args[1]), 4), round(float(record.args[2]), 1)])


def run(args):
    from azalea_amd.policy_trainer import train
    losses = LossLog()
    logging.getLogger().addHandler(losses)
    logging.getLogger().setLevel(logging.INFO)
    policy = make_policy(args, args.seed)
    start = copy.deepcopy(policy.net.state_dict())
    config = dict(seed=args.seed, device="cuda:0", game="azalea_amd.game.hex.HexGame", board_size=args.board,
                  replaybuf_size=args.replay, replaybuf_oversampling=args.oversampling, batch_size=128,
                  lr_initial=args.lr, lr_decay=args.lr_decay, lr_decay_epochs=args.lr_decay_epochs or max(1, args.epochs - 1), momentum=0.9,
                  l2_regularization=1e-4, total_epochs=args.epochs, selfplay_games=args.games, log_interval=args.log_interval,
                  model_checkpoint_interval=0, train_mover_view=args.mover_view)
    steps = args.epochs * (args.replay // 128 + (1 if args.replay % 128 else 0))
    every = steps // args.checkpoints if args.checkpoints else 0
    config["model_checkpoint_interval"] = every
    history = {}
    rundir = tempfile.mkdtemp(prefix="azx_strength_")
    t0 = time.perf_counter()
    if args.world > 1:
        config.update(selfplay_mode=args.selfplay_mode, weight_sync_steps=args.weight_sync_steps)
    if getattr(args, "overlap", False):        # self-play beside the steps on this one GPU (azalea_amd/play_ahead.py)
        config.update(selfplay_overlap=True, weight_sync_steps=args.weight_sync_steps)
    train(policy, config, rundir, device_replay=True, history=history)
    torch.cuda.synchronize()
    secs = time.perf_counter() - t0
    if args.world > 1:
        import torch.distributed as dist
        if dist.get_rank() != 0:
            return {"rank": dist.get_rank(), "actor": history.get("actor")}
    untrained = make_policy(args, args.seed)
    untrained.net.load_state_dict(start)
    untrained.net.to("cuda:0")
    tally = tournament([untrained, policy], args.board, args.rounds)
    w_old, draws, w_new = tally["0-1"]
    games = w_old + draws + w_new
    from azalea_amd import ranking
    try:                                   # ranking.py:46-58 on evaluate's tallies, as compare_cli.py:76 does
        elo = float(ranking.compute_ranking(2, {(0, 1): (w_old, draws, w_new)})[1])
    except ranking.RankingError:
        elo = None                         # a clean sweep has no finite maximum-likelihood score
    if elo is not None and (w_old == 0 or w_new == 0):
        elo = None
    shipped = None
    if args.vs_shipped:
        z = np.load(os.path.join(ROOT, "tests", "golden", "g8_checkpoint.npz"))
        if [int(x) for x in z["cfg"]] != [args.board, args.blocks, args.chans]:
            raise SystemExit("--vs-shipped needs the shipped model's shape: --board 11 --blocks 6 --chans 64")
        ref = make_policy(args, args.seed)
        ref.net.load_state_dict({k[2:]: torch.as_tensor(z[k]) for k in z.files if k.startswith("w:")})
        ref.net.to("cuda:0")
        t3 = tournament([ref, policy, untrained], args.board, args.rounds)
        shipped = {"tally_shipped_draw_trained": t3["0-1"], "tally_shipped_draw_start": t3["0-2"],
                   "trained_win_rate_vs_shipped": t3["0-1"][2] / max(1, sum(t3["0-1"])),
                   "start_win_rate_vs_shipped": t3["0-2"][2] / max(1, sum(t3["0-2"])),
                   "note": "the shipped model searched with THESE settings (its own: 800 simulations, c_puct 0.75)"}
        try:
            e3 = ranking.compute_ranking(3, {tuple(int(x) for x in k.split("-")): tuple(v) for k, v in t3.items()})
            shipped["elo_trained_minus_shipped"] = float(e3[1] - e3[0])
            shipped["elo_start_minus_shipped"] = float(e3[2] - e3[0])
        except ranking.RankingError:
            pass
    curve = None
    if args.checkpoints:
        from azalea_amd.policy import Policy
        marks = [k * every for k in range(1, args.checkpoints) if k * every < steps]
        loaded = [Policy.load("%s/checkpoints/checkpoint.%d.policy.pth" % (rundir, m), device="cuda:0") for m in marks]
        field = [untrained] + loaded + [policy]
        tallies = tournament(field, args.board, args.curve_rounds)
        outcomes = {tuple(int(x) for x in k.split("-")): tuple(v) for k, v in tallies.items()}
        try:
            elos = [round(float(x), 1) for x in ranking.compute_ranking(len(field), outcomes)]
        except ranking.RankingError:
            elos = None
        curve = {"steps": [0] + marks + [steps], "elo": elos, "games_per_pair": args.curve_rounds, "tallies": tallies}
    return {"what": "train() on one GPU, then trained vs starting network (agent 1 vs agent 0), %d games" % games,
            "net": "%dx%d on %dx%d" % (args.blocks, args.chans, args.board, args.board), "sims": args.sims,
            "train_step": history.get("train_step"), "train_mover_view": bool(args.mover_view), "epochs": args.epochs, "steps": steps, "train_seconds": secs,
            "steps_per_sec_incl_selfplay_and_fill": steps / secs, "selfplay_rows_consumed": steps * 128 / args.oversampling,
            "loss_by_step": losses.rows[1:],          # [step, mean loss over the interval, steps/s incl. self-play]
            "tally_untrained_draw_trained": [w_old, draws, w_new], "trained_win_rate": w_new / max(1, games),
            "trained_elo_over_start": elo, "elo_curve": curve, "vs_shipped_model": shipped,
            "world": args.world, "selfplay_mode": history.get("selfplay_mode"), "learner": history.get("learner"),
            "play_ahead": history.get("play_ahead")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--board", type=int, default=7)
    ap.add_argument("--blocks", type=int, default=4)
    ap.add_argument("--chans", type=int, default=32)
    ap.add_argument("--sims", type=int, default=100)
    ap.add_argument("--c", type=float, default=1.0)
    ap.add_argument("--depth", type=int, default=6)
    ap.add_argument("--alpha", type=float, default=0.3)
    ap.add_argument("--epochs", type=int, default=6)
    ap.add_argument("--replay", type=int, default=60000)
    ap.add_argument("--oversampling", type=float, default=4.0)
    ap.add_argument("--games", type=int, default=1024)
    ap.add_argument("--lr", type=float, default=0.05)
    ap.add_argument("--rounds", type=int, default=200)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--log-interval", type=int, default=5000)
    ap.add_argument("--checkpoints", type=int, default=0, help="save this many evenly spaced checkpoints and rank them")
    ap.add_argument("--curve-rounds", type=int, default=40, help="games per pair of the ranking round robin")
    ap.add_argument("--vs-shipped", action="store_true")
    ap.add_argument("--lr-decay", type=float, default=0.1)
    ap.add_argument("--lr-decay-epochs", type=int, default=0, help="StepLR period in epochs (default: only the last epoch decays)")
    ap.add_argument("--mover-view", action="store_true", help="config['train_mover_view']: not the reference's batches")
    ap.add_argument("--overlap", action="store_true", help="config['selfplay_overlap']: self-play runs while the steps run")
    ap.add_argument("--world", type=int, default=1)
    ap.add_argument("--selfplay-mode", default="actor_learner")
    ap.add_argument("--weight-sync-steps", type=int, default=50)
    args = ap.parse_args()
    if args.world == 1:
        print(json.dumps(run(args)))
        return
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_rank_main, args=(args, port), nprocs=args.world, join=True)


def _rank_main(rank, args, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=args.world)
    out = run(args)
    print(json.dumps(out), file=(sys.stdout if rank == 0 else sys.stderr), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
