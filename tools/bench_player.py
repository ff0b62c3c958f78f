#!/usr/bin/env python3
"""What the trainer calls, end to end, at BASELINE configs[2] (4096 games, 11x11, 400 sims, 6x64 net):

  Player.read(size)                 azalea/parallel_player.py:24-28  (replay_buffer.py:121-132 calls it)
  DeviceReplayBuffer.consume(n)     the HBM-resident counterpart (azalea_amd/device_replay.py)

Each call is split into: device time of the self-play (HIP events inside azx_play / azx_replay_fill),
the hand-over (k_rows_export + D2H copies, or k_replay_put), and the host frame construction
(rows_to_frame: the reference's struct-of-lists ReplayDataFrame).  The pool is first put into its steady
state the way bench.py does it (uniform-prior self-play, positions transplanted), so every engine move
finishes ~1 % of the games.  One JSON object on stdout.

    python tools/bench_player.py [--reads 6] [--size 4000] [--games 4096] [--moves 190]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=6)
    ap.add_argument("--size", type=int, default=4000, help="positions per Player.read / consume")
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--board", type=int, default=11)
    ap.add_argument("--sims", type=int, default=400)
    ap.add_argument("--blocks", type=int, default=6)
    ap.add_argument("--chans", type=int, default=64)
    ap.add_argument("--moves", type=int, default=190,
                    help="engine moves of the full-game-length legs (rows/s over the last 60); 0 = skip them")
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    from azalea_amd import AzaleaAgent, HexGame, Player, Policy
    from azalea_amd import parallel_player as pp
    from azalea_amd.device_replay import DeviceReplayBuffer

    n = args.board
    cfg = dict(device="cuda", network="HexNetwork", board_size=n, num_blocks=args.blocks, base_chans=args.chans,
               simulations=args.sims, search_batch_size=10, exploration_coef=0.5, exploration_depth=15,
               exploration_noise_alpha=0.03, exploration_noise_scale=0.25, exploration_temperature=1.0, seed=1)
    torch.manual_seed(0)
    policy = Policy()
    policy.initialize(cfg)
    policy.settings.update(move_sampling=True, move_exploration=True)
    agent = AzaleaAgent(lambda: HexGame(n), policy=policy, device="cuda")
    player = Player(None, [agent], n_games=args.games)
    E = player.device_engine()
    player.prepare_device_engine(E)

    # steady-state pool (untimed), as in bench.py
    bargs = argparse.Namespace(board=n, games=args.games, sims=args.sims, batch=10, noise_scale=0.25, blocks=args.blocks,
                               chans=args.chans, nodes_per_game=0, seed=0xBAD5EED5, desync=int(round(0.76 * n * n)),
                               settle=2 * n * n)
    T = bench.make_engine("tree", bargs, 0, 1, 0)
    bench.settle_pool(T, bargs, 0, 1, 0)
    start_positions = bench.pool_positions(T)
    E.reset(moves=start_positions)
    T.close()

    # instrument the three stages
    acc = {"play_wall": 0.0, "play_device": 0.0, "frame": 0.0}
    real_play, real_frame = E.play, pp.rows_to_frame

    def timed_play(*a, **k):
        t0 = time.perf_counter()
        rows, st = real_play(*a, **k)
        acc["play_wall"] += time.perf_counter() - t0
        acc["play_device"] += st["seconds"]
        return rows, st

    def timed_frame(rows):
        t0 = time.perf_counter()
        f = real_frame(rows)
        acc["frame"] += time.perf_counter() - t0
        return f
    E.play = timed_play
    pp.rows_to_frame = timed_frame

    player.read(1)                       # first harvest: allocations, first-touch of the host buffers
    for k in acc:
        acc[k] = 0.0
    reads = []
    rows_total = 0
    t_all = time.perf_counter()
    for _ in range(args.reads):
        t0 = time.perf_counter()
        frame, metrics = player.read(args.size)
        reads.append(time.perf_counter() - t0)
        rows_total += len(frame)
    wall = time.perf_counter() - t_all
    handover = acc["play_wall"] - acc["play_device"]
    out = {"config": "BASELINE configs[2]: %d games, %dx%d, %d sims, %dx%d net, steady-state pool" % (
               args.games, n, n, args.sims, args.blocks, args.chans),
           "player_read": {
               "reads": args.reads, "size": args.size, "rows": rows_total, "seconds": wall,
               "rows_per_sec": rows_total / wall,
               "engine_device_seconds": acc["play_device"],
               "handover_seconds (k_rows_export + D2H + queue polling)": handover,
               "frame_construction_seconds (rows_to_frame)": acc["frame"],
               "other_host_seconds": wall - acc["play_wall"] - acc["frame"],
               "host_overhead_frac": (wall - acc["play_device"]) / wall,
               "per_read_seconds": reads}}

    # the HBM-resident path: consume() refills the ring on the device
    E.play = real_play
    pp.rows_to_frame = real_frame
    buf = DeviceReplayBuffer(E, capacity=400000)
    buf.consume(1.0, player)
    dev_s, rows2 = 0.0, 0
    real_fill = E.replay_fill

    def timed_fill(*a, **k):
        nonlocal dev_s, rows2
        r, st = real_fill(*a, **k)
        dev_s += st["seconds"]
        rows2 += r
        return r, st
    E.replay_fill = timed_fill
    t0 = time.perf_counter()
    for _ in range(args.reads):
        buf.fresh_counter = 0
        buf.consume(float(args.size), player)
    wall2 = time.perf_counter() - t0
    t1 = time.perf_counter()
    batch = buf.sample(np.random.RandomState(0).randint(0, len(buf), 1024))
    torch.cuda.synchronize()
    t_sample = time.perf_counter() - t1
    out["device_replay_consume"] = {
        "consumes": args.reads, "rows": rows2, "seconds": wall2, "rows_per_sec": rows2 / wall2,
        "engine_device_seconds": dev_s, "host_overhead_frac": (wall2 - dev_s) / wall2,
        "collate_1024_seconds": t_sample, "batch_keys": sorted(batch)}
    player.stop()

    # ---- a whole game length after the transplant (round 3): rows/s over the last 60 moves ---------------------
    # Transplanted games only carry the rows played since the transplant, so the short runs above under-count;
    # after > 121 moves every game handed over is a whole one and rows/s must equal the engine's plies/s.
    if args.moves > 0:
        out["player_read_full_length"] = bench.run_api(argparse.Namespace(**dict(vars(bargs), api_moves=args.moves)), 0, 1, 0,
                                                       start_positions, torch)
        out["device_replay_consume_full_length"] = consume_full_length(args, bargs, start_positions, torch)
    print(json.dumps(out))


def consume_full_length(args, bargs, start, torch):
    """DeviceReplayBuffer.consume (rows never leave HBM) driven for `--moves` engine moves after the transplant."""
    from azalea_amd import AzaleaAgent, HexGame, Player, Policy
    from azalea_amd.device_replay import DeviceReplayBuffer
    n = args.board
    cfg = dict(device="cuda", network="HexNetwork", board_size=n, num_blocks=args.blocks, base_chans=args.chans,
               simulations=args.sims, search_batch_size=10, exploration_coef=0.5, exploration_depth=15,
               exploration_noise_alpha=0.03, exploration_noise_scale=0.25, exploration_temperature=1.0, seed=1)
    torch.manual_seed(0)
    policy = Policy()
    policy.initialize(cfg)
    policy.settings.update(move_sampling=True, move_exploration=True)
    agent = AzaleaAgent(lambda: HexGame(n), policy=policy, device="cuda")
    player = Player(None, [agent], n_games=args.games, gather=False)
    E = player.device_engine()
    player.prepare_device_engine(E)
    E.reset(moves=start)
    buf = DeviceReplayBuffer(E, capacity=1 << 20, shared=False)
    tot = {"plies": 0, "dev": 0.0, "rows": 0}
    real_fill = E.replay_fill

    def fill(*a, **k):
        r, st = real_fill(*a, **k)
        tot["plies"] += st["plies"]
        tot["dev"] += st["seconds"]
        tot["rows"] += r
        return r, st
    E.replay_fill = fill
    chunk = float(8 * args.games)
    t0 = time.perf_counter()
    log = [(0.0, 0, 0)]
    while tot["plies"] < args.moves * args.games:
        buf.fresh_counter = 0
        buf.consume(chunk, player)
        log.append((time.perf_counter() - t0, tot["rows"], tot["plies"]))
    player.stop()
    lo = next(i for i, r in enumerate(log) if r[2] >= max(0, args.moves - 60) * args.games or i == len(log) - 1)
    lo = min(lo, len(log) - 2)
    w0, w1 = log[lo], log[-1]
    dt = w1[0] - w0[0]
    wall = log[-1][0]
    return {"surface": "DeviceReplayBuffer.consume (replay_buffer.py:121-132 on the HBM ring)", "consumes": len(log) - 1,
            "moves_since_transplant": tot["plies"] / args.games, "rows": tot["rows"], "seconds": wall,
            "rows_per_sec": (w1[1] - w0[1]) / dt, "plies_per_sec": (w1[2] - w0[2]) / dt,
            "rows_over_plies": (w1[1] - w0[1]) / max(1, w1[2] - w0[2]),
            "window": "moves %.0f..%.0f after the transplant" % (w0[2] / args.games, w1[2] / args.games),
            "host_overhead_frac": (wall - tot["dev"]) / wall}


if __name__ == "__main__":
    main()
