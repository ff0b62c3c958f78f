#!/usr/bin/env python3
"""The trainer's loop as one GPU runs it (policy_trainer.py:82-90: a training step, then
`replaybuf.consume(batch_size / oversampling, player)`), at the reference's hyper-parameters (6x64 on 11x11, batch 128,
10x oversampling, 400-sim self-play on 4096 concurrent games): the captured training step fed from the HBM ring, with
the refills played inline (DeviceReplayBuffer.consume), and the same steps without any refill -- with the hand-written
step (NativeTrainStep) and with the stock kernels captured as a HIP graph (GraphedTrainStep).
Reports steps/s and the rows the refills brought.  (Round 3 also measured the refills played UNDER the steps by a
background thread on a second engine handle: 344.7 vs 339.6 steps/s, profiles/r3_train_loop_bench.json -- the tower
fills every CU's registers and LDS, so the training kernels queue behind its blocks.)  Round 6: `overlapped` -- the
play-ahead mode of azalea_amd/play_ahead.py: a host thread keeps the engine playing into a bounded backlog on a CU mask
that leaves `--reserve-cus` CUs of every XCD free, the steps run on a high-priority stream beside it and `consume` takes
chunks out of the backlog, waiting only when it is short.
    python tools/bench_train_loop.py [--steps 1200] [--games 4096] [--sims 400] [--modes inline,steps,overlapped]
                                     [--reserve-cus 4] [--priority high|normal]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from torch import optim

from azalea_amd import AzaleaAgent, HexGame, Player, Policy
from azalea_amd.device_replay import DeviceReplayBuffer
from azalea_amd.native_train import NativeTrainStep
from azalea_amd.policy_trainer import GraphedTrainStep


def run(mode, args, step_kind="native"):
    dev = "cuda:0"
    torch.manual_seed(0)
    policy = Policy()
    policy.initialize(dict(device=dev, network="HexNetwork", board_size=11, num_blocks=6, base_chans=64,
                           simulations=args.sims, search_batch_size=10, exploration_coef=0.5, exploration_depth=15,
                           exploration_noise_alpha=0.03, exploration_noise_scale=0.25, exploration_temperature=1.0, seed=1))
    policy.net.to(dev).train()
    policy.settings.update(move_sampling=True, move_exploration=True)
    agent = AzaleaAgent(lambda: HexGame(11), policy=policy, device=dev)
    player = Player(None, [agent], n_games=args.games, gather=False)
    E = player.device_engine()
    buf = DeviceReplayBuffer(E, 100000, shared=False)
    from azalea_amd import engine as eng
    # the pool in steady state, as in bench.py: seeded random legal positions of 0..92 plies, then an untimed fill of
    # two game lengths so that whole games are being handed over; the accounting starts from zero after it
    player.prepare_device_engine(E)
    E.reset(moves=eng.random_prefixes(11, np.arange(args.games), 92, 1))
    buf.consume(args.fill, player)
    buf.fresh_counter = 0
    opt = optim.SGD(policy.net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    B = 128
    gs = (NativeTrainStep if step_kind == "native" else GraphedTrainStep)(policy.net, opt, B, torch.device(dev))
    refills, rows = [], 0
    order = np.random.RandomState(0).randint(0, len(buf), (args.steps + 20, B))
    ahead, ctx = None, None
    if mode == "overlapped":
        from azalea_amd.play_ahead import PlayAhead
        ahead = PlayAhead(player, E, ahead_rows=args.ahead_rows or None, weight_sync_steps=args.weight_sync_steps,
                          reserve_cus=args.reserve_cus)
        buf.ahead = ahead
        ahead.start()
        side = torch.cuda.Stream(dev, priority=-1 if args.priority == "high" else 0)
        side.wait_stream(torch.cuda.current_stream())
        ctx = torch.cuda.stream(side)
        ctx.__enter__()
    warm = 20 if mode != "overlapped" else 400          # overlapped: past the first takes, the backlog in its steady state
    waits0 = 0.0
    for i in range(args.steps + warm):
        if i == warm:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            refills, rows = [], 0
            waits0 = ahead.stats["wait_seconds"] if ahead else 0.0
        gs.step_from_ring(buf, order[i % len(order)] % len(buf))
        if ahead is not None:
            ahead.after_step()
        m = buf.consume(B / 10.0, player) if mode in ("inline", "overlapped") else None
        if m:
            refills.append((time.perf_counter(), i))
            rows += int(m["moves_per_game"])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    out = {"mode": mode, "step": step_kind, "steps": args.steps, "seconds": t1 - t0, "steps_per_sec": args.steps / (t1 - t0),
           "refills": len(refills), "rows_refilled": rows}
    # A refill brings ~one pool move's harvest (~4 000 rows = ~320 steps' worth) and costs a third of a second: whether
    # 4 or 5 of them fall into a 1 600-step window moves steps/s by 15 %.  Over WHOLE refill cycles -- from the end of
    # the first refill of the window to the end of the last -- the rate has no such quantisation.
    if len(refills) >= 3:
        (ta, ia), (tb, ib) = refills[0], refills[-1]
        out["steps_per_sec_whole_cycles"] = (ib - ia) / (tb - ta)
        out["cycles"] = len(refills) - 1
    if ahead is not None:
        ctx.__exit__(None, None, None)
        ahead.stop()
        buf.ahead = None
        c = ahead.counters()
        out.update({"play_ahead": c, "wait_share": (c["wait_seconds"] - waits0) / (t1 - t0), "reserve_cus_per_xcd": args.reserve_cus,
                    "priority": args.priority})
    player.stop()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4800)
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--sims", type=int, default=400)
    ap.add_argument("--fill", type=int, default=60000, help="rows of the untimed initial fill")
    ap.add_argument("--modes", default="inline,steps,overlapped,inline_graph,steps_graph")
    ap.add_argument("--reserve-cus", type=int, default=0, help="overlapped: CUs of every XCD the engine leaves free")
    ap.add_argument("--priority", choices=["high", "normal"], default="high", help="overlapped: the training stream's priority")
    ap.add_argument("--ahead-rows", type=int, default=0, help="overlapped: backlog bound (default: one row per pool slot)")
    ap.add_argument("--weight-sync-steps", type=int, default=50)
    ap.add_argument("--one", default=None, help="internal: run this one mode in this process")
    args = ap.parse_args()
    table = {"inline": ("inline", "native"), "steps": ("steps only", "native"), "overlapped": ("overlapped", "native"),
             "inline_graph": ("inline", "hip_graph"), "steps_graph": ("steps only", "hip_graph")}
    if args.one:          # internal: ONE mode in this process
        print(json.dumps(run(table[args.one][0], args, table[args.one][1])))
        return
    # every mode in a process of its own: a mode leaves state behind (side streams, allocator pools, a trainer's graph)
    # that costs whichever runs next up to 2x (an inline run behind an overlapped one in one process: 346 vs 657 steps/s)
    import subprocess
    res = {}
    for m in args.modes.split(","):
        cmd = [sys.executable, os.path.abspath(__file__), "--one", m] + [a for a in sys.argv[1:]]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=1800)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode or not lines:
            raise SystemExit("mode %s failed: %s" % (m, (r.stderr or r.stdout)[-800:]))
        res[m] = json.loads(lines[-1])
    out = {"what": "training step (batch 128; hand-written / stock kernels captured as a HIP graph) + consume(12.8) per "
                   "step; 6x64 resnet self-play, %d games, %d sims" % (args.games, args.sims), "runs": list(res.values())}
    sps = {m: r.get("steps_per_sec_whole_cycles", r["steps_per_sec"]) for m, r in res.items()}
    if "inline" in sps and "steps" in sps:
        out["selfplay_share_of_loop_native"] = 1.0 - sps["inline"] / sps["steps"]
    if "inline_graph" in sps and "steps_graph" in sps:
        out["selfplay_share_of_loop_hip_graph"] = 1.0 - sps["inline_graph"] / sps["steps_graph"]
    if "inline" in sps and "inline_graph" in sps:
        out["loop_speedup_native_vs_hip_graph"] = sps["inline"] / sps["inline_graph"]
    if "inline" in sps and "overlapped" in sps:
        out["overlap_speedup"] = sps["overlapped"] / sps["inline"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
