"""The N>1 replay path with REAL engines: two processes (ranks) share the one GPU of the test box, each with
its own engine handle; torch.distributed runs over gloo (RCCL needs one GPU per rank), so the record
all-gather takes the CPU fallback -- everything else is the production path: global-game-index sharding,
Player.read's shared refill, DeviceReplayBuffer.consume packing on the device and appending to the ring."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from azalea_amd import AzaleaAgent, HexGame, Player, Policy
    from azalea_amd.device_replay import DeviceReplayBuffer
    fails = []
    n = 5
    cfg = dict(device="cuda:0", network="HexNetwork", board_size=n, num_blocks=1, base_chans=64, simulations=20,
               search_batch_size=10, exploration_coef=0.5, exploration_depth=3, exploration_noise_alpha=0.3,
               exploration_noise_scale=0.25, exploration_temperature=1.0, seed=100 + rank)   # ranks seeded differently
    torch.manual_seed(0)                                  # same weights everywhere
    policy = Policy()
    policy.initialize(cfg)
    policy.settings.update(move_sampling=True, move_exploration=True)
    agent = AzaleaAgent(lambda: HexGame(n), policy=policy, device="cuda:0")
    # a first read smaller than the world: rank 1's quota is 0 and it creates no engine -- the shared seed base
    # is agreed at the top of read(), so nobody waits in a broadcast the other rank never enters
    early = Player(None, [agent], n_games=16)
    f0, m0 = early.read(1)
    n0 = [None] * world
    dist.all_gather_object(n0, len(f0))
    fails += [] if (n0[0] == n0[1] and len(f0) >= 1 and m0["games"] >= 1) else [10]
    early.stop()
    player = Player(None, [agent], n_games=16)
    E = player.device_engine()
    # rank r of W plays the global games r, r + W, ...; the seed base is rank 0's
    fails += [] if (E.cfg.game_index_stride, E.cfg.game_index_offset) == (world, rank) else [1]
    seeds = [None] * world
    dist.all_gather_object(seeds, int(E.cfg.seed))
    fails += [] if seeds[0] == seeds[1] else [2]

    # Player.read: every rank plays its share, every rank gets all rows in rank order
    frame, metrics = player.read(120)
    sig = np.array([len(frame), int(sum(int(s.board.sum()) for s in frame.state)),
                    int(round(1e3 * float(sum(frame.reward))))], np.int64)
    sigs = [torch.zeros(3, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sigs, torch.from_numpy(sig))
    fails += [] if (len(frame) >= 120 and torch.equal(sigs[0], sigs[1])) else [3]
    fails += [] if metrics["moves_per_game"] == len(frame) and metrics["games"] >= 2 else [4]
    for i in range(len(frame)):                           # rows are well-formed positions
        st = frame.state[i]
        if len(st.legal_moves) != int((st.board == 0).sum()) or abs(float(frame.moves_prob[i].sum()) - 1.0) > 1e-5:
            fails += [5]
            break

    # DeviceReplayBuffer.consume: packed on the device, gathered, appended to every rank's HBM ring
    buf = DeviceReplayBuffer(E, capacity=2000)
    m = buf.consume(100.0, player)
    counts = buf.last_exchange["rows_per_rank"]
    fails += [] if (len(counts) == world and all(c >= 50 for c in counts) and len(buf) == sum(counts)) else [6]
    fails += [] if (buf.fresh_counter == sum(counts) - 100 and m["games"] >= 2) else [7]
    rows = buf.rows()
    digest = torch.tensor([int(rows["board"].astype(np.int64).sum()), int(rows["legal_moves"].astype(np.int64).sum()),
                           int(round(1e3 * float(rows["reward"].sum()))), len(rows["reward"])], dtype=torch.int64)
    both = [torch.zeros(4, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(both, digest)
    fails += [] if torch.equal(both[0], both[1]) else [8]
    k = (rows["board"].reshape(len(rows["reward"]), -1) == 0).sum(1)
    fails += [] if np.array_equal(k, (rows["legal_moves"] > 0).sum(1)) else [9]
    player.stop()
    out[rank] = fails
    dist.destroy_process_group()


def test_two_ranks_with_engines_share_reads_and_the_device_ring():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: [], 1: []}      # numbers of the checks that failed, per rank


def _bench(args, world):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AZX_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable]
    if world > 1:      # the driver's launch line; the launcher is a fresh child process (nothing here re-execs)
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())]
    cmd += [os.path.join(root, "bench.py"), "--gpus", str(world)] + args
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-3000:]
    return json.loads(lines[0])


def test_bench_two_ranks_play_the_same_games_as_one():
    """bench.py --gpus 2 the way the driver launches it (torch.distributed.run, one process per rank; gloo here
    because both ranks share the box's one GPU): the ranks shard the pool by global game index, so two ranks with
    32 slots each play exactly the games one rank with 64 slots plays -- the replay exchange after the same number
    of moves hands over the same set of game uids -- and the whole-job sims are the sum over ranks."""
    common = ["--steps", "3", "--warmup", "1", "--sims", "40", "--board", "7", "--blocks", "1", "--workload", "resnet",
              "--no-cpu-baseline", "--exchange-plies", "12", "--settle", "40"]
    one = _bench(common + ["--games", "64"], 1)
    two = _bench(common + ["--games", "32"], 2)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["world"]["backend"] == "gloo"
    x1, x2 = one["replay_allgather"], two["replay_allgather"]
    assert x1["ranks"] == 1 and x2["ranks"] == 2 and len(x2["rows_per_rank"]) == 2 and min(x2["rows_per_rank"]) >= 1
    assert two["world"]["rows_per_rank"] == x2["rows_per_rank"]
    assert x1["games"] >= 4 and x1["game_uids"] == x2["game_uids"]          # the same games, whoever played them
    assert sum(x1["rows_per_rank"]) == sum(x2["rows_per_rank"])
    # 64 games x 3 moves x 50 select_leaf calls in both jobs: value = whole-job sims / max-over-ranks time
    assert round(one["value"] * one["elapsed_s"]) == 64 * 3 * 50 == round(two["value"] * two["elapsed_s"])
    assert one["plies"] == two["plies"] == 64 * 3 and one["games_finished"] == two["games_finished"]
