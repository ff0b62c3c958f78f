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


def test_bench_eight_ranks_dry_run():
    """the driver's N=8 launch line (`python -m torch.distributed.run --nproc-per-node 8 bench.py
    --gpus 8 ...`) with all eight ranks sharing this box's one GPU over gloo: 8 x 8 slots play exactly the games of one
    rank with 64 slots (global game index = slot-game * 8 + rank), the replay exchange carries eight row counts, and
    the whole-job value is the sum over ranks."""
    common = ["--steps", "3", "--warmup", "1", "--sims", "40", "--board", "7", "--blocks", "1", "--workload", "resnet",
              "--no-cpu-baseline", "--exchange-plies", "12", "--settle", "40"]
    one = _bench(common + ["--games", "64"], 1)
    eight = _bench(common + ["--games", "8", "--ref-n1", repr(one["value"])], 8)
    assert eight["n_gpus"] == 8 and eight["world"]["ranks"] == 8 and eight["world"]["backend"] == "gloo"
    # the N > 1 line checks itself, as scalars of the dict the driver's record keeps: how many DISTINCT
    # devices the ranks ran on (here one, shared, over gloo -> rccl_ranks 0: no scaling claim can be read off this run),
    # the per-rank spread, the all-gather's rate, the efficiency against the N = 1 value handed in
    r = eight["roofline"]
    assert r["world_ranks"] == 8 and r["world_backend"] == "gloo" and r["world_distinct_devices"] == 1 and r["rccl_ranks"] == 0
    assert 0 < r["world_rank_sims_per_sec_min"] <= r["world_rank_sims_per_sec_max"] and 0 < r["world_slowest_over_fastest"] <= 1
    assert r["world_rank_sims_per_sec_sum"] >= eight["value"] * (1 - 1e-9)      # whole-job value: total work / slowest rank's time
    assert r["replay_allgather_bytes"] == eight["replay_allgather"]["bytes_gathered"] and r["replay_allgather_gbs"] > 0
    assert abs(r["weak_scaling_eff"] - eight["value"] / (8 * one["value"])) < 1e-9 and r["world_ref_n1_sims_per_sec"] == one["value"]
    assert len(eight["world"]["per_rank"]) == 8 and "tree" not in eight and "config5" not in eight
    x1, x8 = one["replay_allgather"], eight["replay_allgather"]
    assert x8["ranks"] == 8 and len(x8["rows_per_rank"]) == 8 and eight["world"]["rows_per_rank"] == x8["rows_per_rank"]
    assert x1["games"] >= 4 and x1["game_uids"] == x8["game_uids"]
    assert sum(x1["rows_per_rank"]) == sum(x8["rows_per_rank"])
    assert round(one["value"] * one["elapsed_s"]) == 64 * 3 * 50 == round(eight["value"] * eight["elapsed_s"])
    assert one["plies"] == eight["plies"] == 64 * 3 and one["games_finished"] == eight["games_finished"]


def _module_digest(net):
    import hashlib
    h = hashlib.sha256()
    for k, v in sorted(net.state_dict().items()):
        if v.dtype == torch.float32:
            h.update(k.encode())
            h.update(v.detach().cpu().numpy().tobytes())
    return h.hexdigest()


def _train_worker(rank, world, port, rundir, native, mode, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["AZX_FOLLOW_TIMEOUT"] = "300"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import glob
    from azalea_amd import Policy
    from azalea_amd import actor_learner as al
    from azalea_amd import device_replay as dr
    from azalea_amd import distributed as azd
    from azalea_amd import parallel_player as pp
    from azalea_amd.policy_trainer import train
    fails = []
    n = 5
    cfg = dict(device="cuda:0", network="HexNetwork", board_size=n, num_blocks=1, base_chans=64, simulations=20,
               search_batch_size=10, exploration_coef=0.5, exploration_depth=3, exploration_noise_alpha=0.3,
               exploration_noise_scale=0.25, exploration_temperature=1.0, seed=3)
    torch.manual_seed(100 + rank)                       # the ranks START from different networks
    policy = Policy()
    policy.initialize(cfg)
    digests, packed = [], []
    real_fill, real_sync, real_prep = dr.DeviceReplayBuffer._fill_shared, al.Learner.sync_weights, pp.Player.prepare_device_engine

    def fill(self, refill):                             # lock-step: the engine's packed weights at every shared refill
        digests.append(self.engine.weights_digest())
        return real_fill(self, refill)

    def sync(self):                                     # actor / learner, rank 0: the module it has just broadcast
        real_sync(self)
        digests.append(_module_digest(self.net))

    def prep(self, engine):                             # actor / learner, rank 1: the module it packs after a broadcast
        real_prep(self, engine)
        if mode == "actor_learner" and rank != 0:
            digests.append(_module_digest(self._device_policy().net))
            packed.append(engine.weights_digest())
    dr.DeviceReplayBuffer._fill_shared, al.Learner.sync_weights, pp.Player.prepare_device_engine = fill, sync, prep
    tcfg = dict(seed=11, device="cuda:0", replaybuf_oversampling=2, batch_size=16, game="azalea_amd.game.hex.HexGame",
                board_size=n, replaybuf_size=320, lr_initial=0.05, momentum=0.9, l2_regularization=1e-4,
                lr_decay_epochs=100, lr_decay=0.1, total_epochs=1, selfplay_games=16, log_interval=0,
                model_checkpoint_interval=8, selfplay_mode=mode, weight_sync_steps=4, selfplay_ahead_rows=64)
    if not native:
        tcfg["train_step_native"] = False              # native=True: the DEFAULT picks the hand-written step (1x64 on 5x5)
    hist = {}
    try:
        path = train(policy, tcfg, rundir, device_replay=True, history=hist)
    finally:
        dr.DeviceReplayBuffer._fill_shared, al.Learner.sync_weights, pp.Player.prepare_device_engine = real_fill, real_sync, real_prep
    alld = [None] * world
    dist.all_gather_object(alld, digests)
    if mode == "lockstep":
        # 20 steps x 8 rows consumed: several shared refills, each searched with rank 0's weights of that moment on BOTH
        # ranks (the digests are of the engines' packed operands), and those weights moved as rank 0 trained
        # (how many refills 20 steps need depends on the lengths of the games played, i.e. on weights that carry the
        # native step's last-bit run-to-run differences -- its head sums are float atomics: usually 3-4, once in a while 2)
        fails += [] if (len(digests) >= 2 and alld[0] == alld[1]) else [1]
    else:
        # 20 steps, a broadcast before the first and after every 4th: the module rank 1 packed after each broadcast is
        # the one rank 0 had just sent; rank 1 played between announcements and handed its rows over when asked
        fails += [] if (len(alld[0]) >= 1 + 20 // 4 and alld[0] == alld[1]) else [1]
        if rank == 0:
            L = hist["learner"]      # >= 20 steps: the initial buffer holds whole games, so a few rows more than 320
            fails += [] if (hist["train_step"] == ("native" if native else "eager") and L["pulls"] >= 2
                            and 20 <= L["steps"] <= 24 and len(digests) == L["weight_syncs"] == 1 + L["steps"] // 4) else [6]
            if fails:
                print("learner", hist, len(digests), flush=True)
        else:
            A = hist["actor"]
            fails += [] if (A["pulls"] >= 2 and A["rows"] >= 100 and A["max_productions_between_announcements"] >= 1) else [7]
            fails += [] if len(set(packed)) == len(packed) >= 3 else [8]     # the packed operands moved with every broadcast
    fails += [] if len(set(alld[0])) >= (2 if mode == "lockstep" else 3) else [2]
    sd = torch.cat([t.detach().reshape(-1).double().cpu() for t in policy.net.state_dict().values() if t.is_floating_point()])
    sums = [None] * world
    dist.all_gather_object(sums, float(sd.sum()))
    fails += [] if sums[0] == sums[1] else [3]        # both leave with the trained network
    dist.barrier()
    files = sorted(os.path.basename(f) for f in glob.glob(os.path.join(rundir, "checkpoints", "*.policy.pth")))
    fails += [] if files == ["checkpoint.0.policy.pth", "checkpoint.16.policy.pth", "checkpoint.8.policy.pth",
                             "final.policy.pth"] else [4]
    if rank == 0:
        state = torch.load(path, weights_only=False)["policy"]["net"]
        fails += [] if all(torch.equal(state[k].cpu(), v.cpu()) for k, v in policy.net.state_dict().items()) else [5]
    out[rank] = fails
    azd.reset_control_group()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["lockstep", "actor_learner"])
@pytest.mark.parametrize("native", [False, True])
def test_train_two_ranks_rank0_trains_both_play_with_its_weights(tmp_path, native, mode):
    """policy_trainer.train with world 2 (gloo, both ranks on this GPU, real engines, the HBM replay ring).
    lockstep: rank 0 runs the optimizer -- the eager step, or the hand-written one, which is the default where it
    applies -- and broadcasts its network before every shared refill, rank 1 serves self-play; identical engine weight
    digests at every refill.  actor_learner (the default mode): rank 1 plays ahead into its backlog between rank 0's
    announcements, rank 0 never plays, pulls rows and broadcasts every 4 steps.  Either way one set of checkpoint
    files and both ranks leave with the trained network."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_train_worker, args=(world, _free_port(), str(tmp_path / "run"), native, mode, out), nprocs=world, join=True)
    assert dict(out) == {0: [], 1: []}


def test_rccl_single_rank_carries_the_records_and_the_weights():
    """RCCL itself (backend "nccl"), world 1 -- all this box's one GPU allows: the record all-gather, the announcement
    and the weight broadcast run through the RCCL communicator on device tensors, and come back unchanged."""
    port = _free_port()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from azalea_amd import distributed as azd
        from azalea_amd.network import HexNetwork
        rb = azd.record_bytes(25)
        rec = torch.randint(0, 255, (37, rb), dtype=torch.uint8, device="cuda:0")
        parts, counts = azd.all_gather_records(rec)
        assert counts == [37] and torch.equal(parts[0], rec) and parts[0].is_cuda
        empty, c0 = azd.all_gather_records(rec[:0])
        assert c0 == [0] and empty[0].shape[0] == 0
        parts, counts = azd.gather_records(rec, dst=0)              # the actor / learner pull's gather to the learner
        assert counts == [37] and torch.equal(parts[0], rec) and parts[0].is_cuda
        t = torch.tensor([azd.OP_REFILL, 77], dtype=torch.int64, device="cuda:0")
        dist.broadcast(t, src=0)
        assert t.tolist() == [azd.OP_REFILL, 77]
        # announcements travel on a gloo group created beside the RCCL one (pollable, with our own timeout)
        g = azd.control_group()
        assert dist.get_backend(g) == "gloo"
        c = torch.tensor([azd.OP_PULL, 5], dtype=torch.int64)
        w = dist.broadcast(c, src=0, group=g, async_op=True)
        w.wait()
        assert w.is_completed() and c.tolist() == [azd.OP_PULL, 5]
        azd.reset_control_group()
        net = HexNetwork(board_size=5, num_blocks=1, base_chans=64).to("cuda:0")
        before = [v.clone() for v in net.state_dict().values()]
        flat = torch.cat([v.detach().reshape(-1).float() for v in net.state_dict().values() if v.is_floating_point()])
        dist.broadcast(flat, src=0)                    # what broadcast_weights does per network
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(before, net.state_dict().values()))
    finally:
        dist.destroy_process_group()
