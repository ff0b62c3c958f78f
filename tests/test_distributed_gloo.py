"""The N>1 path on CPU: world_size-2 gloo processes exchange replay rows (all-gather of
fixed-size records), sum metrics and broadcast weights exactly as the RCCL path does."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rows(rank, count, n=5):
    rng = np.random.RandomState(100 + rank)
    cells = n * n
    return dict(board=rng.randint(0, 3, (count, n, n)).astype(np.int32),
                color=rng.randint(0, 2, count).astype(np.int32),
                nlegal=rng.randint(1, cells, count).astype(np.int32),
                moves_prob=rng.rand(count, cells).astype(np.float32),
                reward=rng.choice([-1.0, 1.0], count).astype(np.float32),
                game_uid=(np.arange(count) + (rank << 40)).astype(np.int64))


class StubEngine:
    """Stands in for azalea_amd.engine.Engine on the CPU: the harvest queue and the replay ring are numpy
    tables of fixed-size records and the 'device' buffers are host tensors, so DeviceReplayBuffer's shared
    refill (play_device -> rows_pack -> all-gather -> replay_put_records) runs over gloo exactly as it
    does over RCCL.  Self-play is the random mover through the host game loop."""

    def __init__(self, n, seed):
        from azalea_amd import AzaleaAgent, HexGame, _lib
        self.n, self.cells = n, n * n
        self.record_bytes = _lib.record_bytes(self.cells)
        self.torch_device = torch.device("cpu")
        self.agent = AzaleaAgent(lambda: HexGame(n))
        self.agent.seed(seed)
        self.queue = np.zeros((0, self.record_bytes), np.uint8)
        self.ring, self.size, self.write = None, 0, 0
        self.uid = seed << 20

    def replay_create(self, capacity):
        self.ring, self.size, self.write = np.zeros((capacity, self.record_bytes), np.uint8), 0, 0

    def replay_state(self):
        return dict(capacity=len(self.ring), size=self.size, write_idx=self.write)

    def play_device(self, min_positions, max_plies=0):
        from azalea_amd import distributed as azd
        from azalea_amd.play_game import play_game
        recs, games = [], 0
        while sum(len(r) for r in recs) < min_positions:
            _, frame, gm = play_game([self.agent], collect_data=True)
            P = len(frame)
            prob = np.zeros((P, self.cells), np.float32)
            for i, p in enumerate(frame.moves_prob):
                prob[i, :len(p)] = p
            rows = dict(board=np.stack([s.board for s in frame.state]), color=np.array([s.color for s in frame.state]),
                        nlegal=np.array([len(s.legal_moves) for s in frame.state]), moves_prob=prob,
                        reward=np.array(frame.reward, np.float32), game_uid=np.full(P, self.uid, np.int64))
            self.uid += 1
            games += 1
            recs.append(azd.pack_rows(rows, self.cells))
        self.queue = np.concatenate(recs)
        return len(self.queue), dict(games=games, plies=len(self.queue), game_errors=0, seconds=0.0,
                                     sum_reward_last=float(games), sum_search_value=0.0, sum_root_width=0.0,
                                     sum_action_logprob=0.0)

    @staticmethod
    def _host(ptr, nbytes):
        import ctypes
        return np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(ptr))

    def rows_pack(self, first, n, ptr):
        self._host(ptr, n * self.record_bytes)[:] = self.queue[first:first + n].reshape(-1)

    def replay_put_records(self, n, ptr):
        rec = self._host(ptr, n * self.record_bytes).reshape(n, self.record_bytes)
        for r in rec:                                   # replay_buffer.py:134-149: FIFO, oldest overwritten
            self.ring[self.write] = r
            self.write = (self.write + 1) % len(self.ring)
            self.size = min(len(self.ring), self.size + 1)


def _shared_device_replay(rank, world):
    """DeviceReplayBuffer.consume under torch.distributed: every rank plays its share, every rank's ring
    ends up with ALL ranks' rows, in rank order, and the fresh-example accounting sees all of them."""
    from azalea_amd import distributed as azd
    from azalea_amd.device_replay import DeviceReplayBuffer
    E = StubEngine(4, seed=50 + rank)
    buf = DeviceReplayBuffer(E, capacity=500)
    m = buf.consume(40)
    x = buf.last_exchange
    counts = x["rows_per_rank"]
    fails = [] if (len(counts) == world and all(c >= 20 for c in counts) and len(buf) == sum(counts)) else [100]
    fails += [] if (buf.fresh_counter == sum(counts) - 40 and m["games"] >= 2 and m["moves_per_game"] == sum(counts)) else [101]
    # every rank holds the same ring: rank 0's rows first, then rank 1's
    digest = torch.tensor([int(E.ring[:E.size].astype(np.int64).sum()), E.size], dtype=torch.int64)
    both = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(both, digest)
    fails += [] if (torch.equal(both[0], both[1])) else [102]
    mine = azd.unpack_rows(E.queue, 4)
    lo = sum(counts[:rank])
    held = azd.unpack_rows(E.ring[lo:lo + counts[rank]], 4)
    fails += [] if (all(np.array_equal(mine[k], held[k]) for k in mine)) else [103]
    # a refill smaller than the world: rank 1 plays nothing, still takes rank 0's rows
    before = len(buf)
    buf.fresh_counter = 0
    buf.consume(0.5)                                    # refill = 0.5 - (-0.5) = 1 row: rank 0's quota is 1, rank 1's 0
    c2 = buf.last_exchange["rows_per_rank"]
    fails += [] if (c2[1] == 0 and c2[0] >= 1 and len(buf) == before + c2[0]) else [104]
    return fails


class StubPlayEngine:
    """Stands in for the Engine a device-policy Player creates (Player._get_engine): play() returns whole random
    games keyed by the seed base and the global game index the Player configured it with."""
    ROW_METRIC_COLUMNS = (("search_value", 0), ("search_root_width", 1), ("action_logprob", 2),
                          ("search_root_visits", 4), ("search_tree_nodes", 5), ("search_root_children", 6))
    created = []

    def __init__(self, board_size, n_games, seed, game_index_stride, game_index_offset, **kw):
        self.n, self.seed, self.stride, self.offset, self.started = board_size, seed, game_index_stride, game_index_offset, 0
        self._rows = 0
        StubPlayEngine.created.append(self)

    def set_weights(self, *a, **k):
        pass

    def close(self):
        pass

    def play(self, want):
        parts = []
        while sum(len(p["reward"]) for p in parts) < want:
            uid = self.started * self.stride + self.offset
            self.started += 1
            rows = _rows(0, 3 + uid % 4, self.n)
            rows["game_uid"][:] = uid
            rows["nlegal"][:] = 1
            rows["board"][:] = 1
            rows["board"][:, 0, 0] = 0
            parts.append(rows)
        out = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
        self._rows = len(out["reward"])
        return out, dict(games=len(parts), game_errors=0, seconds=0.01)

    def play_row_metrics(self):
        return np.ones((self._rows, 8), np.float32)


def _device_policy_player(rank, world):
    """A Player whose policy holds a HexNetwork takes the engine path.  The shared seed base is agreed at the
    top of read(), where every rank arrives -- also the rank whose quota is 0 (read(1), size < world) and a
    rank that still has games queued -- and a Player that does not gather never enters a collective."""
    from azalea_amd import AzaleaAgent, HexGame, Player, Policy
    from azalea_amd import parallel_player as pp
    real = pp._eng.Engine
    pp._eng.Engine = StubPlayEngine
    fails = []
    try:
        def make(seed):
            cfg = dict(device="cpu", network="HexNetwork", board_size=4, num_blocks=1, base_chans=8, simulations=20,
                       search_batch_size=10, exploration_coef=0.5, exploration_depth=3, exploration_noise_alpha=0.3,
                       exploration_noise_scale=0.25, exploration_temperature=1.0, seed=seed)
            pol = Policy()
            pol.initialize(cfg)
            pol.settings.update(move_sampling=True, move_exploration=True)
            return AzaleaAgent(lambda: HexGame(4), policy=pol, device="cpu")
        StubPlayEngine.created.clear()
        pl = Player(None, [make(200 + rank)], n_games=8)      # ranks seeded differently
        frame1, m1 = pl.read(1)                                # rank 1's quota is 0: it must not hang in a broadcast
        bases = [None] * world
        dist.all_gather_object(bases, pl._seed_base)
        fails += [] if (bases[0] == bases[1] and bases[0] is not None) else [201]
        fails += [] if (len(frame1) >= 1 and m1["games"] >= 1) else [202]
        frame2, m2 = pl.read(30)                               # both ranks produce now; rank 0 may still hold queued games
        sizes = [None] * world
        dist.all_gather_object(sizes, len(frame2))
        fails += [] if (sizes[0] == sizes[1] and len(frame2) >= 30) else [203]
        engines = [e for e in StubPlayEngine.created]
        if engines:                                            # rank r of W plays the games r, r + W, ...
            fails += [] if ((engines[0].stride, engines[0].offset, engines[0].seed) == (world, rank, bases[0])) else [204]
        fails += [] if (set(m2) >= {"search_root_visits", "search_root_children", "search_tree_nodes", "game_error"}) else [205]
        pl.stop()
        # gather=False: reads are independent, no collective anywhere (only rank 0 reads here)
        if rank == 0:
            solo = Player(None, [make(300)], n_games=8, gather=False)
            f3, _ = solo.read(5)
            fails += [] if len(f3) >= 5 else [206]
            solo.stop()
        dist.barrier()
    finally:
        pp._eng.Engine = real
    return fails


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from azalea_amd import distributed as azd
    from azalea_amd.network import HexNetwork
    fails = []
    counts = [7, 12]
    got = azd.all_gather_rows(_rows(rank, counts[rank]), 5)
    want = {k: np.concatenate([_rows(r, counts[r])[k] for r in range(world)]) for k in got}
    fails += [] if (all(np.array_equal(got[k], want[k]) for k in got)) else [1]
    m = azd.all_reduce_metrics({"games": 1 + rank, "reward": 0.5})
    fails += [] if (m == {"games": 3.0, "reward": 1.0}) else [2]
    # the actor / learner pull: records gathered to rank 0 only (ragged counts, an empty contribution)
    cells = 25
    mine = torch.from_numpy(azd.pack_rows(_rows(rank, counts[rank]), cells))
    for dst, give in ((0, [mine, mine]), (1, [mine, mine]), (0, [mine[:0], mine])):
        parts, cnt = azd.gather_records(give[rank], dst=dst)
        want_cnt = [0 if (dst == 0 and give[0].shape[0] == 0 and r == 0) else counts[r] for r in range(world)]
        ok = cnt == want_cnt
        if rank == dst:
            for r in range(world):
                ok = ok and np.array_equal(parts[r].numpy(), azd.pack_rows(_rows(r, counts[r]), cells)[:want_cnt[r]])
        else:
            ok = ok and parts == []
        fails += [] if ok else [20 + dst]
    fails += [] if ([azd.shard_quota(25, r, 2) for r in range(2)] == [13, 12]) else [3]
    torch.manual_seed(rank)
    net = HexNetwork(board_size=5, num_blocks=1, base_chans=8)
    azd.broadcast_weights(net, src=0)
    torch.manual_seed(0)
    ref = HexNetwork(board_size=5, num_blocks=1, base_chans=8)
    fails += [] if (all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), ref.state_dict().values()))) else [4]
    # Player.read sharded over ranks (random mover -> host loop): every rank ends with all rows
    from azalea_amd import AzaleaAgent, HexGame, Player
    agent = AzaleaAgent(lambda: HexGame(4))
    agent.seed(10 + rank)
    pl = Player(None, [agent])
    frame, metrics = pl.read(40)
    pl.stop()
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([len(frame)]))
    fails += [] if (len(frame) >= 40 and sizes[0].item() == sizes[1].item()) else [5]
    fails += [] if (metrics["moves_per_game"] == len(frame)) else [6]
    # size < world: the trailing rank's quota is 0 -- it still joins both collectives with a 0-row table
    pl = Player(None, [agent])
    frame1, m1 = pl.read(1)
    pl.stop()
    dist.all_gather(sizes, torch.tensor([len(frame1)]))
    fails += [] if (len(frame1) >= 1 and sizes[0].item() == sizes[1].item() and m1["games"] == 1) else [7]
    fails += [] if ([azd.shard_quota(12.8, r, 16) for r in (0, 12, 13, 15)] == [1, 1, 0, 0]) else [8]
    fails += [] if (azd.broadcast_int(1000 + rank) == 1000) else [9]
    fails += _shared_device_replay(rank, world)
    fails += _device_policy_player(rank, world)
    out[rank] = fails
    dist.destroy_process_group()


def test_world2_gloo_all_gather_and_broadcast():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: [], 1: []}      # numbers of the checks that failed, per rank


# ---- training-time topology: rank 0 trains, every rank plays with rank 0's live weights ---------------------------

class DigestEngine(StubPlayEngine):
    """StubPlayEngine that remembers a digest of every weight set it was handed (Player._push_weights)."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.digests = []

    def set_weights(self, tensors, on_device=False):
        import hashlib
        h = hashlib.sha256()
        for name in sorted(tensors):
            h.update(name.encode())
            h.update(np.ascontiguousarray(tensors[name]).tobytes())
        self.digests.append(h.hexdigest())


def _train_config(rundir_seed=0):
    return dict(seed=7, device="cpu", replaybuf_oversampling=4, batch_size=8, game="azalea_amd.game.hex.HexGame",
                board_size=4, replaybuf_size=64, lr_initial=0.05, momentum=0.9, l2_regularization=1e-4,
                lr_decay_epochs=100, lr_decay=0.1, total_epochs=3, selfplay_games=8, log_interval=0,
                model_checkpoint_interval=10, selfplay_mode="lockstep")


def _train_worker(rank, world, port, rundir, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import glob
    from azalea_amd import Policy
    from azalea_amd import parallel_player as pp
    from azalea_amd.policy_trainer import train
    real, real_announce = pp._eng.Engine, pp.Player.announce
    pp._eng.Engine = DigestEngine
    synced = []

    def announce(self, op, arg):                       # digest of the module right after every weight broadcast
        real_announce(self, op, arg)
        pol_ = self._device_policy()
        if self.role is not None and pol_ is not None:
            e = DigestEngine.__new__(DigestEngine)
            e.digests = []
            e.set_weights({k: v.detach().cpu().numpy() for k, v in pol_.net.state_dict().items() if v.dtype == torch.float32})
            synced.append(e.digests[0])
    pp.Player.announce = announce
    fails = []
    try:
        cfg = dict(device="cpu", network="HexNetwork", board_size=4, num_blocks=1, base_chans=8, simulations=20,
                   search_batch_size=10, exploration_coef=0.5, exploration_depth=3, exploration_noise_alpha=0.3,
                   exploration_noise_scale=0.25, exploration_temperature=1.0, seed=5)
        torch.manual_seed(100 + rank)                  # the ranks START with different networks
        pol = Policy()
        pol.initialize(cfg)
        StubPlayEngine.created.clear()
        path = train(pol, _train_config(), rundir)
        engines = [e for e in StubPlayEngine.created if isinstance(e, DigestEngine)]
        packed = engines[0].digests if engines else []
        alld = [None] * world
        dist.all_gather_object(alld, synced)
        # every announced production found rank 0's weights of that moment on every rank (they started apart), the
        # weights moved between productions (training happened), and an engine only ever packed synced weights
        fails += [] if (len(synced) >= 3 and all(d == alld[0] for d in alld)) else [301]
        fails += [] if len(set(alld[0])) >= 3 else [302]
        fails += [] if (len(packed) >= 1 and set(packed) <= set(synced)) else [306]
        # everyone leaves train() with the trained network; rank 0 alone wrote checkpoints
        sd = torch.cat([t.detach().reshape(-1).double() for t in pol.net.state_dict().values() if t.is_floating_point()])
        sums = [None] * world
        dist.all_gather_object(sums, float(sd.sum()))
        fails += [] if all(x == sums[0] for x in sums) else [303]
        dist.barrier()
        files = sorted(glob.glob(os.path.join(rundir, "checkpoints", "*.policy.pth")))
        fails += [] if (os.path.basename(path) == "final.policy.pth" and os.path.exists(path)
                        and len(files) == len(set(files)) and any("checkpoint.0." in f for f in files)) else [304]
        if rank == 0:
            state = torch.load(path, weights_only=False)["policy"]["net"]
            same = all(torch.equal(state[k], v) for k, v in pol.net.state_dict().items())
            fails += [] if same else [305]
    finally:
        pp._eng.Engine, pp.Player.announce = real, real_announce
    out[rank] = fails
    azd_reset()
    dist.destroy_process_group()


def azd_reset():
    from azalea_amd import distributed as azd
    azd.reset_control_group()


def _spawn(fn, world, *args):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(fn, args=(world, _free_port()) + args + (out,), nprocs=world, join=True)
    return dict(out)


def test_train_rank0_trains_everyone_plays_with_its_weights(tmp_path):
    """policy_trainer.train under torch.distributed (world 2 and 3): rank 0 runs the optimizer, announces every
    shared Player.read and broadcasts its network first; the other ranks serve self-play.  Every production saw
    the same weights on all ranks although they started from different networks, the weights changed between
    refills (training happened), all ranks return with the trained network, and the checkpoint files were
    written once."""
    for world in (2, 3):
        rundir = str(tmp_path / ("run%d" % world))
        out = _spawn(_train_worker, world, rundir)
        assert out == {r: [] for r in range(world)}


def _world8_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from azalea_amd import distributed as azd
    fails = []
    # ragged all-gather with empty ranks: rank r contributes r % 3 rows
    counts = [r % 3 for r in range(world)]
    got = azd.all_gather_rows(_rows(rank, counts[rank]), 5)
    want = {k: np.concatenate([_rows(r, counts[r])[k] for r in range(world)]) for k in got}
    fails += [] if all(np.array_equal(got[k], want[k]) for k in got) else [401]
    # quotas: 8 ranks, 5 rows wanted -> ranks 0..4 play one row's worth, 5..7 nothing, and they still join
    fails += [] if [azd.shard_quota(5, r, world) for r in range(world)] == [1, 1, 1, 1, 1, 0, 0, 0] else [402]
    from azalea_amd import AzaleaAgent, HexGame, Player
    agent = AzaleaAgent(lambda: HexGame(4))
    agent.seed(10 + rank)
    pl = Player(None, [agent])
    frame, m = pl.read(5)
    pl.stop()
    sizes = [None] * world
    dist.all_gather_object(sizes, len(frame))
    fails += [] if (len(set(sizes)) == 1 and m["games"] == 5 and len(frame) >= 5) else [403]
    # the shared device-ring refill with 8 ranks, three of them idle
    from azalea_amd.device_replay import DeviceReplayBuffer
    E = StubEngine(4, seed=50 + rank)
    buf = DeviceReplayBuffer(E, capacity=500)
    buf.consume(2.5)                                    # refill = 2.5 - (-2.5) = 5 rows
    c = buf.last_exchange["rows_per_rank"]
    fails += [] if (len(c) == world and c[5:] == [0, 0, 0] and min(c[:5]) >= 1 and len(buf) == sum(c)) else [404]
    dig = [None] * world
    dist.all_gather_object(dig, int(E.ring[:E.size].astype(np.int64).sum()))
    fails += [] if len(set(dig)) == 1 else [405]
    # leader / follower announcements reach all 8
    if rank == 0:
        azd.lead(azd.OP_REFILL, 1234)
        got = (azd.OP_REFILL, 1234)
    else:
        got = azd.follow()
    fails += [] if got == (azd.OP_REFILL, 1234) else [406]
    out[rank] = fails
    dist.destroy_process_group()


def test_world8_gloo_quota_zero_ranks_join_the_collectives():
    out = _spawn(_world8_worker, 8)
    assert out == {r: [] for r in range(8)}


# ---- actor / learner: the other ranks play AHEAD while rank 0 trains ---------------------------------------------

class ActorEngine(DigestEngine):
    """DigestEngine with the device-side production surface the actors use (play_device -> rows_pack), on host
    memory: every call plays a couple of whole random games, slowly enough that training steps run beside it."""

    def __init__(self, board_size, n_games, seed, game_index_stride, game_index_offset, **kw):
        super().__init__(board_size, n_games, seed, game_index_stride, game_index_offset, **kw)
        from azalea_amd import _lib
        self.cells = board_size * board_size
        self.record_bytes = _lib.record_bytes(self.cells)
        self.torch_device = torch.device("cpu")
        self.queue = np.zeros((0, self.record_bytes), np.uint8)
        self.produced_with = []          # (digest of the weights in force, rows) per production

    def play_device(self, min_positions, max_plies=0):
        import time
        from azalea_amd import distributed as azd
        time.sleep(0.002)
        rows, _ = self.play(6)
        self.queue = azd.pack_rows(rows, self.cells)
        self.produced_with.append((self.digests[-1] if self.digests else None, len(self.queue)))
        games = len(np.unique(rows["game_uid"]))
        return len(self.queue), dict(games=games, plies=len(self.queue), game_errors=0, seconds=0.002,
                                     sum_reward_last=float(games))

    def rows_pack(self, first, n, ptr):
        StubEngine._host(ptr, n * self.record_bytes)[:] = self.queue[first:first + n].reshape(-1)


def _net_digest(net):
    e = DigestEngine.__new__(DigestEngine)
    e.digests = []
    e.set_weights({k: v.detach().cpu().numpy() for k, v in net.state_dict().items() if v.dtype == torch.float32})
    return e.digests[0]


def _policy(seed):
    from azalea_amd import Policy
    cfg = dict(device="cpu", network="HexNetwork", board_size=4, num_blocks=1, base_chans=8, simulations=20,
               search_batch_size=10, exploration_coef=0.5, exploration_depth=3, exploration_noise_alpha=0.3,
               exploration_noise_scale=0.25, exploration_temperature=1.0, seed=5)
    torch.manual_seed(seed)
    pol = Policy()
    pol.initialize(cfg)
    return pol


def _actor_learner_worker(rank, world, port, rundir, fail_at, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["AZX_FOLLOW_TIMEOUT"] = "60"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import glob
    from azalea_amd import actor_learner as al
    from azalea_amd import distributed as azd
    from azalea_amd import parallel_player as pp
    from azalea_amd import policy_trainer as pt
    real, real_sync = pp._eng.Engine, al.Learner.sync_weights
    pp._eng.Engine = ActorEngine
    sent = []

    def sync(self):
        if fail_at is not None and self.weight_syncs == fail_at:
            raise RuntimeError("injected trainer failure")
        real_sync(self)
        sent.append(_net_digest(self.net))
    al.Learner.sync_weights = sync
    fails = []
    try:
        pol = _policy(100 + rank)                      # the ranks START with different networks
        StubPlayEngine.created.clear()
        cfg = dict(_train_config(), total_epochs=4, weight_sync_steps=3, selfplay_ahead_rows=40)
        del cfg["selfplay_mode"]                       # the default under torch.distributed is actor_learner
        hist = {}
        raised = None
        try:
            path = pt.train(pol, cfg, rundir, history=hist)
        except Exception as exc:                       # noqa: BLE001
            raised = exc
        if fail_at is not None:
            # rank 0 failed between two announcements: it re-raises its error, every actor is told and leaves
            want = RuntimeError if rank == 0 else azd.LeaderLost
            fails += [] if isinstance(raised, want) else [520]
            out[rank] = fails
            return
        fails += [] if raised is None else [500]
        if os.environ.get("AZX_TEST_VERBOSE"):
            print(rank, hist, flush=True)
        engines = [e for e in StubPlayEngine.created if isinstance(e, ActorEngine)]
        if rank == 0:
            L = hist["learner"]
            # rank 0 never played; it pulled rows and sent its network every 3 steps (+ once before the first step)
            fails += [] if (hist["selfplay_mode"] == "actor_learner" and not any(e.produced_with for e in engines)) else [501]
            fails += [] if (L["pulls"] >= 2 and L["weight_syncs"] == 1 + L["steps"] // 3 and L["steps"] >= 24) else [502]
            mine = sent
        else:
            A = hist["actor"]
            eng = engines[0]
            mine = eng.digests
            # the actor produced BETWEEN announcements (self-play ran beside training), never before the first
            # broadcast, never more than its bound ahead, and answered every pull
            fails += [] if (A["max_productions_between_announcements"] >= 1 and A["pulls"] >= 2) else [503]
            fails += [] if all(d is not None for d, _ in eng.produced_with) else [504]
            fails += [] if A["weight_syncs"] == len(eng.digests) >= 2 else [505]
        alld = [None] * world
        dist.all_gather_object(alld, mine)
        # every broadcast left every rank with the same network; it changed from broadcast to broadcast
        fails += [] if (all(d == alld[0] for d in alld) and len(set(alld[0])) >= 3) else [506]
        sd = torch.cat([t.detach().reshape(-1).double() for t in pol.net.state_dict().values() if t.is_floating_point()])
        sums = [None] * world
        dist.all_gather_object(sums, float(sd.sum()))
        fails += [] if all(x == sums[0] for x in sums) else [507]          # everyone leaves with the trained network
        dist.barrier()
        files = sorted(glob.glob(os.path.join(rundir, "checkpoints", "*.policy.pth")))
        fails += [] if (os.path.exists(path) and any("checkpoint.0." in f for f in files)) else [508]
        if rank == 0:
            state = torch.load(path, weights_only=False)["policy"]["net"]
            fails += [] if all(torch.equal(state[k], v) for k, v in pol.net.state_dict().items()) else [509]
    finally:
        pp._eng.Engine, al.Learner.sync_weights = real, real_sync
        out[rank] = fails
        azd_reset()
        dist.destroy_process_group()


def test_actor_learner_followers_play_ahead_while_rank0_trains(tmp_path):
    """policy_trainer.train's default topology under torch.distributed (world 2 and 3): rank 0 trains and never plays;
    the other ranks produce between its announcements into bounded backlogs, hand rows over when it pulls, and pack
    the network it broadcasts every `weight_sync_steps` steps -- identical digests on every rank at every broadcast,
    one set of checkpoint files, everyone returns with the trained network."""
    for world in (2, 3):
        out = _spawn(_actor_learner_worker, world, str(tmp_path / ("al%d" % world)), None)
        assert out == {r: [] for r in range(world)}


def test_actor_learner_trainer_failure_releases_the_actors(tmp_path):
    """a rank 0 that raises (here: at its third weight broadcast) must not leave the others waiting in a
    broadcast -- it announces OP_ABORT, the actors raise LeaderLost and exit, rank 0 re-raises its own error."""
    out = _spawn(_actor_learner_worker, 2, str(tmp_path / "abort"), 2)
    assert out == {0: [], 1: []}


def _lockstep_abort_worker(rank, world, port, rundir, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["AZX_FOLLOW_TIMEOUT"] = "60"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from azalea_amd import distributed as azd
    from azalea_amd import parallel_player as pp
    from azalea_amd import policy_trainer as pt
    real, real_make = pp._eng.Engine, pt.make_train_step
    pp._eng.Engine = DigestEngine

    def make(*a, **k):
        raise ValueError("train_step_native was asked for, but: 13x13 is outside the envelope")
    pt.make_train_step = make
    raised = None
    try:
        pt.train(_policy(100 + rank), _train_config(), rundir)
    except Exception as exc:                           # noqa: BLE001
        raised = exc
    finally:
        pp._eng.Engine, pt.make_train_step = real, real_make
    out[rank] = [] if isinstance(raised, ValueError if rank == 0 else azd.LeaderLost) else [600 + rank]
    azd_reset()
    dist.destroy_process_group()


def test_lockstep_trainer_failure_before_the_first_step_releases_the_followers(tmp_path):
    """The case ADVICE r4 names: the training step cannot be built for this network (a ValueError on rank 0 before
    its loop starts); the followers, already waiting for an announcement, are told and leave."""
    out = _spawn(_lockstep_abort_worker, 2, str(tmp_path / "ls_abort"))
    assert out == {0: [], 1: []}


def _timeout_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import time
    from azalea_amd import distributed as azd
    azd.control_group()
    fails = []
    if rank == 0:
        time.sleep(1.5)                                # silent for longer than the follower's patience
        azd.lead(azd.OP_STOP)                          # (matches the follower's still-posted receive)
    else:
        t0 = time.monotonic()
        try:
            azd.follow(timeout=0.5)
            fails.append(700)
        except azd.LeaderLost:
            fails += [] if 0.4 < time.monotonic() - t0 < 1.4 else [701]
    out[rank] = fails
    dist.barrier()
    azd_reset()
    dist.destroy_process_group()


def test_follow_times_out_when_rank0_goes_silent():
    out = _spawn(_timeout_worker, 2)
    assert out == {0: [], 1: []}


def test_learner_pull_fills_the_device_ring_from_the_actors_backlogs():
    out = _spawn(_pull_ring_worker, 3)
    assert out == {r: [] for r in range(3)}


def _pull_ring_worker(rank, world, port, out):
    """DeviceReplayBuffer.consume in actor / learner mode, without train(): rank 0's ring receives exactly the rows
    the two actors hand over, in rank order, whole chunks only; a pull larger than a backlog makes the actor play."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from azalea_amd import actor_learner as al
    from azalea_amd import distributed as azd
    from azalea_amd.device_replay import DeviceReplayBuffer
    from azalea_amd.network import HexNetwork
    azd.control_group()
    fails = []
    torch.manual_seed(rank)
    net = HexNetwork(board_size=4, num_blocks=1, base_chans=8)
    per_pull = []                                      # this rank's handed-over records, one array per pull
    if rank == 0:
        E = StubEngine(4, seed=50)
        buf = DeviceReplayBuffer(E, capacity=400)
        buf.learner = L = al.Learner(net, weight_sync_steps=1000)
        L.sync_weights()
        m = buf.consume(15)                            # refill 30 -> 15 rows per actor
        c = L.last_pull["rows_per_rank"]
        fails += [] if (c[0] == 0 and min(c[1:]) >= 15 and len(buf) == sum(c) and buf.fresh_counter == sum(c) - 15) else [801]
        fails += [] if (m["games"] >= 2 and m["moves_per_game"] == sum(c)) else [802]
        buf.fresh_counter = 0
        buf.consume(100)                               # refill 200: more than the backlogs hold -> the actors play on
        c2 = L.last_pull["rows_per_rank"]
        fails += [] if (c2[0] == 0 and min(c2[1:]) >= 100) else [803]
        L.stop()
    else:
        class P:                                       # the slice of Player the actor loop uses
            n_games, weight_syncs = 8, 0
            eng = ActorEngine(4, 8, 1234, world, rank)
            pol = type("Pol", (), {"net": net})()

            def _device_policy(self):
                return self.pol

            def device_engine(self):
                return self.eng

            def prepare_device_engine(self, eng):
                eng.set_weights({k: v.detach().numpy() for k, v in net.state_dict().items() if v.dtype == torch.float32})
        real_take = al.RecordBacklog.take

        def take(self, quota):
            recs, tot = real_take(self, quota)
            per_pull.append(np.concatenate([r.numpy() for r in recs]))
            return recs, tot
        al.RecordBacklog.take = take
        try:
            stats = al.serve_selfplay_ahead(P(), ahead_rows=40)
        finally:
            al.RecordBacklog.take = real_take
        fails += [] if (stats["pulls"] == 2 and stats["weight_syncs"] == 1 and stats["rows"] >= 115) else [810]
        fails += [] if stats["max_productions_between_announcements"] >= 1 else [811]
    got = [None] * world
    dist.all_gather_object(got, per_pull)
    if rank == 0:                                      # ring order: pull 1 (actor 1, actor 2), pull 2 (actor 1, actor 2)
        want = np.concatenate([got[r][p] for p in range(2) for r in range(1, world)])
        fails += [] if np.array_equal(E.ring[:E.size], want) else [804]
    out[rank] = fails
    azd_reset()
    dist.destroy_process_group()


# ---- a production that fails AFTER its announcement: carried through the counts collective ---------------

class FailingActorEngine(ActorEngine):
    """ActorEngine whose `fail_rank`'s `fail_call`-th production raises (AZX_ERANGE / a full arena / a HIP error)."""
    fail_rank, fail_call = None, 4

    def play_device(self, min_positions, max_plies=0):
        self.calls = getattr(self, "calls", 0) + 1
        if dist.get_rank() == self.fail_rank and self.calls == self.fail_call:
            raise RuntimeError("injected self-play failure")
        return super().play_device(min_positions, max_plies)


def _production_failure_worker(rank, world, port, rundir, mode, fail_rank, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["AZX_FOLLOW_TIMEOUT"] = "60"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import time
    from azalea_amd import distributed as azd
    from azalea_amd import parallel_player as pp
    from azalea_amd import policy_trainer as pt
    real = pp._eng.Engine
    FailingActorEngine.fail_rank = fail_rank
    raised = None
    t0 = time.monotonic()
    try:
        if mode == "actor_learner":
            pp._eng.Engine = FailingActorEngine
            cfg = dict(_train_config(), total_epochs=40, weight_sync_steps=3, selfplay_ahead_rows=40)
            del cfg["selfplay_mode"]
            pt.train(_policy(100 + rank), cfg, rundir)
        else:
            # lock-step device refill, without train(): rank 0 announces, every rank plays its share -- one of them fails
            from azalea_amd.device_replay import DeviceReplayBuffer
            azd.control_group()
            E = StubEngine(4, seed=50 + rank)
            real_play = E.play_device
            calls = [0]

            def play(min_positions, max_plies=0):
                calls[0] += 1
                if rank == fail_rank and calls[0] == 2:
                    raise RuntimeError("injected self-play failure")
                return real_play(min_positions, max_plies)
            E.play_device = play
            buf = DeviceReplayBuffer(E, capacity=500)
            for _ in range(3):
                if rank == 0:
                    azd.lead(azd.OP_REFILL, 40)
                else:
                    op, arg = azd.follow(30)
                    assert (op, arg) == (azd.OP_REFILL, 40)
                buf.consume(40)
                buf.fresh_counter = 0
    except Exception as exc:                           # noqa: BLE001
        raised = exc
    finally:
        pp._eng.Engine = real
    took = time.monotonic() - t0
    # the failed rank raises ITS error; every other rank leaves the counts collective with PeerFailed, at once
    want = RuntimeError if rank == fail_rank else azd.PeerFailed
    ok = isinstance(raised, want) and (rank != fail_rank or "injected" in str(raised)) and took < 45
    out[rank] = [] if ok else [800 + rank, repr(raised), took]
    azd_reset()
    dist.destroy_process_group()


def test_a_failed_production_is_carried_through_the_counts_collective(tmp_path):
    """rank 0 announces a refill and then ITS share fails (AZX_ERANGE, SearchTreeFull ...) -- the
    followers are already on their way into the record all-gather and would wait there for the RCCL watchdog.  The
    failed rank joins the counts collective with -1 and every rank raises distributed.PeerFailed right after it; the
    same when a follower's share fails, and when an actor's production fails between two pulls of an actor / learner
    run (the learner's next pull raises PeerFailed, the healthy actor too, the failed actor its own error)."""
    for mode, world, fail_rank in (("lockstep", 2, 0), ("lockstep", 3, 2), ("actor_learner", 3, 2)):
        out = _spawn(_production_failure_worker, world, str(tmp_path / ("%s_%d" % (mode, fail_rank))), mode, fail_rank)
        assert out == {r: [] for r in range(world)}, (mode, fail_rank, out)
