"""Player: endless stream of self-play games, read in whole games (azalea/parallel_player.py:17-76).

The reference runs one game per worker process and ships pickled frames back through pipes.
Here a Policy with the device network plays `n_games` games concurrently inside one engine
(throughput mode: device RNG, device move draw); finished games are harvested in whole, buffered
on the host and handed out by `read(size)` exactly as batch_examples does -- whole games until
`len >= size`, metrics summed over the games returned.  With torch.distributed initialised each
rank plays its share and the rows are all-gathered (azalea_amd/distributed.py).

The `pool` argument is accepted for signature compatibility (policy_trainer.py:75) and unused:
there are no worker processes.  Random movers and duck-typed networks go through the host
play_game loop (one game at a time), like the reference's in-process pool (num_workers=0).
"""
import logging
import os
from collections import defaultdict, deque
from typing import Dict, Sequence, Tuple

import numpy as np
import torch

from . import distributed as azdist
from . import engine as _eng
from .game.hex import HexGameState
from .play_game import play_game
from .policy import Policy, SearchTreeFull
from .replay_buffer import ReplayDataFrame

Metrics = Dict[str, float]


def rows_to_frame(rows) -> ReplayDataFrame:
    """Engine rows -> the reference's struct-of-lists frame (replay_buffer.py:11-38).  The per-row
    arrays are cut out of whole-table numpy passes (one nonzero for every row's legal moves, one
    astype per table); only the GameState constructors run per row."""
    frame = ReplayDataFrame()
    P = len(rows["reward"])
    if P == 0:
        return frame
    n = rows["board"].shape[-1]
    boards = np.ascontiguousarray(rows["board"], np.int32).reshape(P, n, n)
    flat = boards.reshape(P, n * n)
    ri, ci = np.nonzero(flat == 0)                                   # row-major: ascending tiles per row
    counts = np.bincount(ri, minlength=P)
    k = np.asarray(rows["nlegal"], np.int64)
    if not np.array_equal(counts, k):
        raise AssertionError("replay rows: legal-move counts do not match the boards")
    ends = np.cumsum(counts)
    legal_all = (ci + 1).astype(np.int32)
    probs = np.ascontiguousarray(rows["moves_prob"], np.float32)
    color = np.asarray(rows["color"]).tolist()
    reward = np.asarray(rows["reward"], np.float32)
    state, mp = frame.state, frame.moves_prob
    s = 0
    for i in range(P):
        e = int(ends[i])
        state.append(HexGameState(color[i], legal_all[s:e], 0, boards[i]))
        mp.append(probs[i, :e - s])
        s = e
    frame.reward.extend(reward)       # np.float32 scalars, like play_game.py:64-65
    return frame


class Player:
    MAX_BARREN_PRODUCTIONS = 1000     # consecutive productions without a finished game before read() gives up

    def __init__(self, pool, agents: Sequence, *, n_games: int = None, gather: bool = True, role: str = None):
        """`gather`: under torch.distributed every rank plays its share of a read and all ranks get all rows.
        `role`: None -- every rank calls read() itself, in lock-step (symmetric); "leader" / "follower" -- the
        training-time topology (azalea_amd/distributed.py: rank 0 announces each shared production and broadcasts
        the trainer's weights first; the followers are driven by policy_trainer.serve_selfplay)."""
        if role not in (None, "leader", "follower"):
            raise ValueError("Player role must be None, 'leader' or 'follower'")
        self.agents = agents
        self.running = True
        self.gather = gather
        self.role = role if (gather and azdist.is_distributed()) else None
        self.learner = None            # actor_learner.Learner: read() pulls the actors' backlogs instead of playing
        self.weight_syncs = 0          # broadcasts of the trainer's weights this player took part in
        self.n_games = n_games or int(os.environ.get("AZX_GAMES", "4096"))
        self._games = deque()          # finished games waiting to be read: (rows dict, metrics)
        self._engine = None
        self._engine_key = None
        self._seed_base = None
        self._skipped = 0              # games dropped because of SearchTreeFull since the last read

    # ---- reference surface -------------------------------------------------------------------
    def read(self, size) -> Tuple[ReplayDataFrame, Metrics]:
        """Whole games until at least `size` positions (parallel_player.py:41-52).  A game that
        overflowed its tree (SearchTreeFull) is skipped like the reference's worker does
        (parallel_player.py:73-76) and counted in metrics['game_error']."""
        if self.learner is not None:
            return self._read_pulled(size)
        shared = self.gather and azdist.is_distributed()
        self.announce(azdist.OP_READ, int(np.ceil(size)))
        self._agree_seed_base()       # a collective when shared: every rank passes here, whatever its quota
        quota = azdist.shard_quota(size) if shared else size
        rows_list, metrics = [], defaultdict(float)
        have, barren = 0, 0
        failure = None
        try:
            while have < quota:
                if not self._games:
                    self._produce(quota - have)
                    if not self._games:
                        barren += 1
                        if barren > self.MAX_BARREN_PRODUCTIONS:
                            raise RuntimeError("self-play produced no finished game in %d attempts" % barren)
                        continue
                rows, gm = self._games.popleft()
                rows_list.append(rows)
                have += len(rows["reward"])
                for name, v in gm.items():
                    metrics[name] += v
        except Exception as exc:
            if not shared:
                raise
            # the other ranks are on their way into this read's record collectives: join the first one with the
            # failure mark so that every rank leaves it (distributed.PeerFailed), then raise what happened here
            failure = exc
            rows_list = []
        if self._skipped:
            metrics["game_error"] += self._skipped
            self._skipped = 0
        n = self.agents[0].game.board_size
        if rows_list:
            rows = {k: np.concatenate([r[k] for r in rows_list]) for k in rows_list[0]}
        else:
            rows = azdist.empty_rows(n)       # quota 0 (size < world): still join the collectives
        if shared:
            try:
                rows = azdist.all_gather_rows(rows, n, failed=failure is not None)
            except azdist.PeerFailed:
                if failure is not None:
                    raise failure
                raise
            metrics = azdist.all_reduce_metrics(dict(metrics))
        return rows_to_frame(rows), dict(metrics)

    def _read_pulled(self, size) -> Tuple[ReplayDataFrame, Metrics]:
        """read() of the learner in actor / learner mode (azalea_amd/actor_learner.py): whole games out of the other
        ranks' backlogs, as a host frame; metrics in play_game's keys, summed over the games returned."""
        n = self.agents[0].game.board_size
        dev = azdist._comm_device()
        parts, counts, st = self.learner.pull(size, dev, azdist.record_bytes(n * n))
        rec = torch.cat(parts).cpu().numpy() if sum(counts) else np.zeros((0, azdist.record_bytes(n * n)), np.uint8)
        rows = azdist.unpack_rows(rec, n)
        metrics = {"games": st.get("games", 0.0), "reward": st.get("sum_reward_last", 0.0),
                   "moves_per_game": float(len(rows["reward"])), "seconds_per_game": st.get("seconds", 0.0),
                   "game_error": st.get("game_errors", 0.0)}
        metrics.update({k[5:]: v for k, v in st.items() if k.startswith("game_") and k != "game_errors"})
        return rows_to_frame(rows), metrics

    def announce(self, op: int, arg: int) -> None:
        """Leader / follower topology only: rank 0 says what all ranks produce next, then its network --
        the trainer's live weights and BatchNorm statistics -- is broadcast, so every rank searches with what
        the trainer holds right now (parallel_player.py:36-38).  Collectives: called by every rank, in the
        same place (the leader from read() / DeviceReplayBuffer.consume, the followers from serve_selfplay
        through the same two methods)."""
        if self.role is None:
            return
        if self.role == "leader":
            azdist.lead(op, arg)
        pol = self._device_policy()
        if pol is not None:
            azdist.broadcast_weights(pol.net, src=0)
            # in-place copies into the parameters do move their version counters, graph replays do not:
            pol.net.weight_updates_outside_autograd = getattr(pol.net, "weight_updates_outside_autograd", 0) + 1
            self.weight_syncs += 1

    def stop(self) -> None:
        self.running = False
        self._games.clear()
        if self._engine is not None:
            self._engine.close()
            self._engine = None

    # ---- device-resident replay (azalea_amd/device_replay.py) ---------------------------------
    def device_engine(self):
        """The engine self-play runs in (created on first use); DeviceReplayBuffer keeps its ring there."""
        pol = self._device_policy()
        if pol is None:
            raise RuntimeError("device-resident replay needs a single agent whose Policy holds a HexNetwork")
        self._agree_seed_base()
        return self._get_engine(pol)

    def prepare_device_engine(self, engine) -> None:
        """Refresh the engine's packed weights from the trainer's live module (SURVEY 8(b) ownership)."""
        pol = self._device_policy()
        if pol is not None and engine is self._engine:
            self._push_weights(engine, pol)

    # ---- production --------------------------------------------------------------------------
    def _device_policy(self):
        pol = getattr(self.agents[0], "policy", None)
        return pol if isinstance(pol, Policy) and pol._uses_device_net() and len(self.agents) == 1 else None

    def _agree_seed_base(self) -> None:
        """Each game draws from its own stream: seed base + GLOBAL game index.  Ranks that share
        their reads use rank 0's base (rank r of W plays the games r, r+W, ...: the set of games does not depend
        on W).  The broadcast is a collective, so it happens where every rank is guaranteed to arrive -- at the
        top of read() / device_engine() -- never inside the production path, which a rank with quota 0 or with
        games still queued skips.  A Player that does not gather derives its base locally."""
        if self._seed_base is not None:
            return
        pol = self._device_policy()
        if pol is None:
            return
        local = int(pol.rng.randint(0, 2 ** 31 - 1))
        if self.gather and azdist.is_distributed():
            self._seed_base = azdist.broadcast_int(local)
        else:
            # no shared index space: the engine counts its own games 0, 1, ... (stride 1, offset 0 in _get_engine);
            # ranks whose policies were seeded alike must still not replay each other's games
            self._seed_base = (local ^ (azdist.rank() * 0x9E3779B1)) & 0x7FFFFFFF

    def _produce(self, want: int) -> None:
        pol = self._device_policy()
        if pol is None:
            self._produce_on_host()
        else:
            self._produce_on_device(pol, want)

    def _produce_on_host(self) -> None:
        """One game through the generic loop (random mover / duck-typed nets / two agents)."""
        for a in self.agents:
            net = getattr(getattr(a, "policy", None), "_net", None)
            if hasattr(net, "eval"):
                net.eval()
        try:
            _, frame, gm = play_game(self.agents, collect_data=True)
        except SearchTreeFull:
            logging.warning("game failed because of SearchTreeFull (skipped)")
            self._skipped += 1
            return
        n = self.agents[0].game.board_size
        P = len(frame)
        rows = dict(board=np.stack([s.board for s in frame.state]).astype(np.int32),
                    color=np.array([s.color for s in frame.state], np.int32),
                    nlegal=np.array([len(s.legal_moves) for s in frame.state], np.int32),
                    moves_prob=np.zeros((P, n * n), np.float32),
                    reward=np.array(frame.reward, np.float32),
                    game_uid=np.full(P, -1, np.int64))
        for i, p in enumerate(frame.moves_prob):
            rows["moves_prob"][i, :len(p)] = p
        self._games.append((rows, dict(gm)))

    def _get_engine(self, pol: Policy):
        n = self.agents[0].game.board_size
        rank, world = ((torch.distributed.get_rank(), torch.distributed.get_world_size())
                       if (self.gather and azdist.is_distributed()) else (0, 1))
        device = (pol.net.device.index or 0) if pol.net.device.type == "cuda" else 0
        key = (n, device, pol.simulations, pol.search_batch_size, float(pol.exploration_coef),
               pol.exploration_depth, pol.exploration_noise_alpha, pol.exploration_noise_scale,
               pol.exploration_temperature, pol.num_blocks, pol.base_chans,
               bool(pol.settings.get("move_sampling")), bool(pol.settings.get("move_exploration")))
        if self._engine is None or key != self._engine_key:
            if self._engine is not None:
                self._engine.close()
            sampling = pol.settings.get("move_sampling", False)
            explore = sampling and pol.settings.get("move_exploration", False)
            if self._seed_base is None:      # read() / device_engine() agree on it before any production
                raise RuntimeError("Player: the seed base must be agreed before the engine is created")
            self._engine = _eng.Engine(
                board_size=n, n_games=self.n_games, simulations=pol.simulations,
                search_batch_size=pol.search_batch_size, exploration_coef=pol.exploration_coef,
                exploration_depth=pol.exploration_depth if sampling else 0,
                noise_alpha=pol.exploration_noise_alpha,
                noise_scale=pol.exploration_noise_scale if explore else 0.0,
                temperature=pol.exploration_temperature if sampling else 0.0,
                evaluator=_eng.EVAL_RESNET, num_blocks=pol.num_blocks, base_chans=pol.base_chans,
                device=device, seed=self._seed_base, game_index_stride=world, game_index_offset=rank)
            self._engine_key = key
        return self._engine

    def _produce_on_device(self, pol: Policy, want: int) -> None:
        eng = self._get_engine(pol)
        self._push_weights(eng, pol)
        rows, st = eng.play(max(1, int(want)))
        uid = rows["game_uid"]
        self._skipped += int(st["game_errors"])
        if len(uid) == 0:
            return
        # play_game's per-game metrics are means over the game's own plies (play_game.py:73-76)
        meta = eng.play_row_metrics()
        starts = np.flatnonzero(np.r_[True, uid[1:] != uid[:-1]])
        ends = np.r_[starts[1:], len(uid)]
        names = [k for k, _ in eng.ROW_METRIC_COLUMNS]
        cols = [c for _, c in eng.ROW_METRIC_COLUMNS]
        for s, e in zip(starts, ends):
            game = {k: v[s:e] for k, v in rows.items()}
            mean = meta[s:e][:, cols].astype(np.float64).mean(0)
            # the key set of play_game.py:69-76 over search_tree.py:109-112 / mcts.py:291 / policy.py:164
            gm = dict(games=1, reward=float(game["reward"][-1]), moves_per_game=int(e - s),
                      seconds_per_game=st["seconds"] / max(1, st["games"]), game_error=0)
            gm.update(zip(names, (float(x) for x in mean)))
            self._games.append((game, gm))

    @staticmethod
    def _push_weights(eng, pol: Policy) -> None:
        net = pol.net
        sd = {k: v for k, v in net.state_dict().items() if v.dtype == torch.float32}
        if net.device.type == "cuda":
            eng.set_weights({k: (v.contiguous().data_ptr(), v.numel()) for k, v in sd.items()}, on_device=True)
        else:
            eng.set_weights({k: v.detach().cpu().numpy() for k, v in sd.items()})
