"""Actor / learner topology of a multi-GPU training run.

The reference keeps `num_workers + 1` self-play games in flight WHILE the trainer trains: `Player.read` only blocks
when the generator has fallen behind (azalea/process_pool.py:31-47, parallel_player.py:17-38), and the workers search
with the trainer's live network (its CUDA-IPC tensors).  One process per GPU gives the same shape:

* rank 0 is the LEARNER: it runs the optimizer and, with more than one rank, never plays.  When its replay buffer's
  fresh-example counter asks for rows (replay_buffer.py:121-132) it PULLS them: one announcement on the control group
  (gloo, host side), then two record collectives (RCCL over xGMI): the per-rank counts, and a GATHER of the records to
  rank 0 -- each actor's block travels once, on its own link into the learner; no actor receives another's rows.  Every `weight_sync_steps` optimizer steps it broadcasts its network (parameters and
  BatchNorm statistics, one flat tensor) -- asynchronously, from a snapshot: the training stream does not wait for the
  actors to join.
* ranks != 0 are ACTORS: they play whole games continuously into a BACKLOG of packed record chunks in HBM
  (`azx_play_device` one pool move at a time, `azx_rows_pack`), polling the control group between moves, and stop
  producing when the backlog holds `ahead_rows` rows (about one pool move's harvest by default) -- the bound on how far
  self-play runs ahead of training.  A pull hands over whole chunks until the actor's share is met; an actor that has
  fallen behind plays until it has its share (the learner waits, as the reference's trainer does).

Staleness: a move is searched with the last network the actor received, so a row is at most
`weight_sync_steps` steps old when it is played plus the time it waits in the backlog -- at most
`(world - 1) * ahead_rows / rows_consumed_per_step` steps.  (The reference's bound is one game: a worker picks up the
live tensors whenever it evaluates.)

Lock-step mode (`selfplay_mode: "lockstep"`, policy_trainer.serve_selfplay) stays as the deterministic mode: there every
production is announced, preceded by a weight broadcast, and played by all ranks including rank 0.
"""
import logging
import time
from collections import deque
from typing import Dict, Optional

import numpy as np
import torch

from . import distributed as azdist

# metric keys a production reports (DeviceReplayBuffer.consume's `st`), summed over chunks and ranks
_SUM_KEYS = ("games", "plies", "game_errors", "seconds", "sum_reward_last")


def actor_quota(size: int, rank: int, world: int) -> int:
    """Rows actor `rank` (1..world-1) owes to a pull of `size` rows: the learner owes none."""
    return azdist.shard_quota(size, rank - 1, world - 1)


class RecordBacklog:
    """Whole games, as chunks of packed records in device memory, waiting to be pulled (FIFO)."""

    def __init__(self):
        self.chunks = deque()          # (uint8 tensor [n, record_bytes], metrics dict)
        self.rows = 0

    def push(self, rec: torch.Tensor, metrics: Dict[str, float]) -> None:
        if rec.shape[0]:
            self.chunks.append((rec, metrics))
            self.rows += rec.shape[0]

    def take(self, quota: int):
        """Whole chunks, oldest first, until >= quota rows (all there is if fewer): (records, summed metrics)."""
        recs, total = [], {}
        have = 0
        while self.chunks and have < quota:
            rec, m = self.chunks.popleft()
            recs.append(rec)
            have += rec.shape[0]
            for k, v in m.items():
                total[k] = total.get(k, 0.0) + v
        self.rows -= have
        return recs, total


class Learner:
    """Rank 0's side: pulls, weight broadcasts, the end of the run."""

    def __init__(self, net: torch.nn.Module, weight_sync_steps: int = 50):
        if weight_sync_steps < 1:
            raise ValueError("weight_sync_steps must be >= 1")
        self.net = net
        self.every = int(weight_sync_steps)
        self.steps = 0
        self.weight_syncs = 0
        self.pulls = 0
        self.last_pull = None
        self._sending = None           # (work, snapshot) of the weight broadcast in flight

    def sync_weights(self) -> None:
        """Announce and send the network.  The send is asynchronous on this rank: a snapshot taken on the training
        stream goes out on the collective's own stream, so training does not wait for the actors to reach their next
        poll; the previous send is waited for first (it has long completed: one is in flight at most)."""
        if self._sending is not None:
            self._sending[0].wait()
        azdist.lead(azdist.OP_WEIGHTS, self.steps)
        self._sending = azdist.send_weights_async(self.net)
        self.weight_syncs += 1

    def after_step(self) -> None:
        """Once per optimizer step: the network goes out every `weight_sync_steps` steps."""
        self.steps += 1
        if self.steps % self.every == 0:
            self.sync_weights()

    def pull(self, size: int, device: torch.device, record_bytes: int):
        """Announce a pull of `size` rows and take part in its collectives with nothing to give.  Returns the
        per-rank record tensors, their row counts and the actors' summed production metrics.  Blocks until every actor
        has reached its next poll (at most one pool move of the slowest one) and has its share -- `Player.read`'s
        contract in the reference, whose trainer waits the same way when the generator is behind."""
        t0 = time.perf_counter()
        azdist.lead(azdist.OP_PULL, int(np.ceil(size)))
        parts, counts = azdist.gather_records(torch.empty((0, record_bytes), dtype=torch.uint8, device=device), dst=0)
        metrics = azdist.all_reduce_metrics({})
        self.pulls += 1
        world = len(counts)
        if "seconds" in metrics:
            metrics["seconds"] /= max(1, world - 1)          # the actors play side by side
        self.last_pull = {"rows_per_rank": counts, "seconds": time.perf_counter() - t0,
                          "bytes_gathered": int(sum(counts)) * record_bytes}
        return parts, counts, metrics

    def stop(self) -> None:
        """Training is over: everyone leaves with the trained network."""
        if self._sending is not None:
            self._sending[0].wait()
            self._sending = None
        azdist.lead(azdist.OP_STOP)
        azdist.broadcast_weights(self.net, src=0)

    def abort(self) -> bool:
        """Rank 0 failed: tell the actors to leave (they raise LeaderLost); see distributed.abort."""
        return azdist.abort()


def serve_selfplay_ahead(player, *, ahead_rows: Optional[int] = None, poll_plies: int = 1,
                         timeout: Optional[float] = None) -> Dict[str, float]:
    """Ranks != 0 of an actor / learner run: play ahead into the backlog, answer rank 0's announcements between
    moves, return when it says stop.  Nothing is produced before the first weight broadcast (the ranks may have
    started from different networks).  Returns counters: productions, rows produced, pulls, weight syncs, and how
    many productions ran between two announcements at most (> 0 = self-play really ran beside training)."""
    pol = player._device_policy()
    if pol is None:
        raise RuntimeError("actor / learner self-play needs a single agent whose Policy holds a HexNetwork")
    eng = player.device_engine()                      # agrees on the seed base: a collective, the learner does it too
    device = getattr(eng, "torch_device", None) or torch.device("cuda", eng.cfg.device)
    world, rank = torch.distributed.get_world_size(), torch.distributed.get_rank()
    ahead = int(ahead_rows) if ahead_rows else int(player.n_games)
    backlog = RecordBacklog()
    game_sums = getattr(eng, "game_metric_sums", None)
    stats = dict(productions=0, rows=0, pulls=0, weight_syncs=0, max_productions_between_announcements=0, waited=0)
    since = 0
    have_weights = False
    carry = {}                      # seconds / plies of productions that finished no game: added to the next chunk's metrics
    failure = []                    # the error of a production that raised: reported at the next pull (PeerFailed on all ranks)

    def produce():
        if failure:
            return
        try:
            n, st = eng.play_device(ahead, max_plies=poll_plies)
            stats["productions"] += 1
            for k in _SUM_KEYS:
                carry[k] = carry.get(k, 0.0) + float(st.get(k, 0.0))
            if not n:
                return
            rec = torch.empty((n, eng.record_bytes), dtype=torch.uint8, device=device)
            if device.type == "cuda":
                torch.cuda.synchronize(device)          # the caching allocator may recycle memory torch kernels still use
            eng.rows_pack(0, n, rec.data_ptr())
            m = dict(carry)
            carry.clear()
            if game_sums is not None:
                m.update({"game_" + k: float(v) for k, v in game_sums(n).items()})
            backlog.push(rec, m)
            stats["rows"] += n
        except Exception as exc:      # keep answering announcements: the learner must not be left in a collective
            logging.exception("actor %d: self-play failed; reporting it at the next pull", rank)
            failure.append(exc)

    pending = azdist.Pending()
    while True:
        if pending.ready() or not have_weights or backlog.rows >= ahead or failure:
            if not pending.ready():
                stats["waited"] += 1
            op, arg = pending.result(timeout)
            stats["max_productions_between_announcements"] = max(stats["max_productions_between_announcements"], since)
            since = 0
            if op == azdist.OP_STOP:
                if failure:
                    azdist.broadcast_weights(pol.net, src=0)     # the collective rank 0 is about to enter
                    raise failure[0]
                break
            if op == azdist.OP_WEIGHTS:
                azdist.broadcast_weights(pol.net, src=0)
                pol.net.weight_updates_outside_autograd = getattr(pol.net, "weight_updates_outside_autograd", 0) + 1
                player.prepare_device_engine(eng)                # pack them: the next move searches with them
                player.weight_syncs += 1
                stats["weight_syncs"] += 1
                have_weights = True
            elif op == azdist.OP_PULL:
                quota = actor_quota(arg, rank, world)
                while backlog.rows < quota and not failure:      # fallen behind: the learner waits for us
                    produce()
                recs, m = ([], {}) if failure else backlog.take(quota)
                rec = torch.cat(recs) if recs else torch.empty((0, eng.record_bytes), dtype=torch.uint8, device=device)
                try:
                    azdist.gather_records(rec, dst=0, failed=bool(failure))
                except azdist.PeerFailed:
                    if failure:
                        raise failure[0]
                    raise
                azdist.all_reduce_metrics(m)
                stats["pulls"] += 1
            else:
                raise RuntimeError("serve_selfplay_ahead: unknown announcement %d" % op)
            pending = azdist.Pending()
            continue
        produce()
        since += 1
    azdist.broadcast_weights(pol.net, src=0)          # everyone leaves with the trained network
    logging.info("actor %d: %s", rank, stats)
    return stats
