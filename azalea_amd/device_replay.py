"""Replay buffer that stays in HBM.

Same contract as ReplayBuffer (azalea/replay_buffer.py:107-149) for the trainer loop of
azalea/policy_trainer.py:51-90 -- `len`, `put`, `consume(num_examples, player)`, `write_idx`,
`fresh_counter`, `state_dict` -- but the rows live in the engine's device ring and minibatches are
collated on the GPU (azx_replay_collate) instead of by DataLoader workers running
prep.torch_batch_replays.  `loader(batch_size)` yields the batches a
`DataLoader(buffer, batch_size, shuffle=True, collate_fn=torch_batch_replays)` would: the same
random permutation (RandomSampler seeds a private generator from the global torch RNG), the same
keys, dtypes and zero padding to the batch's widest row, as device tensors.
"""
from collections import deque
from typing import Dict, Iterator

import numpy as np
import torch

from . import distributed as azdist
from .replay_buffer import ReplayDataFrame

_KEYS = ("color", "legal_moves", "result", "board", "moves_prob", "reward")


class DeviceReplayBuffer:
    learner = None      # (class defaults: a buffer built around an existing ring without __init__ plays its refills inline)
    ahead = None

    def __init__(self, engine, capacity: int, contents: ReplayDataFrame = None, shared: bool = True):
        """`engine`: azalea_amd.engine.Engine whose self-play feeds the buffer (Player._get_engine).
        `contents`: initial rows, like ReplayBuffer(contents).  `shared`: under torch.distributed every
        rank plays its share of a refill and all ranks' rows enter every rank's ring."""
        self.engine = engine
        self.capacity = int(capacity)
        engine.replay_create(self.capacity)
        self.fresh_counter = 0
        self.shared = shared
        self.learner = None             # actor_learner.Learner: refills are PULLED from the actors' backlogs (rank 0 only)
        self.ahead = None               # play_ahead.PlayAhead: refills are TAKEN from this process's own play thread
        self._retired = deque()         # (record chunks, event): chunks whose ring put may still be in flight
        self.last_exchange = None       # timing of the last shared refill (bench / diagnostics)
        self._mover_view = False
        self.device = getattr(engine, "torch_device", None) or torch.device("cuda", engine.cfg.device)
        if contents is not None and len(contents):
            self.put(contents)
            self.fresh_counter = 0          # ReplayBuffer.__init__ starts with nothing fresh

    @property
    def mover_view(self) -> bool:
        """NOT the reference's batch (off by default): minibatches hand out the second player's rows in the view the
        search evaluates them in (azx_replay_set_mover_view, include/azx.h; policy_trainer.train's
        config["train_mover_view"]).  The ring itself always holds absolute colours."""
        return self._mover_view

    @mover_view.setter
    def mover_view(self, on: bool) -> None:
        self.engine.replay_set_mover_view(bool(on))
        self._mover_view = bool(on)

    # ---- ReplayBuffer surface -----------------------------------------------------------------
    def __len__(self) -> int:
        return self.engine.replay_state()["size"]

    @property
    def write_idx(self) -> int:
        return self.engine.replay_state()["write_idx"]

    def put(self, new_data: ReplayDataFrame) -> None:
        """Host rows into the ring in FIFO order (replay_buffer.py:134-149)."""
        P = len(new_data)
        if P == 0:
            return
        n = self.engine.n
        board = np.stack([np.asarray(s.board, np.int32).reshape(n * n) for s in new_data.state])
        color = np.array([s.color for s in new_data.state], np.int32)
        nlegal = np.array([len(p) for p in new_data.moves_prob], np.int32)
        prob = np.zeros((P, n * n), np.float32)
        for i, p in enumerate(new_data.moves_prob):
            prob[i, :len(p)] = p
        self.engine.replay_put(board, color, nlegal, prob, np.asarray(new_data.reward, np.float32))
        self.fresh_counter += P

    def consume(self, num_examples, player=None) -> Dict[str, float]:
        """replay_buffer.py:121-132; the refill is played and stored on the device."""
        self.fresh_counter -= num_examples
        refill = max(0, num_examples - self.fresh_counter)
        if not refill:
            return {}
        if self.ahead is None and getattr(self, "_async_reads", False):
            torch.cuda.synchronize(self.device)        # collate_async reads queued on torch's stream come first
            self._async_reads = False
        if self.ahead is not None:                     # (its put is queued on the stream those reads are on: ordered)
            rows, st = self._take_ahead(int(np.ceil(refill)))
        elif self.learner is not None:
            rows, st = self._pull(int(np.ceil(refill)))
        elif self.shared and azdist.is_distributed():
            rows, st = self.refill_shared(int(np.ceil(refill)), player)
        else:
            if player is not None and hasattr(player, "prepare_device_engine"):
                player.prepare_device_engine(self.engine)     # push the trainer's current weights
            rows, st = self.engine.replay_fill(int(np.ceil(refill)))
            st = dict(st, **self._game_sums(rows))
        self.fresh_counter += rows
        out = {"games": float(st.get("games", 0.0)), "reward": st.get("sum_reward_last", 0.0),
               "moves_per_game": float(rows), "seconds_per_game": st.get("seconds", 0.0),
               "game_error": float(st.get("game_errors", 0.0))}
        out.update({k: float(st.get("game_" + k, 0.0)) for k in self._SEARCH_KEYS})
        return out

    # the per-ply search metrics play_game averages per game (search_tree.py:109-112, mcts.py:291, policy.py:164)
    _SEARCH_KEYS = ("search_value", "search_root_width", "action_logprob", "search_root_visits",
                    "search_tree_nodes", "search_root_children")

    def _game_sums(self, rows):
        """Per-game means of the search metrics, summed over the games just harvested (what Player.read's
        metrics hold); an engine without the per-row metric queue contributes zeros."""
        f = getattr(self.engine, "game_metric_sums", None)
        sums = f(rows) if (f is not None and rows) else {}
        return {"game_" + k: float(sums.get(k, 0.0)) for k in self._SEARCH_KEYS}

    def refill_shared(self, refill: int, player=None):
        """One refill of `refill` rows played by all ranks.  With a leader / follower Player (training: rank 0 is
        the trainer) rank 0 announces it and broadcasts its network first; every rank then packs ITS current
        module into its engine -- after the broadcast these are the same weights everywhere."""
        if player is not None and hasattr(player, "announce"):
            player.announce(azdist.OP_REFILL, refill)
        if player is not None and hasattr(player, "prepare_device_engine"):
            player.prepare_device_engine(self.engine)
        return self._fill_shared(refill)

    def _pull(self, refill: int):
        """Actor / learner mode (azalea_amd/actor_learner.py): this rank trains and does not play; the rows come out
        of the other ranks' backlogs -- one announcement, the same two record collectives as a shared refill, and
        every rank's records appended to the ring in rank order."""
        parts, counts, st = self.learner.pull(refill, self.device, self.engine.record_bytes)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)         # the gathered records were written on torch's / RCCL's streams
        for part, count in zip(parts, counts):
            if count:
                self.engine.replay_put_records(count, part.data_ptr())
        self.last_exchange = dict(self.learner.last_pull)
        return int(sum(counts)), st

    def _take_ahead(self, refill: int):
        """Play-ahead mode (azalea_amd/play_ahead.py): whole chunks out of this process's own backlog -- waiting only when
        it is short -- appended to the ring by a kernel queued on the CURRENT (training) stream: behind the collate reads
        already queued there, ahead of the next ones, and never on the engine's stream, which the play thread keeps busy.
        A chunk's memory is released once the stream has passed its put."""
        chunks, st = self.ahead.take(refill)
        cuda = self.device.type == "cuda"
        stream = torch.cuda.current_stream(self.device) if cuda else None
        rows = 0
        for rec in chunks:
            n = int(rec.shape[0])
            if cuda:
                self.engine.replay_put_records_async(n, rec.data_ptr(), stream.cuda_stream)
            else:
                self.engine.replay_put_records(n, rec.data_ptr())
            rows += n
        if cuda and chunks:
            ev = torch.cuda.Event()
            ev.record(stream)
            self._retired.append((chunks, ev))
            while self._retired and self._retired[0][1].query():
                self._retired.popleft()
        return rows, st

    def _fill_shared(self, refill: int):
        """One refill played by all ranks: each rank plays its share of whole games into its harvest
        queue, packs the rows into fixed-size records ON THE DEVICE (k_rows_pack), the ranks all-gather
        the record buffers (RCCL over xGMI; two collectives: counts, payload) and every rank appends
        all of them, in rank order, to its HBM ring (k_records_put).  This replaces the reference's
        pickled frames over worker pipes (process_pool.py:31-47, parallel_player.py:48-49); nothing but
        the W row counts and the summed metrics visits the host."""
        import time
        eng = self.engine
        quota = azdist.shard_quota(refill)
        t0 = time.perf_counter()
        failure = None
        n_local, st = 0, {}
        t1 = t0
        try:
            if quota > 0:
                n_local, st = eng.play_device(quota)
            t1 = time.perf_counter()
            rec = torch.empty((n_local, eng.record_bytes), dtype=torch.uint8, device=self.device)
            if n_local:
                if self.device.type == "cuda":
                    torch.cuda.synchronize(self.device)     # the caching allocator may recycle memory torch kernels still use
                eng.rows_pack(0, n_local, rec.data_ptr())
        except Exception as exc:
            # AZX_ERANGE, a full tree arena, a HIP error: the other ranks are already on their way into the record
            # collectives of this announced refill -- join the first with the failure mark, so that every rank leaves it
            # with PeerFailed instead of hanging until the RCCL watchdog (distributed.PeerFailed)
            failure = exc
            rec = torch.empty((0, eng.record_bytes), dtype=torch.uint8, device=self.device)
        t2 = time.perf_counter()
        try:
            parts, counts = azdist.all_gather_records(rec, failed=failure is not None)
        except azdist.PeerFailed:
            if failure is not None:
                raise failure
            raise
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        t3 = time.perf_counter()
        for part, count in zip(parts, counts):
            if count:
                eng.replay_put_records(count, part.data_ptr())
        t4 = time.perf_counter()
        self.last_exchange = {"rows_per_rank": counts, "play_seconds": t1 - t0, "pack_seconds": t2 - t1,
                              "allgather_seconds": t3 - t2, "ring_put_seconds": t4 - t3,
                              "bytes_gathered": int(sum(counts)) * eng.record_bytes}
        keys = ("games", "plies", "game_errors", "seconds", "sum_reward_last") + tuple("game_" + k for k in self._SEARCH_KEYS)
        st = dict(st, **self._game_sums(n_local))
        total = azdist.all_reduce_metrics({k: float(st.get(k, 0.0)) for k in keys})
        total["seconds"] /= max(1, len(counts))         # ranks play side by side
        return int(sum(counts)), total

    # ---- minibatches --------------------------------------------------------------------------
    def sample(self, indices) -> Dict[str, torch.Tensor]:
        """prep.torch_batch_replays([self[i] for i in indices]) as device tensors."""
        idx = np.asarray(indices, np.int64).reshape(-1)
        B, cells = len(idx), self.engine.n * self.engine.n
        dev = self.device
        if self.ahead is not None:
            return self._sample_on_stream(idx)
        out = dict(color=torch.empty(B, dtype=torch.int64, device=dev),
                   legal_moves=torch.empty((B, cells), dtype=torch.int32, device=dev),
                   result=torch.empty(B, dtype=torch.int64, device=dev),
                   board=torch.empty((B, cells), dtype=torch.int32, device=dev),
                   moves_prob=torch.empty((B, cells), dtype=torch.float32, device=dev),
                   reward=torch.empty(B, dtype=torch.float32, device=dev))
        torch.cuda.synchronize(dev)      # the allocator may hand back memory torch kernels still use
        k = self.engine.replay_collate(idx, {name: t.data_ptr() for name, t in out.items()})
        n = self.engine.n
        out["legal_moves"] = out["legal_moves"][:, :k].contiguous()
        out["moves_prob"] = out["moves_prob"][:, :k].contiguous()
        out["board"] = out["board"].view(B, n, n)
        return out

    def _sample_on_stream(self, idx) -> Dict[str, torch.Tensor]:
        """`sample` while a play thread keeps the engine's stream busy: the collate runs on the current stream (where the
        ring puts of play-ahead mode are), the batch's widest row is found on the host side of it."""
        B, n = len(idx), self.engine.n
        cells, dev = n * n, self.device
        out = dict(color=torch.empty(B, dtype=torch.int64, device=dev),
                   legal_moves=torch.empty((B, cells), dtype=torch.int32, device=dev),
                   result=torch.empty(B, dtype=torch.int64, device=dev),
                   board=torch.empty((B, cells), dtype=torch.int32, device=dev),
                   moves_prob=torch.empty((B, cells), dtype=torch.float32, device=dev),
                   reward=torch.empty(B, dtype=torch.float32, device=dev))
        self.collate_async(idx, out)
        k = max(1, int((out["legal_moves"] > 0).sum(1).max().item())) if B else 0
        out["legal_moves"] = out["legal_moves"][:, :k].contiguous()
        out["moves_prob"] = out["moves_prob"][:, :k].contiguous()
        out["board"] = out["board"].view(B, n, n)
        return out

    def collate_into(self, indices, out: Dict[str, torch.Tensor]) -> int:
        """`sample` without allocations: the six batch tensors are the caller's (full width: legal_moves /
        moves_prob / board [B, cells], contiguous, the dtypes `sample` uses) and are written in place, zero padded to
        all cells; returns the batch's widest row.  For a consumer with static input buffers (GraphedTrainStep)."""
        idx = np.asarray(indices, np.int64).reshape(-1)
        B, cells = len(idx), self.engine.n * self.engine.n
        want = dict(color=((B,), torch.int64), legal_moves=((B, cells), torch.int32), result=((B,), torch.int64),
                    board=((B, cells), torch.int32), moves_prob=((B, cells), torch.float32), reward=((B,), torch.float32))
        for name, (shape, dtype) in want.items():
            t = out[name]
            if t.dtype != dtype or t.numel() != int(np.prod(shape)) or not t.is_contiguous() or t.device != self.device:
                raise ValueError("collate_into: '%s' must be a contiguous %s tensor of %s elements on %s"
                                 % (name, dtype, shape, self.device))
        torch.cuda.synchronize(self.device)      # whatever still reads the buffers (a graph replay) has finished
        return int(self.engine.replay_collate(idx, {name: out[name].data_ptr() for name in want}))

    def collate_async(self, indices, out: Dict[str, torch.Tensor]) -> None:
        """`collate_into` enqueued on torch's current stream and not synchronised (azx_replay_collate_async): the
        consumer's work on that stream -- NativeTrainStep's kernels -- runs after it, the host moves on.  Full-width
        rows, no max_k.  Ring writes wait for these reads: `consume` synchronises before a refill."""
        idx = np.asarray(indices, np.int64).reshape(-1)
        B, cells = len(idx), self.engine.n * self.engine.n
        want = dict(color=((B,), torch.int64), legal_moves=((B, cells), torch.int32), result=((B,), torch.int64),
                    board=((B, cells), torch.int32), moves_prob=((B, cells), torch.float32), reward=((B,), torch.float32))
        for name, (shape, dtype) in want.items():
            t = out[name]
            if t.dtype != dtype or t.numel() != int(np.prod(shape)) or not t.is_contiguous() or t.device != self.device:
                raise ValueError("collate_async: '%s' must be a contiguous %s tensor of %s elements on %s"
                                 % (name, dtype, shape, self.device))
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.engine.replay_collate_async(idx, {name: out[name].data_ptr() for name in want}, stream)
        self._async_reads = True

    def epoch_indices(self) -> np.ndarray:
        """The order a fresh `iter(DataLoader(..., shuffle=True))` visits the rows in: the loader
        draws its worker base seed from the global RNG first, then RandomSampler seeds a private
        generator from it and takes one randperm."""
        torch.empty((), dtype=torch.int64).random_()
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
        gen = torch.Generator()
        gen.manual_seed(seed)
        return torch.randperm(len(self), generator=gen).numpy()

    def loader(self, batch_size: int, drop_last: bool = False) -> Iterator[Dict[str, torch.Tensor]]:
        """One epoch of shuffled minibatches (policy_trainer.py:51-56, :82)."""
        order = self.epoch_indices()
        for s in range(0, len(order), batch_size):
            chunk = order[s:s + batch_size]
            if drop_last and len(chunk) < batch_size:
                return
            yield self.sample(chunk)

    # ---- checkpointing (replay_buffer.py:151-165) ---------------------------------------------
    def rows(self, indices=None) -> Dict[str, np.ndarray]:
        """Host copy of ring rows (all rows held by default) in collated form."""
        idx = np.arange(len(self)) if indices is None else np.asarray(indices, np.int64)
        if not len(idx):
            return {}
        view = self._mover_view
        if view:
            self.mover_view = False      # checkpoints hold the rows as the ring does: absolute colours
        try:
            return {k: v.cpu().numpy() for k, v in self.sample(idx).items()}
        finally:
            if view:
                self.mover_view = True

    def state_dict(self) -> Dict:
        return {"rows": self.rows(), "write_idx": self.write_idx, "fresh_counter": self.fresh_counter}

    def load_state_dict(self, state: Dict) -> None:
        rows = state["rows"]
        self.engine.replay_create(self.capacity)
        if rows:
            nlegal = (rows["legal_moves"] > 0).sum(1).astype(np.int32)
            P, n = len(nlegal), self.engine.n
            prob = np.zeros((P, n * n), np.float32)
            prob[:, :rows["moves_prob"].shape[1]] = rows["moves_prob"]
            self.engine.replay_put(rows["board"].reshape(P, n * n), rows["color"].astype(np.int32), nlegal,
                                   prob, rows["reward"])
            self.engine.replay_set_state(P, int(state["write_idx"]) % self.capacity)
        self.fresh_counter = state["fresh_counter"]
