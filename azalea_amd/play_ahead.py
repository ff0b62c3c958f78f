"""Self-play that runs WHILE the trainer trains, on ONE GPU.

The reference keeps `num_workers + 1` games in flight during `supervised_step`: the trainer only blocks in `Player.read`
when the generator has fallen behind (azalea/process_pool.py:29-47, parallel_player.py:17-38, replay_buffer.py:121-132).
`actor_learner.py` gives a multi-GPU run that shape (ranks != 0 play ahead).  This is the same idea where there is one
GPU and one process:

* a host THREAD owns the engine's play calls: it keeps `azx_play_device` producing whole games, one pool move at a
  time, packs each harvest into a chunk of fixed-size records in HBM (`azx_rows_pack`) and appends it to a BACKLOG, and
  stops producing while the backlog holds `ahead_rows` rows (default: one row per pool slot, about one pool move's
  harvest) -- the bound on how far self-play runs ahead of training;
* the trainer's thread keeps the ring: `DeviceReplayBuffer.consume` takes whole chunks out of the backlog when its
  fresh-example counter asks for rows -- waiting only if the backlog is short -- and appends them with
  `azx_replay_put_records_async` ON THE TRAINING STREAM, where they are ordered with the collate reads and the step;
* every `weight_sync_steps` optimizer steps the trainer snapshots its network on the training stream (parameters and
  BatchNorm statistics); the play thread packs the snapshot into the engine before its next pool move.  No collective.
* the two kinds of work meet on the device, not on the host: the trainer's streams are created at the high stream
  priority, so a workgroup slot a retiring tower block frees goes to a waiting training kernel first.  Optionally
  (`reserve_cus`, off by default) the engine's streams carry a CU mask that leaves some compute units of every XCD free
  (`azx_reserve_cus`).

What it buys on ONE GPU, measured (6x64 on 11x11, 4096 games, 400 sims, batch 128, 10x oversampling;
tools/bench_train_loop.py over whole refill cycles, 4 800 steps, three alternating runs on one box,
profiles/r6_train_loop_overlap.json): **inline 638-692 steps/s, play-ahead 645-679; 634 vs 689 and 667 vs 735 on two other
boxes -- 0.95-1.10x, nothing that survives box variance.**  (A first reading of
"1.11x" came from 1 600-step windows, which hold 4 or 5 refills of a third of a second each by chance.)  The timeline
(tools/prof_overlap.sh, profiles/r6_train_loop_overlap_timeline.txt) says why: the two sides ARE resident together
66-69 % of the time and the tower does not slow down (7.7 ms a launch either way), but the training step does, 0.50 ->
1.5 ms: its ~31 dependent kernels have 256 workgroups each; when one of them ends, the slots it held are refilled with
tower blocks (200 us each) before the next one is launched, so every kernel of the chain waits for tower blocks to
retire again (~48 us against 12-16 alone).  The trainer, not self-play, bounds the loop (`waits` 0, the play thread
parked a third of the time), at the same rate the inline loop reaches by taking turns.  Leaving CUs free for the
trainer (`reserve_cus`) does not change that: an XCD deals workgroups to its shader engines round-robin whatever is
free elsewhere (tools/microbench/cu_mask.hip), so reserved CUs serve only the workgroups dealt to their own engine,
and one CU per shader engine costs the tower 14 %.  What would: a training step that keeps its slots -- one persistent
kernel with grid barriers instead of 31 launches.  One real defect the timeline exposed is fixed: three of the step's
kernels ran 1024-thread blocks of 80-88 registers a thread, which do not fit the half of a CU's register file one
retiring tower block frees, and waited 0.2-0.9 ms for a CU with both tower blocks gone (train_kernels.hip:
TRN_MID_THREADS).  With more than one GPU the same idea (actor_learner.py) does not share a device and has no such limit.

Staleness: a move is searched with the last snapshot packed, taken at most `weight_sync_steps` steps before the move
started (plus the snapshot the thread was busy packing); a row then waits in the backlog for at most
`(ahead_rows + one harvest) / rows_consumed_per_step` steps.  With the defaults at the reference's configuration (4096
slots, batch 128 / 10x oversampling = 12.8 rows per step, 50 steps) that is <= 50 + ~640 steps; the reference's bound
is one game's duration, its workers reading the live tensors whenever they evaluate.

Inline refills (`DeviceReplayBuffer.consume` playing its refill itself) stay as the deterministic mode, as `lockstep` is
for more than one rank: `config["selfplay_overlap"] = True` selects this one in `policy_trainer.train`.
"""
import logging
import threading
from typing import Dict, Optional

import torch

from .actor_learner import _SUM_KEYS, RecordBacklog


class PlayAhead:
    def __init__(self, player, engine, *, ahead_rows: Optional[int] = None, weight_sync_steps: int = 50,
                 poll_plies: int = 1, reserve_cus: int = 0):
        if weight_sync_steps < 1:
            raise ValueError("weight_sync_steps must be >= 1")
        self.player, self.engine = player, engine
        self.ahead = int(ahead_rows) if ahead_rows else int(engine.G)
        self.every = int(weight_sync_steps)
        self.poll_plies = int(poll_plies)
        self.reserve = int(reserve_cus)
        self.device = getattr(engine, "torch_device", None) or torch.device("cuda", engine.cfg.device)
        self.backlog = RecordBacklog()
        self._cv = threading.Condition()
        self._stop = False
        self._thread = None
        self._failure = None
        self._carry = {}
        # weight snapshot handshake: the trainer fills `_snap` and records `_snap_event` when `_snap_state` is "free";
        # the play thread packs it and hands it back
        self._snap = None
        self._snap_names = None
        self._snap_event = None
        self._snap_state = "free"
        self.steps = 0
        self.reserved_cus = 0
        self.stats = dict(productions=0, rows=0, takes=0, waits=0, wait_seconds=0.0, weight_syncs=0, snapshots_skipped=0,
                          max_backlog_rows=0)

    # ---- trainer's thread -------------------------------------------------------------------------
    def start(self) -> None:
        """Pack the trainer's current network (blocking, like an inline refill does), reserve the CUs and start playing."""
        if self._thread is not None:
            return
        self.player.prepare_device_engine(self.engine)
        if self.reserve > 0 and hasattr(self.engine, "reserve_cus"):
            self.reserved_cus = self.engine.reserve_cus(self.reserve)
            self._holding = True
        self._thread = threading.Thread(target=self._run, name="azx-play-ahead", daemon=True)
        self._thread.start()

    def after_step(self) -> None:
        """Once per optimizer step: every `weight_sync_steps` steps the network is snapshot on the current (training)
        stream for the play thread -- unless it is still packing the previous snapshot, then at the next step."""
        self.steps += 1
        if self.steps % self.every == 0:
            self._due = True
        if not self._due:
            return
        if self._snap_state != "free":
            self.stats["snapshots_skipped"] += 1
            return
        pol = self.player._device_policy()
        sd = {k: v for k, v in pol.net.state_dict().items() if v.dtype == torch.float32}
        if self._snap is None:
            self._snap_names = list(sd)
            self._snap = [torch.empty_like(v) for v in sd.values()]
        torch._foreach_copy_(self._snap, [sd[k].detach() for k in self._snap_names])
        ev = torch.cuda.Event() if self.device.type == "cuda" else None
        if ev is not None:
            ev.record(torch.cuda.current_stream(self.device))
        self._snap_event = ev
        self._snap_state = "ready"
        self._due = False
        with self._cv:
            self._cv.notify_all()           # a thread parked on a full backlog packs it right away

    _due = False
    _wanted = 0
    _holding = False
    _failure_raised = False

    def take(self, quota: int):
        """Whole chunks, oldest first, until >= quota rows: (list of record tensors, summed production metrics).
        Returns at once when the backlog holds enough; otherwise waits for the play thread (the reference's trainer
        waits in Player.read the same way when the generator is behind)."""
        import time
        with self._cv:
            self.stats["takes"] += 1
            if self.backlog.rows < quota and self._failure is None:
                self.stats["waits"] += 1
                t0 = time.perf_counter()
                self._wanted = quota            # a take larger than the bound: the thread plays until it is met
                self._cv.notify_all()
                while self.backlog.rows < quota and self._failure is None and not self._stop:
                    self._cv.wait(0.05)
                self._wanted = 0
                self.stats["wait_seconds"] += time.perf_counter() - t0
            if self._failure is not None:
                self._failure_raised = True
                raise self._failure
            recs, m = self.backlog.take(quota)
            self._cv.notify_all()               # room in the backlog again
        return recs, m

    def stop(self) -> None:
        """Stop playing (after the move in flight), give the engine its CUs back."""
        with self._cv:
            self._stop = True
            self._cv.notify_all()
        if self._thread is not None:
            self._thread.join()
            self._thread = None
        if self._failure is not None and not self._failure_raised:
            logging.warning("play-ahead: the play thread had failed (%r); no take() was left to raise it", self._failure)
        if self.reserved_cus and self._holding and hasattr(self.engine, "reserve_cus"):
            try:
                self.engine.reserve_cus(0)
            except Exception:                   # the engine may already be gone (Player.stop)
                logging.debug("play-ahead: could not release the reserved CUs", exc_info=True)
        self._holding = False

    # ---- play thread ------------------------------------------------------------------------------
    def _pack_snapshot(self) -> None:
        if self._snap_state != "ready":
            return
        if self._snap_event is not None:
            self._snap_event.synchronize()          # the copies were queued on the training stream
        self.engine.set_weights({k: (t.data_ptr(), t.numel()) for k, t in zip(self._snap_names, self._snap)},
                                on_device=True, sync=False)
        self.player.weight_syncs += 1
        self.stats["weight_syncs"] += 1
        self._snap_state = "free"

    def _produce(self) -> None:
        eng = self.engine
        n, st = eng.play_device(self.ahead, max_plies=self.poll_plies)
        self.stats["productions"] += 1
        for k in _SUM_KEYS:
            self._carry[k] = self._carry.get(k, 0.0) + float(st.get(k, 0.0))
        if not n:
            return
        if self.device.type == "cuda":
            with torch.cuda.stream(self._alloc_stream):     # chunks come from a pool of their own (see DeviceReplayBuffer._take_ahead)
                rec = torch.empty((n, eng.record_bytes), dtype=torch.uint8, device=self.device)
        else:
            rec = torch.empty((n, eng.record_bytes), dtype=torch.uint8)
        eng.rows_pack(0, n, rec.data_ptr())
        m, self._carry = dict(self._carry), {}
        sums = getattr(eng, "game_metric_sums", None)
        if sums is not None:
            m.update({"game_" + k: float(v) for k, v in sums(n).items()})
        with self._cv:
            self.backlog.push(rec, m)
            self.stats["rows"] += n
            self.stats["max_backlog_rows"] = max(self.stats["max_backlog_rows"], self.backlog.rows)
            self._cv.notify_all()

    def _run(self) -> None:
        try:
            if self.device.type == "cuda":
                torch.cuda.set_device(self.device)
                self._alloc_stream = torch.cuda.Stream(self.device)
            while True:
                with self._cv:
                    while (self.backlog.rows >= max(self.ahead, self._wanted) and not self._stop
                           and self._snap_state != "ready"):
                        self._cv.wait(0.02)
                    if self._stop:
                        return
                    full = self.backlog.rows >= max(self.ahead, self._wanted)
                self._pack_snapshot()
                if not full:
                    self._produce()
        except BaseException as exc:      # noqa: BLE001 -- handed to the trainer's thread, which raises it from take()
            logging.exception("play-ahead thread failed")
            with self._cv:
                self._failure = exc
                self._cv.notify_all()

    def counters(self) -> Dict[str, float]:
        return dict(self.stats, steps=self.steps, ahead_rows=self.ahead, weight_sync_steps=self.every,
                    reserved_cus=self.reserved_cus)
