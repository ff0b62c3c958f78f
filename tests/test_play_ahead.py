"""azalea_amd/play_ahead.py on the CPU: the thread protocol of self-play beside training on one GPU (the backlog bound, whole
chunks, the trainer waiting only when the backlog is short, the weight snapshot handshake, a failing play thread) with a
stub engine; the GPU side -- the CU mask, the stream priorities, the async ring put -- is tests/test_gpu_play_ahead.py."""
import ctypes
import threading
import time

import numpy as np
import pytest
import torch

from azalea_amd import _lib
from azalea_amd.device_replay import DeviceReplayBuffer
from azalea_amd.play_ahead import PlayAhead


class Cfg:
    device = 0


class StubEngine:
    """Harvest queue and ring as numpy tables of fixed-size records in host memory (tests/test_distributed_gloo.py's
    idea): each play_device call 'finishes' `per_call` rows, tagged with the weights in force and a running row id."""
    n, cells, G = 3, 9, 16

    def __init__(self, per_call=8, delay=0.002, fail_at=None):
        self.cfg = Cfg()
        self.record_bytes = _lib.record_bytes(self.cells)
        self.torch_device = torch.device("cpu")
        self.per_call, self.delay, self.fail_at = per_call, delay, fail_at
        self.calls = 0
        self.next_row = 0
        self.weights = None            # sum of the tensors last packed
        self.packed = []
        self.queue = None
        self.ring = None
        self.lock = threading.Lock()
        self.in_play = False
        self.overlapped_puts = 0

    def set_weights(self, tensors, on_device=False, sync=True):
        tot = 0.0
        for name, (ptr, cnt) in tensors.items():
            tot += float(np.ctypeslib.as_array((ctypes.c_float * cnt).from_address(ptr)).astype(np.float64).sum())
        self.weights = tot
        self.packed.append(tot)

    def play_device(self, min_positions, max_plies=0):
        self.calls += 1
        if self.fail_at is not None and self.calls == self.fail_at:
            raise RuntimeError("injected AZX_ERANGE")
        self.in_play = True
        time.sleep(self.delay)
        self.in_play = False
        n = self.per_call if self.calls % 3 else 0          # every third move finishes no game
        q = np.zeros((n, self.record_bytes), np.uint8)
        ids = np.arange(self.next_row, self.next_row + n, dtype=np.int64)
        q[:, :8] = ids.view(np.uint8).reshape(n, 8)
        q[:, 8:12] = np.full(n, np.float32(self.weights or 0.0)).view(np.uint8).reshape(n, 4)
        self.next_row += n
        self.queue = q
        return n, dict(games=1.0 if n else 0.0, plies=float(self.G), game_errors=0.0, seconds=self.delay, sum_reward_last=1.0)

    def rows_pack(self, first, n, ptr):
        np.ctypeslib.as_array((ctypes.c_uint8 * (n * self.record_bytes)).from_address(ptr))[:] = self.queue[first:first + n].reshape(-1)

    def replay_create(self, capacity):
        self.ring, self.size, self.write = np.zeros((capacity, self.record_bytes), np.uint8), 0, 0

    def replay_state(self):
        return dict(capacity=len(self.ring), size=self.size, write_idx=self.write)

    def replay_put_records(self, n, ptr):
        if self.in_play:
            self.overlapped_puts += 1                        # the trainer's put ran while the play thread was in a move
        rec = np.ctypeslib.as_array((ctypes.c_uint8 * (n * self.record_bytes)).from_address(ptr)).reshape(n, self.record_bytes)
        for r in rec:
            self.ring[self.write] = r
            self.write = (self.write + 1) % len(self.ring)
            self.size = min(len(self.ring), self.size + 1)


class StubPolicy:
    def __init__(self):
        self.net = torch.nn.Linear(4, 3)


class StubPlayer:
    def __init__(self):
        self.pol = StubPolicy()
        self.weight_syncs = 0
        self.prepared = 0

    def _device_policy(self):
        return self.pol

    def prepare_device_engine(self, engine):
        self.prepared += 1
        engine.set_weights({k: (v.data_ptr(), v.numel()) for k, v in self.pol.net.state_dict().items()})


def _wait(cond, seconds=5.0):
    t0 = time.monotonic()
    while not cond():
        assert time.monotonic() - t0 < seconds, "timed out"
        time.sleep(0.002)


def test_the_play_thread_runs_ahead_up_to_its_bound_and_hands_over_whole_chunks():
    E, P = StubEngine(), StubPlayer()
    A = PlayAhead(P, E, ahead_rows=20, weight_sync_steps=3, reserve_cus=0)
    A.start()
    assert P.prepared == 1                               # the trainer's network is packed before the first move
    _wait(lambda: A.backlog.rows >= 20)
    time.sleep(0.05)
    calls, rows = E.calls, A.backlog.rows
    assert 20 <= rows < 20 + E.per_call                  # it stopped at the bound: at most one harvest beyond it
    time.sleep(0.05)
    assert E.calls == calls                              # ... and stays parked while nothing is taken
    recs, m = A.take(17)                                 # whole chunks, oldest first, until >= 17 rows
    got = sum(r.shape[0] for r in recs)
    assert got == 24 and all(r.shape[0] == E.per_call for r in recs)
    ids = np.concatenate([r.numpy()[:, :8].copy().view(np.int64).ravel() for r in recs])
    assert np.array_equal(ids, np.arange(24))
    # moves 1, 2 and 4 finished games; move 3 finished none and its plies / seconds travel with move 4's chunk
    # (not dropped)
    assert m["plies"] == 4 * E.G and m["games"] == 3.0
    _wait(lambda: A.backlog.rows >= 20)                  # room again: it plays on
    A.stop()
    assert A.counters()["max_backlog_rows"] < 20 + E.per_call and A.counters()["waits"] == 0


def test_the_trainer_waits_only_when_the_backlog_is_short():
    E, P = StubEngine(delay=0.02), StubPlayer()
    A = PlayAhead(P, E, ahead_rows=8, reserve_cus=0)
    A.start()
    t0 = time.monotonic()
    recs, _ = A.take(24)                                 # more than the backlog may ever hold: the thread keeps producing
    assert sum(r.shape[0] for r in recs) >= 24 and time.monotonic() - t0 >= 0.05
    assert A.stats["waits"] == 1 and A.stats["wait_seconds"] > 0
    A.stop()


def test_weight_snapshots_reach_the_engine_between_moves():
    E, P = StubEngine(), StubPlayer()
    A = PlayAhead(P, E, ahead_rows=8, weight_sync_steps=2, reserve_cus=0)
    A.start()
    first = E.weights
    with torch.no_grad():
        for p in P.pol.net.parameters():
            p.add_(1.0)                                  # an optimizer step
    A.after_step()                                       # step 1: nothing due
    assert A._snap_state == "free"
    want = float(sum(v.double().sum() for v in P.pol.net.state_dict().values()))
    A.after_step()                                       # step 2: snapshot
    with torch.no_grad():
        for p in P.pol.net.parameters():
            p.add_(100.0)                                # later steps do not leak into the snapshot being packed
    _wait(lambda: len(E.packed) >= 2)
    assert abs(E.packed[1] - want) < 1e-3 and abs(E.packed[1] - first) > 1.0
    assert P.weight_syncs == 1 and A.stats["weight_syncs"] == 1
    A.stop()


def test_a_failing_play_thread_is_raised_in_the_trainer():
    E, P = StubEngine(fail_at=3), StubPlayer()
    A = PlayAhead(P, E, ahead_rows=1000, reserve_cus=0)
    A.start()
    with pytest.raises(RuntimeError, match="injected AZX_ERANGE"):
        for _ in range(100):
            A.take(8)
    A.stop()


def test_consume_takes_its_refills_from_the_backlog_in_fifo_order():
    """DeviceReplayBuffer.consume in play-ahead mode: the fresh-example rule of replay_buffer.py:121-132, the rows out of
    the backlog in the order they were played, while the play thread keeps playing."""
    E, P = StubEngine(delay=0.005), StubPlayer()
    buf = DeviceReplayBuffer(E, capacity=64, shared=False)
    A = PlayAhead(P, E, ahead_rows=16, reserve_cus=0)
    buf.ahead = A
    A.start()
    total = 0
    for _ in range(40):
        m = buf.consume(2.5)
        if m:
            total += int(m["moves_per_game"])
            assert m["games"] >= 1.0
    A.stop()
    assert total >= 40 * 2.5 - 2.5 and len(buf) == min(64, total)
    assert buf.fresh_counter == total - 40 * 2.5
    held = E.ring[:E.size, :8].copy().view(np.int64).ravel()
    order = np.r_[held[E.write:], held[:E.write]] if E.size == 64 else held
    assert np.array_equal(order, np.arange(total - len(order), total))     # FIFO, oldest overwritten
    assert E.overlapped_puts > 0                          # ring puts ran while a move was in flight
