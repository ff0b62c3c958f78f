"""Multi-GPU glue: one process per GPU, torch.distributed over RCCL ("nccl" backend on ROCm).

Self-play games are independent, so ranks never talk on the data path.  The only exchanges are
(1) the replay-row all-gather when a Player.read is shared by all ranks and (2) a weight
broadcast when the trainer updates the network.  Rows travel as ONE fixed-size
byte record each (AZX_RECORD_BYTES, include/azx.h: uid i64 | reward f32 | color i16 | k i16 |
moves_prob f32[cells] | board u8[cells]), padded to the largest per-rank count: two collectives
per refill (counts, payload).  On the GPU path the records are packed from the engine's harvest
queue by a kernel, gathered as device tensors and appended to the HBM ring by a kernel
(DeviceReplayBuffer.consume); the host twins here serve Player.read's frames and the gloo tests.
"""
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


_in_data_collective = 0


def _data_collective(fn):
    """Marks the record / metric / weight collectives: a failure raised INSIDE one cannot be followed by an
    announcement (the other ranks are in that collective, not listening) -- see abort()."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        global _in_data_collective
        _in_data_collective += 1
        out = fn(*a, **k)
        _in_data_collective -= 1          # deliberately not in a finally: an exception leaves the mark set
        return out
    return wrapped


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _comm_device() -> torch.device:
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def record_bytes(cells: int) -> int:
    """AZX_RECORD_BYTES (include/azx.h)."""
    return _lib.record_bytes(cells)


def record_dtype(cells: int) -> np.dtype:
    """numpy view of one record -- the layout k_rows_pack / k_records_put use on the device:
    0 game_uid i64 | 8 reward f32 | 12 color i16 | 14 nlegal i16 | 16 moves_prob f32[cells] | board u8[cells] | pad."""
    return np.dtype({"names": ["game_uid", "reward", "color", "nlegal", "moves_prob", "board"],
                     "formats": ["<i8", "<f4", "<i2", "<i2", ("<f4", (cells,)), ("u1", (cells,))],
                     "offsets": [0, 8, 12, 14, 16, 16 + 4 * cells],
                     "itemsize": record_bytes(cells)})


def pack_rows(rows: Dict[str, np.ndarray], cells: int) -> np.ndarray:
    """rows as returned by Engine.play -> uint8 [P, record_bytes(cells)] (host twin of k_rows_pack)."""
    P = len(rows["reward"])
    rec = np.zeros(P, record_dtype(cells))
    rec["game_uid"] = rows["game_uid"]
    rec["reward"] = rows["reward"]
    rec["color"] = rows["color"]
    rec["nlegal"] = rows["nlegal"]
    rec["moves_prob"] = np.asarray(rows["moves_prob"], np.float32).reshape(P, cells)
    rec["board"] = np.asarray(rows["board"]).reshape(P, cells)
    return rec.view(np.uint8).reshape(P, record_bytes(cells))


def unpack_rows(rec: np.ndarray, board_size: int) -> Dict[str, np.ndarray]:
    cells = board_size * board_size
    r = np.ascontiguousarray(rec, np.uint8).reshape(-1, record_bytes(cells)).view(record_dtype(cells)).reshape(-1)
    P = len(r)
    return dict(board=r["board"].astype(np.int32).reshape(P, board_size, board_size),
                moves_prob=np.ascontiguousarray(r["moves_prob"]), reward=np.ascontiguousarray(r["reward"]),
                color=r["color"].astype(np.int32), nlegal=r["nlegal"].astype(np.int32),
                game_uid=np.ascontiguousarray(r["game_uid"]))


def empty_rows(board_size: int) -> Dict[str, np.ndarray]:
    """A rank with nothing to contribute still joins the collectives with a 0-row table."""
    return unpack_rows(np.zeros((0, record_bytes(board_size * board_size)), np.uint8), board_size)


class PeerFailed(RuntimeError):
    """Some rank's production failed between an announcement and its record collective.  The failed rank still joins
    the counts all-gather -- with -1 for its count -- and EVERY rank raises this right after it, at the same point of the
    protocol: nobody is left waiting in the payload collective (or in the metric one behind it) for a rank that will
    never arrive.  The failed rank re-raises its own error instead (DeviceReplayBuffer._fill_shared, Player.read,
    actor_learner.serve_selfplay_ahead)."""


def _gather_counts(n_local: int, dev: torch.device, failed: bool) -> List[int]:
    world = dist.get_world_size()
    counts_t = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts_t, torch.tensor([-1 if failed else int(n_local)], dtype=torch.int64, device=dev))
    counts = [int(c) for c in counts_t.tolist()]
    bad = [r for r, c in enumerate(counts) if c < 0]
    if bad:
        raise PeerFailed("self-play failed on rank%s %s: the production is abandoned on every rank"
                         % ("s" if len(bad) > 1 else "", ", ".join(str(r) for r in bad)))
    return counts


@_data_collective
def all_gather_records(rec: torch.Tensor, failed: bool = False) -> Tuple[List[torch.Tensor], List[int]]:
    """rec: uint8 [P, record_bytes] on the communication device (HBM under RCCL).  Two collectives:
    the per-rank counts, then the records padded to the largest count -- a direct all-gather over
    xGMI.  Returns the per-rank record tensors (views trimmed to their counts) and the counts, in rank
    order; nothing touches the host except the W counts.  `failed`: this rank has nothing to give because its
    production raised -- every rank then raises PeerFailed after the counts."""
    world = dist.get_world_size()
    if rec.is_cuda and dist.get_backend() != "nccl":
        # functional fallback (tests, bench.py with AZX_BENCH_BACKEND=gloo): the CPU backend cannot gather
        # device tensors, so the records make a host round trip; RCCL gathers them in HBM
        parts, counts = all_gather_records(rec.cpu(), failed)
        return [p.to(rec.device) for p in parts], counts
    dev = rec.device
    counts = _gather_counts(rec.shape[0], dev, failed)
    most = max(counts)
    if most == 0:
        return [rec[:0] for _ in range(world)], counts
    if rec.shape[0] == most:
        padded = rec.contiguous()
    else:
        padded = torch.zeros((most, rec.shape[1]), dtype=torch.uint8, device=dev)
        padded[:rec.shape[0]] = rec
    out = torch.empty((world * most, rec.shape[1]), dtype=torch.uint8, device=dev)   # rank blocks, concatenated
    dist.all_gather_into_tensor(out, padded)
    return [out[r * most:r * most + c] for r, c in enumerate(counts)], counts


@_data_collective
def gather_records(rec: torch.Tensor, dst: int = 0, failed: bool = False) -> Tuple[List[torch.Tensor], List[int]]:
    """The actor / learner pull: only rank `dst` needs the records, so each rank SENDS its block to it -- over xGMI
    that is one point-to-point transfer per actor, each on its own link into `dst`, instead of an all-gather that
    lands every actor's rows on every other actor as well ((W - 1) x the bytes, and a W x buffer on ranks that drop
    it).  Same two collectives: the per-rank counts (all ranks learn the padding), then the padded blocks.  Returns
    (per-rank record tensors, counts) on `dst` and ([], counts) elsewhere.  `failed`: as in all_gather_records."""
    world = dist.get_world_size()
    if rec.is_cuda and dist.get_backend() != "nccl":          # gloo over device tensors (tests): a host round trip
        parts, counts = gather_records(rec.cpu(), dst, failed)
        return [p.to(rec.device) for p in parts], counts
    dev = rec.device
    counts = _gather_counts(rec.shape[0], dev, failed)
    most = max(counts)
    me = dist.get_rank()
    if most == 0:
        return ([rec[:0] for _ in range(world)] if me == dst else []), counts
    if rec.shape[0] == most:
        padded = rec.contiguous()
    else:
        padded = torch.zeros((most, rec.shape[1]), dtype=torch.uint8, device=dev)
        padded[:rec.shape[0]] = rec
    if me != dst:
        dist.gather(padded, None, dst=dst)
        return [], counts
    blocks = [torch.empty((most, rec.shape[1]), dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.gather(padded, blocks, dst=dst)
    return [b[:c] for b, c in zip(blocks, counts)], counts


def all_gather_rows(rows: Dict[str, np.ndarray], board_size: int, failed: bool = False) -> Dict[str, np.ndarray]:
    """Host rows (Player.read's frame path): every rank contributes its rows -- possibly none -- and every
    rank gets all rows, in rank order.  `failed`: as in all_gather_records."""
    if not is_distributed():
        return rows
    cells = board_size * board_size
    local = torch.from_numpy(pack_rows(rows, cells)).to(_comm_device())
    parts, _ = all_gather_records(local, failed)
    rec = torch.cat(parts).cpu().numpy() if parts else np.zeros((0, record_bytes(cells)), np.uint8)
    return unpack_rows(rec, board_size)


@_data_collective
def all_reduce_metrics(metrics: Dict[str, float]) -> Dict[str, float]:
    """Sum the per-rank self-play metrics (parallel_player.py:50-51 sums over games).  Ranks may hold
    different key sets (one that played nothing holds none), so the dicts themselves are gathered."""
    if not is_distributed():
        return metrics
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, {k: float(v) for k, v in metrics.items()})
    total: Dict[str, float] = {}
    for part in parts:
        for k, v in part.items():
            total[k] = total.get(k, 0.0) + v
    return total


def broadcast_int(value: int, src: int = 0) -> int:
    """Every rank gets rank `src`'s integer (the shared seed base of the global game index)."""
    if not is_distributed():
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=_comm_device())
    dist.broadcast(t, src=src)
    return int(t.item())


# ---- training-time topology: rank 0 is the trainer, every rank plays ------------------------------
# The reference has ONE trainer whose workers always search with its live weights (parallel_player.py:36-38:
# the agents -- and with them the network's CUDA-IPC tensors -- are handed to the worker processes).  Here rank 0
# runs the optimizer; before every shared production it says what is to be produced (`lead`), broadcasts the
# network (parameters AND BatchNorm statistics) and all ranks play their share.  The other ranks sit in
# policy_trainer.serve_selfplay, doing what rank 0 announces (`follow`): their control flow never depends on
# their own arithmetic, so replicas cannot drift apart and nobody waits in a collective the others skip.
OP_STOP, OP_READ, OP_REFILL = 0, 1, 2
# actor / learner mode (azalea_amd/actor_learner.py): the followers play continuously, rank 0 only trains
OP_PULL, OP_WEIGHTS, OP_ABORT = 3, 4, 5


class LeaderLost(RuntimeError):
    """A follower waited longer than its timeout for rank 0's next announcement, or rank 0 said it is aborting."""


_control = None


def control_group():
    """The group announcements travel on: gloo (host side), so that a follower can POLL for the next one while its
    GPU keeps playing, and so that waiting has a timeout of ours instead of the RCCL watchdog's.  Records and weights
    stay on the default group (RCCL).  Created collectively: every rank must make its first lead() / follow() /
    control_group() call at the same point (train() does, right after the ranks split)."""
    global _control
    if _control is None:
        if dist.get_backend() == "gloo":
            _control = dist.group.WORLD
        else:        # gloo's own timeout must not fire before ours (AZX_FOLLOW_TIMEOUT may exceed its 30 min default)
            from datetime import timedelta
            _control = dist.new_group(backend="gloo", timeout=timedelta(seconds=follow_timeout() + 300.0))
    return _control


def reset_control_group() -> None:
    """Forget the control group (after destroy_process_group; tests that re-initialise torch.distributed)."""
    global _control
    _control = None


def rank() -> int:
    return dist.get_rank() if is_distributed() else 0


def follow_timeout() -> float:
    """Seconds a follower waits for an announcement before it gives up (AZX_FOLLOW_TIMEOUT, default 30 min)."""
    import os
    return float(os.environ.get("AZX_FOLLOW_TIMEOUT", "1800"))


def lead(op: int, arg: int = 0) -> None:
    """Rank 0: announce the next collective production (OP_READ: Player.read(arg); OP_REFILL: a shared
    DeviceReplayBuffer refill of arg rows; OP_PULL: hand over arg rows played ahead; OP_WEIGHTS: a weight broadcast
    follows; OP_STOP: training is over; OP_ABORT: rank 0 failed, leave without another collective)."""
    if is_distributed():
        t = torch.tensor([int(op), int(arg)], dtype=torch.int64)
        dist.broadcast(t, src=0, group=control_group())


def abort() -> bool:
    """Rank 0 failed: tell the other ranks to leave (their follow() raises LeaderLost) -- unless the failure came out
    of a data collective, in which case they are not listening and their own timeouts end them.  True if announced."""
    if not is_distributed() or _in_data_collective:
        return False
    lead(OP_ABORT)
    return True


class Pending:
    """A follower's posted receive of rank 0's next announcement: `ready()` polls, `result()` waits (with a
    timeout) and returns (op, arg)."""

    def __init__(self):
        self.t = torch.zeros(2, dtype=torch.int64)
        self.work = dist.broadcast(self.t, src=0, group=control_group(), async_op=True)

    def ready(self) -> bool:
        return self.work.is_completed()

    def result(self, timeout: Optional[float] = None) -> Tuple[int, int]:
        import time
        limit = follow_timeout() if timeout is None else timeout
        deadline = time.monotonic() + limit
        pause = 1e-4
        while not self.work.is_completed():
            if time.monotonic() > deadline:
                raise LeaderLost("no announcement from rank 0 within %.0f s" % limit)
            time.sleep(pause)
            pause = min(0.01, pause * 2)
        try:
            self.work.wait()
        except RuntimeError as exc:         # the control group itself failed (rank 0 died, gloo's own timeout)
            raise LeaderLost("control group failed: %s" % exc) from exc
        op, arg = (int(x) for x in self.t.tolist())
        if op == OP_ABORT:
            raise LeaderLost("rank 0 aborted the run")
        return op, arg


def follow(timeout: Optional[float] = None) -> Tuple[int, int]:
    """Ranks != 0: wait for rank 0's announcement (LeaderLost after `timeout` / AZX_FOLLOW_TIMEOUT seconds, or when
    rank 0 announces that it is aborting)."""
    return Pending().result(timeout)


def send_weights_async(net: torch.nn.Module):
    """Rank 0 of an actor / learner run: the same flat broadcast as `broadcast_weights`, from a SNAPSHOT of the network
    taken on the current stream, as an asynchronous collective -- the training stream does not wait for the actors to
    join (they notice the announcement between two pool moves, up to a few hundred ms later).  Returns (work, snapshot):
    keep both until the next call."""
    tensors = [t for t in net.state_dict().values() if t.is_floating_point()]
    dev = _comm_device()
    flat = torch.cat([t.detach().reshape(-1).to(dev, torch.float32) for t in tensors])
    return dist.broadcast(flat, src=rank(), async_op=True), flat


@_data_collective
def broadcast_weights(net: torch.nn.Module, src: int = 0) -> None:
    """One flat broadcast of parameters + buffers (2.1 MB for the 6x64 net): conv / linear weights and the
    BatchNorm affine terms AND running statistics -- self-play folds the statistics into the convolutions, so a
    rank with stale ones would search with another network."""
    if not is_distributed():
        return
    tensors = [t for t in net.state_dict().values() if t.is_floating_point()]
    dev = _comm_device()
    flat = torch.cat([t.detach().reshape(-1).to(dev, torch.float32) for t in tensors])
    dist.broadcast(flat, src=src)
    o = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[o:o + n].reshape(t.shape).to(t.device, t.dtype))
        o += n


def shard_quota(size: int, rank: Optional[int] = None, world: Optional[int] = None) -> int:
    """Positions this rank must produce so that the ranks together reach `size`."""
    if rank is None:
        rank, world = (dist.get_rank(), dist.get_world_size()) if is_distributed() else (0, 1)
    base, extra = divmod(int(np.ceil(size)), world)
    return base + (1 if rank < extra else 0)
