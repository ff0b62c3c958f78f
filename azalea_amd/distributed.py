"""Multi-GPU glue: one process per GPU, torch.distributed over RCCL ("nccl" backend on ROCm).

Self-play games are independent, so ranks never talk on the data path.  The only exchanges are
(1) the replay-row all-gather when a Player.read is shared by all ranks and (2) a weight
broadcast when the trainer updates the network.  Rows travel as ONE fixed-size
byte record each (board u8[cells] | moves_prob f32[cells] | reward f32 | color i16 | k i16 |
uid i64), padded to the largest per-rank count: two collectives per refill (counts, payload).
Works with the gloo backend on CPU too (tests).
"""
from typing import Dict, Optional

import numpy as np
import torch
import torch.distributed as dist


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _comm_device() -> torch.device:
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def record_bytes(cells: int) -> int:
    return cells + 4 * cells + 4 + 2 + 2 + 8


def pack_rows(rows: Dict[str, np.ndarray], cells: int) -> np.ndarray:
    """rows as returned by Engine.play -> uint8 [P, record_bytes(cells)]."""
    P = len(rows["reward"])
    rec = np.zeros((P, record_bytes(cells)), np.uint8)
    o = 0
    rec[:, o:o + cells] = rows["board"].reshape(P, cells).astype(np.uint8); o += cells
    rec[:, o:o + 4 * cells] = np.ascontiguousarray(rows["moves_prob"], np.float32).view(np.uint8).reshape(P, 4 * cells); o += 4 * cells
    rec[:, o:o + 4] = np.ascontiguousarray(rows["reward"], np.float32).view(np.uint8).reshape(P, 4); o += 4
    rec[:, o:o + 2] = np.ascontiguousarray(rows["color"], np.int16).view(np.uint8).reshape(P, 2); o += 2
    rec[:, o:o + 2] = np.ascontiguousarray(rows["nlegal"], np.int16).view(np.uint8).reshape(P, 2); o += 2
    rec[:, o:o + 8] = np.ascontiguousarray(rows["game_uid"], np.int64).view(np.uint8).reshape(P, 8)
    return rec


def unpack_rows(rec: np.ndarray, board_size: int) -> Dict[str, np.ndarray]:
    cells = board_size * board_size
    P = len(rec)
    o = 0
    board = rec[:, o:o + cells].astype(np.int32).reshape(P, board_size, board_size); o += cells
    prob = np.ascontiguousarray(rec[:, o:o + 4 * cells]).view(np.float32).reshape(P, cells); o += 4 * cells
    reward = np.ascontiguousarray(rec[:, o:o + 4]).view(np.float32).reshape(P); o += 4
    color = np.ascontiguousarray(rec[:, o:o + 2]).view(np.int16).reshape(P).astype(np.int32); o += 2
    k = np.ascontiguousarray(rec[:, o:o + 2]).view(np.int16).reshape(P).astype(np.int32); o += 2
    uid = np.ascontiguousarray(rec[:, o:o + 8]).view(np.int64).reshape(P)
    return dict(board=board, moves_prob=prob, reward=reward, color=color, nlegal=k, game_uid=uid)


def all_gather_rows(rows: Dict[str, np.ndarray], board_size: int) -> Dict[str, np.ndarray]:
    """Every rank contributes its rows; every rank gets all rows, in rank order."""
    if not is_distributed():
        return rows
    cells = board_size * board_size
    dev = _comm_device()
    world = dist.get_world_size()
    local = torch.from_numpy(pack_rows(rows, cells)).to(dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([local.shape[0]], dtype=torch.int64, device=dev))
    counts = [int(c.item()) for c in counts]
    most = max(counts)
    padded = torch.zeros((most, record_bytes(cells)), dtype=torch.uint8, device=dev)
    padded[:local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)
    rec = torch.cat([p[:c] for p, c in zip(parts, counts)]).cpu().numpy()
    return unpack_rows(rec, board_size)


def all_reduce_metrics(metrics: Dict[str, float]) -> Dict[str, float]:
    """Sum the per-rank self-play metrics (parallel_player.py:50-51 sums over games)."""
    if not is_distributed():
        return metrics
    keys = sorted(metrics)
    t = torch.tensor([float(metrics[k]) for k in keys], dtype=torch.float64, device=_comm_device())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return dict(zip(keys, t.tolist()))


def broadcast_weights(net: torch.nn.Module, src: int = 0) -> None:
    """One flat broadcast of parameters + buffers (2.1 MB for the 6x64 net)."""
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
