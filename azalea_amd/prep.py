"""Batching helpers: list of dataclass rows -> dict of zero-padded arrays (azalea/prep.py)."""
import dataclasses as dc
from typing import Dict, Sequence

import numpy as np
import torch


def pad(mats, size=None):
    """Stack scalars, or zero-pad n-d arrays of ragged shape to a common size (prep.py:70-86)."""
    first = mats[0]
    if isinstance(first, (int, float)):
        assert size is None
        return np.array(mats)
    shapes = np.array([np.shape(m) for m in mats], dtype=np.int64).reshape(len(mats), -1)
    largest = shapes.max(axis=0)
    if size is None:
        size = largest
    else:
        assert (largest <= np.asarray(size)).all()
    out = np.zeros((len(mats),) + tuple(int(s) for s in size), dtype=first.dtype)
    for i, m in enumerate(mats):
        out[(i,) + tuple(slice(s) for s in np.shape(m))] = m
    return out


def _columns(seq: Sequence) -> Dict[str, list]:
    names = [f.name for f in dc.fields(seq[0])]
    return {n: [getattr(row, n) for row in seq] for n in names}


def batch_games(seq) -> Dict[str, np.ndarray]:
    """Game states -> {'color','legal_moves','result','board'} arrays (prep.py:15-21)."""
    return {name: pad(col) for name, col in _columns(seq).items()}


def batch_replays(seq) -> Dict[str, np.ndarray]:
    """Replay records -> flat dict: the state's fields, then moves_prob and reward (prep.py:24-33)."""
    cols = _columns([rec.state for rec in seq])
    for name, col in _columns(seq).items():
        if name != "state":
            cols[name] = col
    return {name: pad(col) for name, col in cols.items()}


def torch_batch_replays(seq) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(v) for k, v in batch_replays(seq).items()}
