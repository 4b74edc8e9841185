"""Multi-GPU candidate sharding: one process per GPU, one tiny exchange.

The discrete search shards naturally over candidates (contiguous ranges of the
linear grid index); every rank keeps all clouds resident and scores its range
with ``Optimizer.grid_search``.  The only exchange of the path is the arg-min:
each rank contributes one (cost, index) record, the records are all-gathered
(RCCL on GPUs -- backend "nccl" -- or gloo on CPUs) and every rank resolves the
lexicographic minimum locally, so the winner is the lowest index among equal
costs exactly as the reference's strict '<' scan picks it
(src/camera_calibration_optimization.cpp:542).
"""

from __future__ import annotations

from typing import Tuple

import numpy as np


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous candidate range of `rank`; ranges tile [0,total) in rank order."""
    return (total * rank) // world, (total * (rank + 1)) // world


def merge_descents(fvals, xvals, first_start: int, device=None, group=None):
    """Multi-start refinement sharded over ranks (replicas, no data-path collective): every rank
    has run the starts [first_start, first_start + len(fvals)) of one global list; the best
    descent overall -- lowest final cost, lowest start index on ties -- is returned on every
    rank as (fval, global_start_index, xval).  One all-gather of dim + 2 doubles per rank."""
    import torch
    import torch.distributed as dist

    f = np.asarray(fvals, dtype=np.float64)
    x = np.asarray(xvals, dtype=np.float64).reshape(len(f), -1) if len(f) else np.zeros((0, 0))
    if len(f):
        k = int(np.lexsort((np.arange(len(f)), f))[0])
        rec = np.concatenate([[f[k], float(first_start + k)], x[k]])
    else:
        rec = None
    if not (dist.is_available() and dist.is_initialized()):
        if rec is None:
            raise ValueError("merge_descents: no descents")
        return float(rec[0]), int(rec[1]), rec[2:].copy()
    world = dist.get_world_size(group)
    dim = torch.tensor([0 if rec is None else len(rec) - 2], dtype=torch.int64, device=device)
    dist.all_reduce(dim, op=dist.ReduceOp.MAX, group=group)
    width = int(dim.item()) + 2
    mine = torch.full((width,), float("inf"), dtype=torch.float64, device=device)
    if rec is not None:
        mine[: len(rec)] = torch.from_numpy(rec)
    else:
        mine[1] = -1.0                                   # an empty shard
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    recs = torch.stack(out).cpu().numpy()
    valid = recs[:, 1] >= 0
    if not valid.any():
        raise ValueError("merge_descents: no descents on any rank")
    order = np.lexsort((recs[:, 1], recs[:, 0]))
    w = next(int(i) for i in order if valid[i])
    return float(recs[w, 0]), int(recs[w, 1]), recs[w, 2:].copy()


def merge_argmin(cost: float, index: int, device=None, group=None) -> Tuple[float, int, int]:
    """All-gathers (cost, index) over the default process group and returns
    (best_cost, best_index, winner_rank); index < 0 marks an empty shard.
    Works without an initialised process group (single process)."""
    import torch
    import torch.distributed as dist

    from .binding import argmin_merge

    if not (dist.is_available() and dist.is_initialized()):
        return float(cost), int(index), 0
    world = dist.get_world_size(group)
    # indices < 2^53 are exact in float64, so one dtype carries the whole record
    rec = torch.tensor([float(cost), float(index)], dtype=torch.float64, device=device)
    out = [torch.empty_like(rec) for _ in range(world)]
    dist.all_gather(out, rec, group=group)
    recs = torch.stack(out).cpu().numpy()
    w = argmin_merge(recs[:, 0], recs[:, 1].astype(np.int64))
    if w < 0:
        return 1e9, -1, -1
    return float(recs[w, 0]), int(recs[w, 1]), int(w)
