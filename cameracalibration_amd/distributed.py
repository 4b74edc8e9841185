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
