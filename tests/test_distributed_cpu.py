"""world_size-2 (and 3) gloo tests of the sharded discrete search: each rank
scans its candidate range (CPU oracle standing in for the GPU call), the
(cost, index) records are all-gathered and merged exactly as bench.py and the
multi-GPU path do.  Result must equal the unsharded scan, ties included."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, ret):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from cameracalibration_amd.distributed import merge_argmin, shard_range
    from cameracalibration_amd.synthetic import make_scene
    from oracle import oracle as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s = make_scene(3, 150, seed=5)
        o = O.OracleScene(s.clouds, s.base_T_flange)
        if mode == "ties":     # zero step on the outermost axis: every shard sees the same costs
            g = O.make_grid([-20] * 3 + [-5] * 3, [0, 20, 20, 5, 5, 5], [world * 2, 3, 3, 3, 3, 3])
        else:
            g = O.make_grid([-20] * 3 + [-5] * 3, [20] * 3 + [5] * 3, [3] * 6)
        total = int(O.lib().orc_grid_size(g))
        b, e = shard_range(total, rank, world)
        if mode == "empty" and rank == 1:
            e = b                                           # an empty shard contributes index -1
        idx, cost, _, _ = o.grid_search(s.plane, s.pose_init, g, b, e)
        best_cost, best_idx, winner = merge_argmin(cost, idx)
        ret[rank] = (best_cost, best_idx, winner, idx)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "plain"), (3, "plain"), (2, "ties"), (2, "empty")])
def test_sharded_search_equals_unsharded(world, mode, oracle):
    from cameracalibration_amd.synthetic import make_scene
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), mode, ret), nprocs=world, join=True)
    s = make_scene(3, 150, seed=5)
    o = oracle.OracleScene(s.clouds, s.base_T_flange)
    if mode == "ties":
        g = oracle.make_grid([-20] * 3 + [-5] * 3, [0, 20, 20, 5, 5, 5], [world * 2, 3, 3, 3, 3, 3])
    else:
        g = oracle.make_grid([-20] * 3 + [-5] * 3, [20] * 3 + [5] * 3, [3] * 6)
    want_idx, want_cost, _, costs = o.grid_search(s.plane, s.pose_init, g, want_costs=True)
    results = [ret[r] for r in range(world)]
    assert len({r[:3] for r in results}) == 1, "ranks disagree on the merged result"
    best_cost, best_idx, winner, _ = results[0]
    if mode == "empty":
        total = len(costs)
        lo = (total * 1) // world
        keep = np.concatenate([costs[:lo]])              # rank 1 scanned nothing
        assert results[1][3] == -1
        assert best_idx == int(np.argmin(keep)) and best_cost == keep[best_idx]
    else:
        assert (best_idx, best_cost) == (want_idx, want_cost)
    if mode == "ties":
        assert winner == 0 and best_idx < len(costs) // (world * 2)


def test_shard_ranges_tile_the_grid():
    from cameracalibration_amd.distributed import shard_range
    for total in (1, 7, 729, 11 ** 6, 10 ** 8):
        for world in (1, 2, 3, 4, 8):
            edges = [shard_range(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))


def test_merge_without_process_group():
    from cameracalibration_amd.distributed import merge_argmin
    assert merge_argmin(0.25, 17) == (0.25, 17, 0)


def _gd_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from cameracalibration_amd.distributed import merge_descents, shard_range
    from cameracalibration_amd.synthetic import make_scene
    from oracle import oracle as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s = make_scene(2, 120, seed=9)
        o = O.OracleScene(s.clouds, s.base_T_flange)
        starts = _starts(s)
        b, e = shard_range(len(starts), rank, world)
        opts = O.gd_default_opts()
        opts.max_iterations = 6
        rs = [o.gd_minimize(s.plane, x0, opts) for x0 in starts[b:e]]
        fval, idx, x = merge_descents([r.fval for r in rs], [list(r.xval[:6]) for r in rs], b)
        ret[rank] = (fval, idx, tuple(x))
    finally:
        dist.destroy_process_group()


def _starts(s):
    rng = np.random.default_rng(2)
    st = s.pose_true + rng.uniform(-1, 1, (5, 6)) * np.array([0.004] * 3 + [0.02] * 3)
    return np.concatenate([st, st[1:2]])                 # a duplicate start: equal cost, higher index


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_multi_start_descent(world, oracle):
    """Restarts sharded over ranks (the CPU oracle standing in for the GPU call): every rank ends
    with the best descent of the whole list, the lower start index winning a tie."""
    from cameracalibration_amd.synthetic import make_scene
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_gd_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    s = make_scene(2, 120, seed=9)
    o = oracle.OracleScene(s.clouds, s.base_T_flange)
    opts = oracle.gd_default_opts()
    opts.max_iterations = 6
    rs = [o.gd_minimize(s.plane, x0, opts) for x0 in _starts(s)]
    f = np.array([r.fval for r in rs])
    k = int(np.lexsort((np.arange(len(f)), f))[0])
    results = [ret[r] for r in range(world)]
    assert len(set(results)) == 1, "ranks disagree on the merged result"
    assert results[0][0] == f[k] and results[0][1] == k
    assert results[0][2] == tuple(rs[k].xval[:6])
    assert rs[1].fval == rs[5].fval and k != 5


def test_merge_descents_without_process_group():
    from cameracalibration_amd.distributed import merge_descents
    f, i, x = merge_descents([0.5, 0.25, 0.25], [[1] * 6, [2] * 6, [3] * 6], 10)
    assert (f, i, list(x)) == (0.25, 11, [2.0] * 6)
