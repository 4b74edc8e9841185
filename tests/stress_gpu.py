"""Randomised GPU-vs-oracle stress run.  By hand on a GPU box, for as long as wanted:

    python tests/stress_gpu.py [seconds] [seed] [exact|search|gd|ingest]

and, as a bounded fixed-seed slice, under pytest -m gpu (tests/test_gpu_stress_slice.py).

Draws scenes, poses, planes, grids and optimiser options at random -- sizes around the kernels'
run/segment/pass boundaries, coordinate scales over many decades, special values -- and checks the
same bars as the parity tests: exact path bit-equal to the oracle (cost, per-cloud mean and maximum,
caller planes, per-point report, device voxel grid), descents bit-equal, searches equal to the
oracle's winner, the separable pass within the fast tolerance of the dense kernel with the same winner,
plane fit and RANSAC to float rounding with identical inlier counts.  Prints one line per mismatch and a
summary; exit status 1 if anything differed.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import cameracalibration_amd as cc                      # noqa: E402
from cameracalibration_amd.synthetic import make_scene  # noqa: E402
from oracle import oracle as O                          # noqa: E402

BOUNDARY_SIZES = [1, 2, 7, 8, 9, 63, 64, 65, 511, 512, 513, 2047, 2048, 2049, 4095, 4096, 4097, 8191, 8192, 8193,
                  16383, 16384, 16385, 32768, 40000]


def same(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def random_clouds(rng):
    n_clouds = int(rng.integers(1, 6))
    scale = 10.0 ** rng.uniform(-3, 2)
    clouds = []
    for _ in range(n_clouds):
        n = int(rng.choice(BOUNDARY_SIZES)) if rng.random() < 0.7 else int(rng.integers(1, 30000))
        c = (rng.normal(0, scale, (n, 3)) + rng.normal(0, scale, 3)).astype(np.float32)
        kind = rng.random()
        if kind < 0.1:
            c[:, 2] = 0.0                                   # on a coordinate plane: exact zeros likely
        elif kind < 0.2 and n > 4:
            c[rng.integers(0, n, 3)] *= np.float32(1e6)     # a few huge points
        elif kind < 0.25 and n > 4:
            c[rng.integers(0, n)] = np.float32(1e-42)       # denormal coordinates
        elif kind < 0.3 and n > 4:
            c[rng.integers(0, n), rng.integers(0, 3)] = rng.choice([np.nan, np.inf, -np.inf])
        clouds.append(np.ascontiguousarray(c))
    B = np.zeros((n_clouds, 16))
    for j in range(n_clouds):
        B[j] = O.pose_to_matrix(np.concatenate([rng.normal(0, scale, 3), rng.uniform(-3, 3, 3)])).reshape(16)
    return clouds, B, scale


def check_exact(opt, rng, stats):
    clouds, B, scale = random_clouds(rng)
    plane = rng.normal(0, 1, 4).astype(np.float32)
    plane[3] *= np.float32(scale)
    if not np.any(plane[:3]):
        plane[2] = 1
    opt.upload_clouds(0, clouds, B)
    opt.set_plane(plane)
    o = O.OracleScene(clouds, B)
    n_poses = int(rng.choice([1, 2, 13, 30]))
    poses = np.concatenate([rng.normal(0, scale, (n_poses, 3)), rng.uniform(-3, 3, (n_poses, 3))], axis=1)
    want = np.array([o.get_error(p, plane) for p in poses])
    got = opt.get_error(poses)
    stats["exact"] += 1
    if not same(got, want):
        stats["bad"] += 1
        print("MISMATCH exact", [len(c) for c in clouds], scale, got, want, flush=True)
    cost, mean, mx = opt.per_cloud_error(poses[0])
    om, ox = o.per_cloud_error(poses[0], plane)
    if not (same(mean, om) and same(mx, ox)):
        stats["bad"] += 1
        print("MISMATCH per-cloud", [len(c) for c in clouds], scale, mean, om, mx, ox, flush=True)
    planes = rng.normal(0, 1, (n_poses, 4)) * 10.0 ** rng.uniform(-200, 200)
    want = np.array([o.get_error_plane(p, pl) for p, pl in zip(poses, planes)])
    got = opt.get_error(poses, planes)
    if not same(got, want):
        stats["bad"] += 1
        print("MISMATCH caller planes", [len(c) for c in clouds], planes[0], got, want, flush=True)


def random_grid(rng):
    counts = [int(rng.integers(1, 6)) for _ in range(6)]
    mins = list(rng.uniform(-20, 0, 3)) + list(rng.uniform(-5, 0, 3))
    steps = [float(rng.choice([0.0, rng.uniform(0.5, 8)], p=[0.1, 0.9])) for _ in range(3)] + \
            [float(rng.choice([0.0, rng.uniform(0.2, 2)], p=[0.1, 0.9])) for _ in range(3)]
    return mins, steps, counts


def check_search(opt, rng, stats):
    s = make_scene(int(rng.integers(1, 5)), int(rng.choice([50, 300, 1000, 5000])), seed=int(rng.integers(1 << 30)),
                   ragged=bool(rng.integers(2)))
    opt.upload_clouds(0, s.clouds, s.base_T_flange)
    opt.set_plane(s.plane)
    o = O.OracleScene(s.clouds, s.base_T_flange)
    mins, steps, counts = random_grid(rng)
    g = cc.Grid.make(mins, steps, counts)
    og = O.make_grid(mins, steps, counts)
    total = g.size
    begin = int(rng.integers(0, total))
    end = int(rng.integers(begin, total + 1))
    if rng.random() < 0.5:
        begin, end = 0, total
    oi, oc, _, _ = o.grid_search(s.plane, s.pose_init, og, begin, end)
    stats["search"] += 1
    res = {}
    for mode in (0, 1):
        opt.set_search_mode(mode)
        idx, cost, pose, _ = opt.grid_search(s.pose_init, g, begin, end)
        res[mode] = opt.grid_costs(s.pose_init, g, begin, end) if end > begin else np.zeros(0, np.float32)
        if (idx, cost) != (oi, oc) and not (end == begin and idx == -1):
            stats["bad"] += 1
            print("MISMATCH search mode", mode, counts, steps, begin, end, (idx, cost), (oi, oc), flush=True)
    opt.set_search_mode(0)
    if end > begin and not np.allclose(res[1], res[0], rtol=1e-5, atol=0):
        stats["bad"] += 1
        k = int(np.argmax(np.abs(res[1] - res[0]) / res[0]))
        print("MISMATCH separable costs", counts, steps, begin, end, k, res[1][k], res[0][k], flush=True)


def check_gd(opt, rng, stats):
    wild = rng.random() < 0.25
    if wild:
        # arbitrary clouds, robot poses and plane (scales over decades, special values): the error
        # bound of the line-search screening has to hold there too
        clouds, B, scale = random_clouds(rng)
        clouds = [c[:3000] for c in clouds]
        if rng.random() < 0.4:
            # robot "poses" that are not rigid: scaled and sheared 3x3 parts
            B = B.copy()
            B[:, [0, 1, 2, 4, 5, 6, 8, 9, 10]] *= 10.0 ** rng.uniform(-1, 1, (len(B), 9))
        plane = rng.normal(0, 1, 4).astype(np.float32)
        plane[3] *= np.float32(scale)
        if not np.any(plane[:3]):
            plane[2] = 1
        pose_init = np.concatenate([rng.normal(0, scale, 3), rng.uniform(-3, 3, 3)])

        class S:
            pass
        s = S()
        s.clouds, s.base_T_flange, s.plane, s.pose_init = clouds, B, plane, pose_init
    else:
        s = make_scene(int(rng.integers(1, 4)), int(rng.choice([60, 400, 2500])), seed=int(rng.integers(1 << 30)),
                       ragged=bool(rng.integers(2)))
    opt.upload_clouds(0, s.clouds, s.base_T_flange)
    opt.set_plane(s.plane)
    opt.set_gd_screening(bool(rng.random() < 0.85))
    opt.set_chain_split(bool(rng.random() < 0.85))
    opt.set_gd_pipelining(bool(rng.random() < 0.5))
    o = O.OracleScene(s.clouds, s.base_T_flange)
    go, oo = cc.GdOptions.default(), O.gd_default_opts()
    it = int(rng.integers(1, 4 if wild else 12))
    mom = float(rng.choice([0.0, 0.4, 0.9]))
    c2 = float(rng.choice([0.9, 0.2, -1.0]))
    dec = float(rng.choice([0.8, 0.5, 0.3]))
    max_it = int(rng.choice([0, 3, 40]))
    eps = float(rng.choice([1.4901161193847656e-08, 1e-6, 1e-4]))
    for opts in (go, oo):
        opts.max_iterations = it
        opts.momentum = mom
        opts.wolfe_c2 = c2
        opts.wolfe_decrease = dec
        opts.wolfe_max_it = max_it
        opts.wolfe_min_step = 1e-5 if c2 < 0 else 1e-10
        opts.fd_eps = eps
    dim10 = rng.random() < 0.3
    x0 = s.pose_init + rng.normal(0, 1, 6) * np.array([0.002] * 3 + [0.01] * 3)
    if dim10:
        x0 = np.concatenate([x0, s.plane.astype(np.float64)])
    r = opt.gradient_descent(x0, go)
    w = o.gd_minimize(s.plane, x0, oo)
    stats["gd"] += 1
    n = 10 if dim10 else 6
    if not (same(r.xval[:n], np.array(w.xval[:n])) and
            (r.iterations, bool(r.converged), r.n_evals) == (w.iterations, bool(w.converged), w.n_evals) and
            same(r.fval, w.fval)):
        stats["bad"] += 1
        print("MISMATCH gd", n, it, mom, c2, dec, max_it, eps, r.xval[:n], list(w.xval[:n]), r.iterations, w.iterations,
              r.n_evals, w.n_evals, flush=True)


def check_ingest(opt, rng, stats):
    """Device voxel grid and the per-point error report: bit-equal; plane fit and RANSAC: same
    minimiser / same winner and inlier count, coefficients to float rounding."""
    n_clouds = int(rng.integers(1, 4))
    clouds = []
    for _ in range(n_clouds):
        n = int(rng.choice([1, 5, 700, 20000, 90000]))
        c = rng.uniform(-0.4, 0.4, (n, 3)).astype(np.float32) * np.float32([1, 1, rng.choice([1.0, 0.01])])
        if n > 100 and rng.random() < 0.3:
            c[rng.integers(0, n, 5), rng.integers(0, 3)] = np.nan
        clouds.append(np.ascontiguousarray(c))
    B = np.tile(np.eye(4), (n_clouds, 1, 1))
    opt.upload_clouds(1, clouds, B)
    leaf = float(rng.choice([0.02, 0.005, 0.1, 0.0013]))
    stats["ingest"] += 1
    try:
        sizes = opt.downsample_raw(leaf)
    except cc.CcalError as e:
        # a cloud whose every point is non-finite leaves nothing to downsample
        if not any(np.isfinite(c).all(axis=1).any() for c in clouds):
            return
        if "empty" in str(e):
            return
        raise
    for j, c in enumerate(clouds):
        want = O.voxel_grid(c, leaf)
        got = opt.download_cloud(0, j)
        if sizes[j] != len(want) or not same(got, want):
            stats["bad"] += 1
            print("MISMATCH voxel", len(c), leaf, sizes[j], len(want), flush=True)
    s = make_scene(int(rng.integers(1, 5)), int(rng.choice([40, 900, 12000])), seed=int(rng.integers(1 << 30)),
                   ragged=bool(rng.integers(2)), noise=float(rng.choice([0.0005, 0.003])))
    opt.upload_clouds(1, s.clouds, s.base_T_flange)
    opt.set_plane(s.plane)
    o = O.OracleScene(s.clouds, s.base_T_flange)
    pose = s.pose_init + rng.normal(0, 1, 6) * np.array([0.003] * 3 + [0.01] * 3)
    viewer = bool(rng.integers(2))
    wd, wc = o.point_errors(pose, s.plane.astype(np.float64), int(viewer))
    gd, gc = opt.point_errors(pose, viewer_arithmetic=viewer, which=1)
    if not (same(gd, wd) and np.array_equal(gc, wc)):
        stats["bad"] += 1
        bad_d = np.flatnonzero(~((gd == wd) | (np.isnan(gd) & np.isnan(wd))))
        bad_c = np.flatnonzero(gc != wc)
        print("MISMATCH point errors", viewer, s.n_points, len(bad_d), len(bad_c),
              [(int(i), float(gd[i]).hex(), float(wd[i]).hex()) for i in bad_d[:3]],
              [(int(i), int(gc[i]), int(wc[i]), float(gd[i]).hex()) for i in bad_c[:3]], flush=True)
    want = O.fit_plane(o.combined_cloud(pose))
    got = opt.fit_plane(pose, which=1)
    if not np.allclose(got, want, rtol=0, atol=3e-8):
        stats["bad"] += 1
        print("MISMATCH fit_plane", s.n_points, got, want, flush=True)
    if s.n_points >= 40:
        n_hyp, seed, thr = int(rng.choice([16, 100, 700])), int(rng.integers(1 << 30)), float(rng.choice([0.001, 0.004]))
        want, want_inl, want_h = o.ransac_plane(pose, n_hyp, thr, seed)
        got, got_inl, got_h = opt.ransac_plane(pose, n_hyp, thr, seed, which=1)
        if (got_inl, got_h) != (want_inl, want_h) or not np.allclose(got, want, rtol=0, atol=5e-7):
            stats["bad"] += 1
            print("MISMATCH ransac", s.n_points, n_hyp, thr, seed, got, want, got_inl, want_inl, flush=True)


def one_round(opt, rng, stats, only=""):
    u = rng.random()
    if only:
        {"exact": check_exact, "search": check_search, "gd": check_gd, "ingest": check_ingest}[only](opt, rng, stats)
    elif u < 0.35:
        check_exact(opt, rng, stats)
    elif u < 0.6:
        check_search(opt, rng, stats)
    elif u < 0.8:
        check_gd(opt, rng, stats)
    else:
        check_ingest(opt, rng, stats)


def run_rounds(rounds, seed):
    """A fixed number of rounds from a fixed seed (what tests/test_gpu_stress_slice.py runs): the same
    scenes and checks on every run.  Returns the statistics; stats["bad"] counts mismatches."""
    rng = np.random.default_rng(seed)
    O.lib()
    stats = {"exact": 0, "search": 0, "gd": 0, "ingest": 0, "bad": 0}
    old = np.seterr(all="ignore")
    opt = cc.Optimizer(0)
    try:
        for _ in range(rounds):
            one_round(opt, rng, stats)
    finally:
        opt.close()
        np.seterr(**old)
    return stats


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
    only = sys.argv[3] if len(sys.argv) > 3 else ""
    rng = np.random.default_rng(seed)
    O.lib()
    opt = cc.Optimizer(0)
    stats = {"exact": 0, "search": 0, "gd": 0, "ingest": 0, "bad": 0}
    t0 = time.time()
    last = t0
    np.seterr(all="ignore")
    while time.time() - t0 < seconds:
        one_round(opt, rng, stats, only)
        if time.time() - last > 20:
            last = time.time()
            print("progress", stats, flush=True)
    opt.close()
    print("done", stats, flush=True)
    sys.exit(1 if stats["bad"] else 0)


if __name__ == "__main__":
    main()
