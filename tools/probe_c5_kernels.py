"""The kernels of the plane fits and the point report at the 100 clouds x 10^6 points size, for
rocprofv3 (kernel trace and PMC passes): batched RANSAC (inlier_mfma_kernel, exact_count_kernel,
ransac_hyp_kernel), the least-squares plane fit (moments_kernel) and the per-point error report
(point_error_kernel).  Prints the library's own HIP-event timings of the same launches."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cameracalibration_amd as cc
from cameracalibration_amd.synthetic import make_scene

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 100
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
s = make_scene(nc, npts)
rng = np.random.default_rng(5)
clouds = []
for c in s.clouds:                     # a fifth of the points lifted off the plane: hypotheses differ in their counts
    c = c.copy()
    k = len(c) // 5
    c[rng.choice(len(c), k, replace=False), 2] += rng.uniform(0.005, 0.05, k).astype(np.float32)
    clouds.append(c)
opt = cc.Optimizer(0)
opt.upload_clouds(cc.CCAL_RAW, clouds, s.base_T_flange)
opt.set_plane(s.plane)
opt.set_profiling(True)
for rep in range(2):
    opt.profile(reset=True)
    t = time.perf_counter()
    plane, inl, win = opt.ransac_plane(s.pose_true, 4096, 0.01, 7)
    t_r = time.perf_counter() - t
    for _ in range(3):
        fit = opt.fit_plane(s.pose_true)
    dist, cls = opt.point_errors(s.pose_true)
    p = opt.profile()
    print(f"rep {rep}: ransac {t_r:.3f} s (bracket pass {p.ransac_score_ms:.2f} ms = {p.ransac_tests / p.ransac_score_ms / 1e9:.2f} T tests/s, "
          f"{p.ransac_recounted} hypotheses re-counted in {p.ransac_recount_ms:.2f} ms), winner {win} with {inl} inliers; "
          f"moments {p.moments_ms / p.moments_calls:.3f} ms = {p.moments_points / p.moments_calls * 12 / (p.moments_ms / p.moments_calls) / 1e9:.2f} TB/s; "
          f"point report {p.point_error_ms:.3f} ms = {p.point_error_points * (12 + 9) / p.point_error_ms / 1e9:.2f} TB/s (12 B read, 9 B written per point); "
          f"planes {plane} {fit}", flush=True)
