"""One-off scale check at the C5 size: batched RANSAC (4096 hypotheses x 100 clouds x 1e6 points)
and 64 lock-step gradient descents (iteration cap given on the command line)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cameracalibration_amd as cc
from cameracalibration_amd.synthetic import make_scene
nc, npts, cap = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
s = make_scene(nc, npts)
opt = cc.Optimizer(0)
opt.upload_clouds(0, s.clouds, s.base_T_flange); opt.set_plane(s.plane)
t = time.time(); plane, inl, _ = opt.ransac_plane(s.pose_true, 4096, 0.01, 7, which=0); dt = time.time() - t
print(f"ransac 4096 hyp x {nc*npts:.3g} pts: {dt:.2f} s, inliers {inl} of {nc*npts}, plane {plane}, truth {s.plane}", flush=True)
rng = np.random.default_rng(1)
starts = s.pose_true + rng.uniform(-1, 1, (64, 6)) * np.array([0.01] * 3 + [0.04] * 3)
o = cc.GdOptions.default(); o.max_iterations = cap
t = time.time(); rs = opt.gradient_descent(starts, o); dt = time.time() - t
print(f"64 descents x {cap} iterations: {dt:.2f} s, evals {sum(r.n_evals for r in rs)}, best fval {min(r.fval for r in rs):.6g}", flush=True)
