"""Same-box A/B of library builds on the exact path: whole-chain requests (many poses), a descent, a C5-like
lock-step batch.  Usage: python tools/ab_exact.py <library file name under cameracalibration_amd/>"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import cameracalibration_amd.binding as b
which = sys.argv[1]
b.lib_path = lambda: os.path.join(os.getcwd(), "cameracalibration_amd", which)
import cameracalibration_amd as cc
from cameracalibration_amd.synthetic import make_scene
s = make_scene(20, 300000)
opt = cc.Optimizer(0)
opt.upload_clouds(0, s.clouds, s.base_T_flange); opt.set_plane(s.plane)
out = []
for n in (1, 13, 256, 1024):
    poses = np.tile(s.pose_init, (n, 1)) + np.arange(n)[:, None] * 1e-6
    ref = opt.get_error(poses)
    t = time.time()
    for _ in range(5): opt.get_error(poses)
    dt = (time.time() - t) / 5
    out.append(f"n={n}: {dt * 1e3:.3f} ms ({dt / n * 1e6:.2f} us/pose, sum {float(np.sum(ref)):.17g})")
o = cc.GdOptions.default()
t = time.time(); r = opt.gradient_descent(s.pose_init, o); dt = time.time() - t
out.append(f"gd {r.iterations} {r.n_evals} {r.fval!r} {dt:.4f} s")
print(which, " | ".join(out), flush=True)
