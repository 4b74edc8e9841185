"""Rate of whole-chain exact evaluations (exact_chain_kernel) at 20 x 300k points: 256 and 1024 poses per call."""
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
import cameracalibration_amd as cc
from cameracalibration_amd.synthetic import make_scene
s = make_scene(20, 300000)
opt = cc.Optimizer(0)
opt.upload_clouds(0, s.clouds, s.base_T_flange); opt.set_plane(s.plane)
for n in (256, 1024):
    poses = np.tile(s.pose_init, (n, 1)) + np.arange(n)[:, None] * 1e-6
    opt.get_error(poses)
    t = time.time(); e = opt.get_error(poses); dt = time.time() - t
    print(f"{n} poses: {dt*1e3:.2f} ms, {dt/n*1e6:.2f} us/pose, checksum {float(np.sum(e)):.17g}", flush=True)
    t = time.time(); c, m, mx = opt.per_cloud_error(poses[0]); dt = time.time() - t
    print(f"per-cloud report of one pose {dt*1e3:.2f} ms")
