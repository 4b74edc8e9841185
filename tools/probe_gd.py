import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cameracalibration_amd as cc
from cameracalibration_amd.synthetic import make_scene
nc, npts = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
s = make_scene(nc, npts)
opt = cc.Optimizer(0)
opt.upload_clouds(0, s.clouds, s.base_T_flange); opt.set_plane(s.plane)
for n in (1, 6, 13, 20, 25, 30, 40, 52, 64, 100, 128, 256):
    poses = np.tile(s.pose_init, (n, 1)) + np.arange(n)[:, None] * 1e-6
    opt.get_error(poses)
    t = time.time(); 
    for _ in range(10): opt.get_error(poses)
    dt = (time.time()-t)/10
    print(f"exact eval n={n}: {dt*1e3:.2f} ms  {dt/n*1e6:.1f} us/pose", flush=True)
o = cc.GdOptions.default(); o.max_iterations = iters
t = time.time(); r = opt.gradient_descent(s.pose_init, o); print("gd", iters, "it", r.n_evals, time.time() - t, flush=True)
