import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cameracalibration_amd as cc
from cameracalibration_amd.synthetic import make_scene
nc, npts, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
s = make_scene(nc, npts)
opt = cc.Optimizer(0)
opt.upload_clouds(0, s.clouds, s.base_T_flange); opt.set_plane(s.plane)
poses = np.tile(s.pose_init, (n, 1)) + np.arange(n)[:, None] * 1e-6
for _ in range(3): opt.get_error(poses)
