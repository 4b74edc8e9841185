"""Times the least-squares plane fit (moments kernel, HBM-bound: 12 B per point) at a given size."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cameracalibration_amd as cc
from cameracalibration_amd.synthetic import make_scene
nc, npts = int(sys.argv[1]), int(sys.argv[2])
s = make_scene(nc, npts)
opt = cc.Optimizer(0)
opt.upload_clouds(1, s.clouds, s.base_T_flange)
for rep in range(5):
    t = time.time(); p = opt.fit_plane(s.pose_true, which=1); dt = time.time() - t
    print(f"fit_plane {nc*npts:.3g} points: {dt*1e3:.2f} ms -> {12*nc*npts/dt/1e9:.0f} GB/s  plane {p}", flush=True)
