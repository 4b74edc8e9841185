"""Upload time at 20 x 300k and 100 clouds x 1e6 points (later calls of each: buffers already allocated)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cameracalibration_amd as cc
from cameracalibration_amd.synthetic import make_scene
opt = cc.Optimizer(0)
for nc, npts in ((20, 300000), (100, 1000000)):
    s = make_scene(nc, npts)
    for rep in range(4):
        t = time.time(); opt.upload_clouds(0, s.clouds, s.base_T_flange); dt = time.time() - t
        print(f"{nc} x {npts}: upload {dt*1e3:.1f} ms ({nc*npts*12/dt/1e9:.1f} GB/s of caller data)", flush=True)
    opt.set_plane(s.plane)
    print("cost at the initial pose", opt.get_error(s.pose_init)[0], opt.get_error_fast(s.pose_init[None])[0], flush=True)
