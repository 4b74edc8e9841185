"""Times the device voxel-grid downsample and the per-point error report at a given size."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cameracalibration_amd as cc
from cameracalibration_amd.synthetic import make_scene
nc, npts = int(sys.argv[1]), int(sys.argv[2])
s = make_scene(nc, npts)
opt = cc.Optimizer(0)
t = time.time(); opt.upload_clouds(1, s.clouds, s.base_T_flange); print(f"upload raw {time.time()-t:.2f} s", flush=True)
for rep in range(3):
    t = time.time(); n = opt.downsample_raw(0.02); dt = time.time() - t
    print(f"downsample_raw {nc*npts:.3g} points -> {int(np.sum(n))} centroids: {dt*1e3:.1f} ms", flush=True)
opt.set_plane(s.plane)
t = time.time(); d, c = opt.point_errors(s.pose_true, which=1); dt = time.time() - t
print(f"point_errors {nc*npts:.3g} points: {dt*1e3:.1f} ms (includes the copy of {d.nbytes/1e6:.0f} MB to the host)", flush=True)
