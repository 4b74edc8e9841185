"""Fast-path screening figures: costs and bounds of a few dozen poses, as raw bytes (to compare two builds
bit for bit: PROBE_LIB=<other libccal.so>), and the time of a call.
usage: probe_screen.py <clouds> <points per cloud> <out file>"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cameracalibration_amd as cc
if os.environ.get('PROBE_LIB'):
    import cameracalibration_amd.binding as _b
    _b.lib_path = lambda: os.path.abspath(os.environ['PROBE_LIB'])
from cameracalibration_amd.synthetic import make_scene
nc, npts = int(sys.argv[1]), int(sys.argv[2])
s = make_scene(nc, npts, ragged=True)   # partial last groups and blocks
opt = cc.Optimizer(0)
opt.upload_clouds(0, s.clouds, s.base_T_flange); opt.set_plane(s.plane)
rng = np.random.default_rng(5)
out = []
for n in (1, 13, 16, 17, 32, 48, 64, 100):
    poses = s.pose_init[None, :] + rng.normal(size=(n, 6)) * np.array([2e-3] * 3 + [5e-3] * 3)
    c, b = opt.get_error_bounded(poses)
    out += [np.asarray(c, np.float64), np.asarray(b, np.float64)]
    for _ in range(3):
        opt.get_error_bounded(poses)
    t = time.perf_counter()
    for _ in range(50):
        opt.get_error_bounded(poses)
    print("poses", n, "call_us", (time.perf_counter() - t) / 50 * 1e6, flush=True)
np.concatenate(out).tofile(sys.argv[3])
print("wrote", sys.argv[3], flush=True)
