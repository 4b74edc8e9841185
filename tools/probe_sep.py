import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cameracalibration_amd as cc
if os.environ.get('PROBE_LIB'):
    import cameracalibration_amd.binding as _b
    _b.lib_path = lambda: os.path.abspath(os.environ['PROBE_LIB'])
from cameracalibration_amd.synthetic import make_scene
nc, npts, axis = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
s = make_scene(nc, npts)
opt = cc.Optimizer(0)
opt.upload_clouds(0, s.clouds, s.base_T_flange); opt.set_plane(s.plane)
g = cc.Grid.make([-20] * 3 + [-5] * 3, [40.0 / (axis - 1)] * 3 + [10.0 / (axis - 1)] * 3, [axis] * 6)
modes = [int(m) for m in sys.argv[4].split(',')] if len(sys.argv) > 4 else [1, 0]
for mode in modes:
    opt.set_search_mode(mode)
    for rep in range(3):
        t = time.time(); idx, cost, pose, st = opt.grid_search(s.pose_init, g); dt = time.time() - t
        print(f"mode {mode}: idx {idx} cost {cost:.9g} wall {dt*1e3:.1f} ms kernel {st.kernel_ms:.2f} ms total_dev {st.total_ms:.2f} ms rescored {st.n_rescored}", flush=True)
