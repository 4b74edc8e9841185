"""Times the dense search of the benchmark scene with a given build of the library (same-box A/B of kernel variants)."""
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
opt.set_search_mode(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
g = cc.Grid.make([-20] * 3 + [-5] * 3, [40.0 / 9] * 3 + [10.0 / 9] * 3, [10] * 6)
for rep in range(4):
    idx, cost, pose, st = opt.grid_search(s.pose_init, g)
    print(which, idx, cost, f"kernel {st.kernel_ms:.2f} ms", flush=True)
