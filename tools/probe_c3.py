"""One-off scale check at the C3 size (100 clouds x 1e6 points, 1e7-candidate grid; other grid
counts as a third argument, e.g. 20,20,25,20,20,25 for the 1e8-candidate C4 grid)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cameracalibration_amd as cc
from cameracalibration_amd.synthetic import make_scene
nc, npts = int(sys.argv[1]), int(sys.argv[2])
t = time.time(); s = make_scene(nc, npts); print(f"scene {time.time()-t:.1f} s", flush=True)
opt = cc.Optimizer(0)
t = time.time(); opt.upload_clouds(0, s.clouds, s.base_T_flange); opt.set_plane(s.plane); print(f"upload {time.time()-t:.1f} s", flush=True)
counts = [int(c) for c in sys.argv[3].split(",")] if len(sys.argv) > 3 else [20, 20, 25, 10, 10, 10]
g = cc.Grid.make([-20] * 3 + [-5] * 3, [40.0 / (c - 1) for c in counts[:3]] + [10.0 / (c - 1) for c in counts[3:]], counts)
opt.set_search_mode(1)
t = time.time(); idx, cost, pose, st = opt.grid_search(s.pose_init, g); dt = time.time() - t
print(f"separable {g.size:.3g} candidates: idx {idx} cost {cost:.9g} wall {dt:.2f} s kernel {st.kernel_ms:.0f} ms launches {st.kernel_launches} rescored {st.n_rescored}", flush=True)
opt.set_search_mode(0)
n = 200000
b = max(0, idx - n // 2)
t = time.time(); idx2, cost2, _, st2 = opt.grid_search(s.pose_init, g, b, b + n); dt = time.time() - t
print(f"dense {n} candidates around it: idx {idx2} cost {cost2:.9g} wall {dt:.2f} s -> {n*nc*npts/dt:.3g} residuals/s", flush=True)
t = time.time(); e = opt.get_error(np.tile(pose, (13, 1))); print(f"exact 13 poses {time.time()-t:.3f} s", e[0], flush=True)
o = cc.GdOptions.default(); o.max_iterations = 20
t = time.time(); r = opt.gradient_descent(pose, o); print(f"gd 20 it {time.time()-t:.2f} s fval {r.fval}", flush=True)
