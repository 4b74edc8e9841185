"""Ad-hoc GPU probe: fast-path throughput at a few sizes (not the benchmark)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cameracalibration_amd as cc
from cameracalibration_amd.synthetic import make_scene

def main():
    nc, npts, ncand_axis = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    s = make_scene(nc, npts)
    opt = cc.Optimizer(0)
    t = time.time(); opt.upload_clouds(0, s.clouds, s.base_T_flange); opt.set_plane(s.plane)
    print("upload s", time.time() - t, flush=True)
    n = ncand_axis
    g = cc.Grid.make([-20] * 3 + [-5] * 3, [40.0 / (n - 1)] * 3 + [10.0 / (n - 1)] * 3, [n] * 6)
    for rep in range(3):
        t = time.time()
        idx, cost, pose, st = opt.grid_search(s.pose_init, g)
        dt = time.time() - t
        print(f"grid {g.size} cands x {s.n_points} pts: wall {dt*1e3:.1f} ms kernel {st.kernel_ms:.1f} ms "
              f"total_dev {st.total_ms:.1f} ms  res/s(kernel) {st.residuals/st.kernel_ms*1e3:.3e} "
              f"frac {st.residuals*8/st.kernel_ms*1e3/157.3e12:.3f} rescored {st.n_rescored} best {idx} {cost:.6e}", flush=True)
    print("true pose err", pose - s.pose_true)
    if len(sys.argv) <= 4: return
    t = time.time(); r = opt.gradient_descent(pose); print("gd", r.iterations, r.converged, r.fval, r.n_evals, time.time() - t, r.xval - s.pose_true)

main()
